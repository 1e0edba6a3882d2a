#!/bin/bash
# A/B of the swizzled 128x64 LDS layout: the snapshot's library (SWZ on) against one rebuilt on the box with -DVP_IGEMM16_SWZ_ON=0
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/swz
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_properties.py tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_f16x2.py -q -x -m gpu > $O/pytest.log 2>&1
rc=$?
tail -n 4 $O/pytest.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants"
cp vae_play_amd/libvaeplay_hip.so $O/lib_swz1.so
( cd vae_play_amd/csrc && touch igemm16.h && make -j12 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-result -DVP_IGEMM16_SWZ_ON=0" > $O/build0.log 2>&1 ) || { tail -n 5 $O/build0.log; exit 1; }
cp vae_play_amd/libvaeplay_hip.so $O/lib_swz0.so
for v in 1 0 1 0 1 0; do
  cp $O/lib_swz$v.so vae_play_amd/libvaeplay_hip.so
  timeout -k 10 200 python3 $B --tags-out $O/tags$v.json > $O/b$v.json 2> $O/b$v.err || exit 1
  python3 -c "import json;d=json.load(open('$O/b$v.json'));t=json.load(open('$O/tags$v.json'));print('SWZ=$v', d['ms_per_step'], {k:round(v['ms']*1e3) for k,v in t.items() if k in ('dec3.fwd','dec1.dgrad','enc1.fwd','enc1.dgrad','dec0.fwd','enc2.dgrad','enc3.dgrad')})"
done
for v in 1 0; do
  cp $O/lib_swz$v.so vae_play_amd/libvaeplay_hip.so
  timeout -k 10 200 python3 $B --precision f16x2 > $O/x$v.json 2> $O/x$v.err || exit 1
  python3 -c "import json;d=json.load(open('$O/x$v.json'));print('f16x2 SWZ=$v', d['ms_per_step'])"
done
rm -f $O/lib_swz0.so $O/lib_swz1.so
