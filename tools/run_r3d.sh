set -o pipefail
O=gpurun_out/r3d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
E1=128x64x32768; E2=256x128x8192; E3=512x256x2048; D0=512x512x2048; D1=512x256x8192; D2=256x128x32768; D3=128x64x131072
python tools/ab_multi.py --rounds 5 --steps 20 VP_WGRAD5=0 VP_WGRAD5_BLOCKS=128 VP_WGRAD5_BLOCKS=144 VP_WGRAD5_BLOCKS=160 VP_WGRAD5_BLOCKS=176 \
  VP_WGRAD5_SPEC=$E1:256+$E2:256+$E3:256 VP_WGRAD5_SPEC=$E1:192+$E2:192+$E3:192 VP_WGRAD5_SPEC=$D3:96+$D2:96 VP_WGRAD5_SPEC=$D3:160+$D2:160+$D1:160 \
  VP_WGRAD5_SPEC=$D0:192+$E3:192 VP_WGRAD5_SPEC=$E1:256 > $O/ab_spec.log 2>&1; echo "ab rc=$?"; tail -12 $O/ab_spec.log
python tools/ab_multi.py --rounds 4 --steps 20 --gan VP_WGRAD5=0 VP_WGRAD5_BLOCKS=128 VP_WGRAD5_BLOCKS=192 VP_WGRAD5_BLOCKS=256 > $O/ab_gan.log 2>&1; echo "gan rc=$?"; tail -5 $O/ab_gan.log
