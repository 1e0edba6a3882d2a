#!/bin/bash
# Bench lines (with roofline + cpu_baseline) and rocprofv3 kernel tables of the widened rows: fused VAE-GAN step, BE heads,
# train_BE_GAN iteration, font GAN at config-5 size.  Outputs under gpurun_out/wide; summaries go to profiles/ via profiles/make_summary.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wide
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 tools/bench_vaegan.py --path fused --steps 20 --warmup 5 > $O/vaegan_fused_bench.json 2> $O/vaegan.err || exit 1
timeout -k 10 300 python3 tools/bench_be_heads.py --precision bf16x3 > $O/be_heads_bench.json 2> $O/be_heads.err || exit 1
timeout -k 10 300 python3 tools/bench_be_gan.py --precision bf16x3 > $O/be_gan_bench.json 2> $O/be_gan.err || exit 1
timeout -k 10 500 python3 tools/bench_font.py --img 256 --batch 64 --precision bf16x3 --steps 3 --warmup 1 > $O/font256_bench.json 2> $O/font256.err || exit 1
echo "benches done"
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_vaegan -o t -- python3 tools/bench_vaegan.py --path fused --steps 10 --warmup 3 --cpu-steps 0 > $O/ks_vaegan.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_be_heads -o t -- python3 tools/bench_be_heads.py --precision bf16x3 --cpu-steps 0 > $O/ks_be_heads.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_be_gan -o t -- python3 tools/bench_be_gan.py --precision bf16x3 --cpu-steps 0 > $O/ks_be_gan.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_font256 -o t -- python3 tools/bench_font.py --img 256 --batch 64 --precision bf16x3 --steps 2 --warmup 1 --cpu-steps 0 > $O/ks_font256.log 2>&1 || exit 1
echo "tables done"
du -sh $O
