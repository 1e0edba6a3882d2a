#!/bin/bash
# A/B of the tap-pair weight gradient (VP_WGRAD_PAIR=0/1) on the headline step and the fused VAE-GAN step; run through gpurun.
O=gpurun_out/pair; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_vaegan.py tests/test_gpu_engine_gan.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
grep -q " passed" $O/tests.log && ! grep -q failed $O/tests.log || exit 1
for v in 0 1 0 1; do
  VP_WGRAD_PAIR=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline pair=$v', d['ms_per_step'])"
  VP_WGRAD_PAIR=$v timeout -k 10 200 python tools/bench_vaegan.py --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vaegan   pair=$v', d['ms_per_step'])"
done
