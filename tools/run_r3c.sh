set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/ab_multi.py --rounds 5 --steps 20 VP_WGRAD5=0 VP_WGRAD5_BLOCKS=256 VP_WGRAD5_BLOCKS=192 VP_WGRAD5_BLOCKS=128 VP_WGRAD5_BLOCKS=96 VP_WGRAD5_BLOCKS=64 VP_SIDE_WGRAD=0 VP_SIDE_WGRAD=0,VP_WGRAD5=0 > $O/ab_blocks.log 2>&1; echo "ab rc=$?"; cat $O/ab_blocks.log | tail -12
tools/kbench/wbench B=32 reps=20 clock=1 old=0 layers=dec0,dec1,dec2,enc3 > $O/wbench_m16.log 2>&1; cat $O/wbench_m16.log
python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants --tags-out $O/tags_f32.json > $O/bench_f32.json 2> $O/bench_f32.err; echo "f32 rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_engine_contract.py tests/test_gpu_bench_two_ranks.py tests/test_parallel_gloo.py tests/test_gpu_rccl_one_rank.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_new.log
