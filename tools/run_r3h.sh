set -o pipefail
O=gpurun_out/r3h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/kbench/kbench B=8 img=32 reps=2 mode=all buf=1 m16=2 v1=1 > $O/kbench_small.log 2>&1; cut -c1-330 $O/kbench_small.log | tail -8
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python tools/bench_be_gan.py --precision bf16x3 > $O/be_gan_bench.json 2> $O/be_gan_bench.err; echo "be_gan rc=$?"; cat $O/be_gan_bench.json | cut -c1-900
