set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants"
VP_WGRAD5=0 python $B > $O/bench_w5_0.json 2> $O/bench_w5_0.err && python $B --tags-out $O/tags_w5_1.json > $O/bench_w5_1.json 2> $O/bench_w5_1.err && VP_WGRAD5=0 python $B > $O/bench_w5_0b.json 2>> $O/bench_w5_0.err && python $B > $O/bench_w5_1b.json 2>> $O/bench_w5_1.err && python $B --precision f16x2 > $O/bench_f16_w5_1.json 2>> $O/bench_w5_1.err && VP_WGRAD5=0 python $B --precision f16x2 > $O/bench_f16_w5_0.json 2>> $O/bench_w5_0.err || exit 1
for f in $O/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
echo "== tests"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
echo "== f32 profile"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_f32 -o t -- python3 bench.py --steps 10 --warmup 3 --no-settle --no-cpu-baseline --no-variants --precision f32 > $O/ks_f32.log 2>&1; echo "rocprof rc=$?"
ls $O/ks_f32/* | head
