set -o pipefail
O=gpurun_out/r3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants --tags-out $O/tags_f32.json > $O/bench_f32.json 2> $O/bench_f32.err; echo "f32 rc=$?"; python -c "import json; d=json.load(open('$O/bench_f32.json')); print('f32 ms', d['ms_per_step'], d['roofline']['kernel'][:30], d['roofline']['frac'])"
VP_F32_FAST=0 VP_WGRAD5F=0 python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/bench_f32_old.json 2>> $O/bench_f32.err; python -c "import json; d=json.load(open('$O/bench_f32_old.json')); print('f32 old ms', d['ms_per_step'])"
VP_WGRAD5F=0 python bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/bench_f32_ft.json 2>> $O/bench_f32.err; python -c "import json; d=json.load(open('$O/bench_f32_ft.json')); print('f32 fast F/T only ms', d['ms_per_step'])"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python bench.py --no-cpu-baseline --no-variants > $O/bench_bf16.json 2>$O/bench_bf16.err; python -c "import json; d=json.load(open('$O/bench_bf16.json')); print('bf16x3 ms', d['ms_per_step'])"
