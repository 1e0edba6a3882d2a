set -o pipefail
O=gpurun_out/r3i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; python -c "
import json; d=json.load(open('$O/bench_full.json')); print('bf16x3', d['ms_per_step'], d['value']); print('exact_f32', d.get('exact_f32',{}).get('ms_per_step'))
for v in d['variants']: print(v['workload'], v['precision'], v.get('ms_per_step'), v.get('images_per_sec'))"
python tools/bench_vaegan.py --path fused --steps 20 --warmup 5 --cpu-steps 0 > $O/vaegan.json 2>$O/vaegan.err; tail -c 400 $O/vaegan.json
