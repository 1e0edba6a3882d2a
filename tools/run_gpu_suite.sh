#!/bin/bash
# What the driver runs at round end, in one gpurun call: the GPU tests, smoke() and the default bench line.
# usage (from the container): gpurun --timeout 1200 -- 'bash tools/run_gpu_suite.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/suite
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1
rc=$?
tail -n 6 $O/pytest.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2> $O/smoke.err || exit 1
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms; exact_f32", d.get("exact_f32", {}).get("ms_per_step"))
PY
