#!/bin/bash
# HBM-side traffic of the weight-gradient kernels at the CU budget they run with in the concurrent step (160), serial schedule
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof160
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
P="bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline --no-variants"
VP_SIDE_WGRAD=0 VP_WGRAD_MAIN_CUS=160 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $P > $O/fetch.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 VP_WGRAD_MAIN_CUS=160 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $P > $O/write.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 VP_WGRAD_MAIN_CUS=160 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/ks.log 2>&1 || exit 1
echo done
