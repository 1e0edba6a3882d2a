#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun; outputs under gpurun_out/prof, copied into profiles/ by hand):
#   bench line, rocprofv3 kernel tables (serial / concurrent schedule; bf16x3, f16x2 and exact f32), MFMA-busy and HBM-traffic PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants"
P="bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline --no-variants"
timeout -k 10 500 python3 bench.py --tags-out $O/tags_bf16x3.json > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "bench done"
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_serial -o t -- python3 $B > $O/ks_serial.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_conc -o t -- python3 $B > $O/ks_conc.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_x2_serial -o t -- python3 $B --precision f16x2 > $O/ks_x2_serial.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_f32 -o t -- python3 $B --precision f32 --tags-out $O/tags_f32.json > $O/ks_f32.log 2>&1 || exit 1
echo "kernel stats done"
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o t -- python3 $P > $O/mfma.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_x2 -o t -- python3 $P --precision f16x2 > $O/mfma_x2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_f32 -o t -- python3 $P --precision f32 > $O/mfma_f32.log 2>&1 || exit 1
echo "mfma done"
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $P > $O/fetch.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $P > $O/write.log 2>&1 || exit 1
echo "traffic done"
find $O -name "*.csv" | head -40
du -sh $O
# afterwards, in the repository (the CSVs come back under gpurun_out/prof):
#   python profiles/make_summary.py  gpurun_out/prof/ks_serial/t_kernel_stats.csv 222 "<title>" > profiles/rNN_x_summary_serial.md   (also ks_conc, ks_x2_serial, ks_f32)
#   python profiles/make_mfma_busy.py gpurun_out/prof/mfma/t_counter_collection.csv > profiles/rNN_x_mfma_busy.md                     (also mfma_x2, mfma_f32)
#   python profiles/make_traffic.py  gpurun_out/prof/fetch/t_counter_collection.csv gpurun_out/prof/write/t_counter_collection.csv 5 profiles/rNN_x_traffic.json > profiles/rNN_x_traffic.md
#   python profiles/make_roofline.py gpurun_out/prof 222 5 > profiles/rNN_x_roofline_table.md
