#!/bin/bash
# VAE-GAN fused-step measurement set (run on the GPU box through gpurun): golden-vector error diagnostics, rocprofv3 kernel tables
# of tools/bench_vaegan.py --path fused on the serial and the concurrent schedule, A/B of the side stream and the statistics epilogue.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/vgf; mkdir -p $O
timeout -k 10 200 python tests/diag/gan_golden_errors.py > $O/golden_err.log 2>&1; tail -40 $O/golden_err.log
B="tools/bench_vaegan.py --path fused --steps 20 --warmup 3 --cpu-steps 0"
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_serial -o t -- python3 $B > $O/ks_serial.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_conc -o t -- python3 $B > $O/ks_conc.log 2>&1
tail -2 $O/ks_serial.log $O/ks_conc.log
VP_SIDE_WGRAD=0 timeout -k 10 100 python3 $B 2>/dev/null
VP_FUSE_BN_STATS=0 timeout -k 10 100 python3 $B 2>/dev/null
