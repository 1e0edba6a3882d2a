set -o pipefail
O=gpurun_out/r3f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants"
P="bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline --no-variants"
python $B --tags-out $O/tags.json > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bf16x3 ms', d['ms_per_step'], d['roofline'])"
timeout -k 10 300 python -m pytest tests/test_gpu_capture_guard.py tests/test_gpu_engine.py tests/test_gpu_engine_gan.py tests/test_gpu_f16x2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_serial -o t -- python3 $B > $O/ks_serial.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o t -- python3 $P > $O/mfma.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $P > $O/fetch.log 2>&1 || exit 1
VP_SIDE_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $P > $O/write.log 2>&1 || exit 1
python tools/ab_multi.py --rounds 4 --steps 20 --gan VP_WGRAD_SIDE_CUS=128 VP_WGRAD_SIDE_CUS=160 VP_WGRAD_SIDE_CUS=144 > $O/ab_gan.log 2>&1; tail -3 $O/ab_gan.log
python tools/ab_multi.py --rounds 4 --steps 20 VP_WGRAD_SIDE_CUS=160 VP_WGRAD_SIDE_CUS=152 VP_WGRAD_SIDE_CUS=168 > $O/ab_vae.log 2>&1; tail -3 $O/ab_vae.log
du -sh $O
