set -o pipefail
O=gpurun_out/r3g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants"
python $F --tags-out $O/tags_f32.json > $O/f32_side1.json 2> $O/f32.err; VP_F32_SIDE=0 python $F > $O/f32_side0.json 2>> $O/f32.err; VP_ADAM_OUTER_EARLY=0 python $F > $O/f32_side1_early0.json 2>> $O/f32.err
VP_WGRAD_SIDE_CUS=128 python $F > $O/f32_side1_c128.json 2>> $O/f32.err; VP_WGRAD_SIDE_CUS=192 python $F > $O/f32_side1_c192.json 2>> $O/f32.err
for f in $O/f32_*.json; do python -c "import json; d=json.load(open('$f')); print('$f', d['ms_per_step'])"; done
python tools/ab_multi.py --rounds 5 --steps 20 VP_IGEMM16P_M16=1 VP_IGEMM16P_M16=0 > $O/ab_m16.log 2>&1; tail -2 $O/ab_m16.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
