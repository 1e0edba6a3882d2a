set -o pipefail
O=gpurun_out/r3j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python tools/bench_be_heads.py --precision bf16x3 > $O/be_heads.json 2>$O/be_heads.err; echo "heads rc=$?"; python -c "import json; d=json.load(open('$O/be_heads.json')); print(d['ms_per_step'], json.dumps(d['roofline'])[:600])"
python tools/bench_be_gan.py --precision bf16x3 > $O/be_gan.json 2>$O/be_gan.err; echo "gan rc=$?"; python -c "import json; d=json.load(open('$O/be_gan.json')); print(d['ms_per_step'], json.dumps(d['roofline'])[:600])"
python tools/bench_font.py --img 256 --batch 64 --precision bf16x3 --steps 3 --warmup 1 --cpu-steps 0 > $O/font256.json 2>$O/font256.err; echo "font rc=$?"; python -c "import json; d=json.load(open('$O/font256.json')); print(d['ms_per_step'], json.dumps(d['roofline'])[:900])"
