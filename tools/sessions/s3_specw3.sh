cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py tests/test_varlen_gpu.py tests/test_step_parity_gpu.py -q -x -m gpu 2>&1 | tail -3
bash tools/prof_one.sh tools/siib_ab.py 256 63871 2>&1 | grep -E "scores|ms per call|siib_spec|siib_db"
for i in 1 2; do python bench.py --length 63871 --steps 8 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('np', d['ms_per_step'])"; python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p', d['ms_per_step'])"; done
