cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py tests/test_step_parity_gpu.py -q -x -m gpu 2>&1 | tail -2
bash tools/prof_one.sh tools/siib_ab.py 256 63871 2>&1 | grep -E "scores|ms per call|siib_lag|cluster4"
for r in 1 2; do for v in 64 0; do if [ $v = 0 ]; then unset NELE_EIGH_P4_BATCH; else export NELE_EIGH_P4_BATCH=$v; fi; echo -n "P4_BATCH=$v "; python bench.py --steps 10 --warmup 3 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
