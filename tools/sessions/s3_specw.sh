cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py -q -x -m gpu -k "siib" 2>&1 | tail -5
for w in 1 0; do echo "NELE_SIIB_SPECW=$w"; NELE_SIIB_SPECW=$w bash tools/prof_one.sh tools/siib_ab.py 256 63871 2>&1 | grep -E "scores|ms per call|siib_spec|siib_db|siib_mask"; done
for w in 1 0 1 0; do echo "NELE_SIIB_SPECW=$w"; NELE_SIIB_SPECW=$w python bench.py --length 63871 --steps 8 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
for w in 1 0; do echo "NELE_SIIB_SPECW=$w L=64000"; NELE_SIIB_SPECW=$w python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
