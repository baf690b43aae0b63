cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 64 32 48; do echo -n "P4_BATCH=$v "; NELE_EIGH_P4_BATCH=$v python bench.py --steps 10 --warmup 3 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done; done
