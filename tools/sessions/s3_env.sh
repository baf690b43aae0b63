cd $GRAFT_REPO_ROOT
run32() { python bench.py --batch 32 --metrics "siib&estoi" --steps 16 --warmup 4 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run256() { python bench.py --steps 8 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
echo -n "default B32 "; run32
echo -n "DEV_KERNARG=1 B32 "; HIP_FORCE_DEV_KERNARG=1 run32
echo -n "DEV_KERNARG=0 B32 "; HIP_FORCE_DEV_KERNARG=0 run32
done
echo -n "default B256 "; run256
echo -n "DEV_KERNARG=1 B256 "; HIP_FORCE_DEV_KERNARG=1 run256
