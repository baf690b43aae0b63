cd $GRAFT_REPO_ROOT
run() { python bench.py --batch 32 --metrics "siib&estoi" --steps 16 --warmup 4 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for r in 1 2; do
echo -n "default "; run
echo -n "P4_BATCH=64 "; NELE_EIGH_P4_BATCH=64 run
echo -n "P4_BATCH=16 "; NELE_EIGH_P4_BATCH=16 run
echo -n "SPECW=0 "; NELE_SIIB_SPECW=0 run
echo -n "STFT_WAVE=0 "; NELE_STFT_WAVE=0 run
done
