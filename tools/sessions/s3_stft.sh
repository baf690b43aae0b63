cd $GRAFT_REPO_ROOT
python -m pytest tests/test_features_gpu.py tests/test_varlen_gpu.py -q -x -m gpu 2>&1 | tail -5
for w in 1 0 1 0; do echo "NELE_STFT_WAVE=$w"; NELE_STFT_WAVE=$w python bench.py --steps 10 --warmup 3 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
bash tools/prof_one.sh tools/feat_check.py 2>&1 | tail -16
