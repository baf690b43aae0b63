cd $GRAFT_REPO_ROOT
python -m pytest tests/test_features_gpu.py -q -x -m gpu 2>&1 | tail -2
for w in 1 0; do echo "NELE_STFT_WAVE=$w"; NELE_STFT_WAVE=$w bash tools/prof_one.sh tools/feat_check.py 2>&1 | grep -E "istft|stft_band|features"; done
