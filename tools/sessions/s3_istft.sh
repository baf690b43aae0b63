cd $GRAFT_REPO_ROOT
python -m pytest tests/test_features_gpu.py tests/test_varlen_gpu.py -q -x -m gpu 2>&1 | tail -2
bash tools/prof_one.sh tools/feat_check.py 2>&1 | grep -E "istft|stft_band|features|band_from|imcra"
