cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py tests/test_step_parity_gpu.py -q -x -m gpu 2>&1 | tail -2
bash tools/prof_one.sh tools/siib_ab.py 256 63871 2>&1 | grep -E "scores|ms per call|invit"
