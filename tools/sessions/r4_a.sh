#!/bin/bash
# round 4, first GPU call: the new -m gpu tests, the eigensolver alone, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
python -m pytest tests/test_train_gpu.py tests/test_epoch_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4a/tests.txt
python tools/eigh_time.py 256 420 5 > gpurun_out/r4a/eigh.txt 2>&1
python tools/eigh_time.py 64 420 5 >> gpurun_out/r4a/eigh.txt 2>&1
python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -5 gpurun_out/r4a/bench.err
cat gpurun_out/r4a/tests.txt gpurun_out/r4a/eigh.txt
