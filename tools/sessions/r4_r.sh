#!/bin/bash
# configs[1] (B = 32, SIIB + ESTOI): event timeline, host enqueue time, kernel sums (multi-stream and serial)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4r; export TMPDIR=/tmp
CH_B=32 CH_METRICS='siib&estoi' python tools/chains.py 2>&1 | tail -12 | tee gpurun_out/r4r/chains32.txt
python tools/host_time.py 32 2>&1 | head -50 | tee gpurun_out/r4r/host32.txt
ARGS="--batch 32 --metrics siib&estoi --steps 7 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4r/ms -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/r4r/ms.log 2>&1
NELE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4r/ser -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/r4r/ser.log 2>&1
cd $GRAFT_REPO_ROOT
tail -c 400 gpurun_out/r4r/ms.log; echo
python tools/kstats.py gpurun_out/r4r/ms 8 6
python tools/kstats.py gpurun_out/r4r/ser 8 45
