#!/bin/bash
# non-periodic length (L = 63 871) against the headline length, kernel by kernel (NELE_SERIAL=1: every kernel alone)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4v; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated"
cd /tmp
for L in 64000 63871; do
NELE_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4v/ser$L -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --length $L > $GRAFT_REPO_ROOT/gpurun_out/r4v/ser$L.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/r4v -name "*kernel_trace.csv" -delete
python tools/kstats.py gpurun_out/r4v/ser64000 4 40 > gpurun_out/r4v/k64000.txt
python tools/kstats.py gpurun_out/r4v/ser63871 4 40 > gpurun_out/r4v/k63871.txt
head -3 gpurun_out/r4v/k64000.txt gpurun_out/r4v/k63871.txt
