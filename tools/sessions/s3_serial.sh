cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py -q -x -m gpu -k "estoi" 2>&1 | tail -5
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/s3_serial; rm -rf $O; mkdir -p $O; cd /tmp
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT; find $O -name "*kernel_trace.csv" -size +20M -delete
python tools/kstats.py gpurun_out/s3_serial 4 60
