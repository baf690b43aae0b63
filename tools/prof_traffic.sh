export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c; mkdir -p $R/gpurun_out/pmc_$c
  cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-utts 0 2>&1 | tail -1 | cut -c1-80
done
