#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
for rep in 1 2; do for pre in 0 1; do for B in 32 128; do
  m='siib&estoi'; [ $B = 128 ] && m='siib&haspi&estoi'
  echo "B=$B prefetch=$pre $(NELE_PREFETCH=$pre python bench.py --batch $B --metrics $m --steps 12 --warmup 3 --cpu-utts 0 --companions 0 --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],3))")"
done; done; done | tee gpurun_out/r4s/prefetch.txt
