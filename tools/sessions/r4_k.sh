#!/bin/bash
# round 4, final measurements on the current tree: the whole -m gpu suite, tools/prof_round.sh r04 (kernel stats multi-stream / serial + three
# PMC passes), tools/make_traffic_json.py, the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/prof_round.sh r04 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
python tools/make_traffic_json.py gpurun_out/r04_prof 256 64000 'siib&haspi&estoi' bf16 r04 2>&1 | grep -E "eigh|wgrad_dma|conv16_kernel<4" 
mkdir -p gpurun_out/r04_prof/out; cp profiles/r04/traffic.json gpurun_out/r04_prof/out/
python tools/kstats.py gpurun_out/r04_prof/stats 4 8
python tools/kstats.py gpurun_out/r04_prof/stats_serial 4 16
python bench.py > gpurun_out/r04_prof/out/bench_default.json 2> gpurun_out/r04_prof/out/bench_default.err; tail -c 300 gpurun_out/r04_prof/out/bench_default.json
# the eigensolver alone (256 matrices of order 420 in one call: the figure the r03 verdict's "<= 7 ms" is about)
cd /tmp; rm -rf /tmp/pe; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py 256 420 3 > /tmp/pe.log 2>&1
cd $GRAFT_REPO_ROOT; cp /tmp/pe/*/*_kernel_stats.csv gpurun_out/r04_prof/out/eigh_standalone_kernel_stats.csv; tail -1 /tmp/pe.log | tee gpurun_out/r04_prof/out/eigh_standalone.txt
timeout 100 python tools/eigh_time.py 256 420 5 2>&1 | tail -1 | tee -a gpurun_out/r04_prof/out/eigh_standalone.txt
