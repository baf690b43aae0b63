#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r04 2>&1 | tail -15
cd $GRAFT_REPO_ROOT
python tools/make_traffic_json.py gpurun_out/r04_prof 256 64000 'siib&haspi&estoi' bf16 r04 2>&1 | tail -70
mkdir -p gpurun_out/r04_prof/out; cp profiles/r04/traffic.json gpurun_out/r04_prof/out/
python tools/kstats.py gpurun_out/r04_prof/stats 4 12
python tools/kstats.py gpurun_out/r04_prof/stats_serial 4 12
python bench.py > gpurun_out/r04_prof/out/bench_default.json 2> gpurun_out/r04_prof/out/bench_default.err; tail -c 600 gpurun_out/r04_prof/out/bench_default.json
