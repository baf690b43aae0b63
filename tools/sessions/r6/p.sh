#!/bin/bash
# r6 final measurement pass: whole -m gpu suite, profile passes of the default bench workload and of the enhancement path (kernel stats +
# PMC), their summaries, the stand-alone eigensolver, host <-> device copy rates, and the default bench line.  -> gpurun_out/, copied to profiles/r06/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06_gputest.txt
bash tools/prof_round.sh r06 2>&1 | tail -3
bash tools/prof_infer.sh r06 2>&1 | tail -6
cd $GRAFT_REPO_ROOT
python tools/make_traffic_json.py gpurun_out/r06_prof 256 64000 'siib&haspi&estoi' bf16 r06 2>&1 | tail -2
python tools/make_infer_traffic_json.py gpurun_out/r06_infer r06 2>&1 | tail -2
cp profiles/r06/traffic.json profiles/r06/infer_traffic.json gpurun_out/ 2>/dev/null
python tools/eigh_time.py 256 420 5 2>&1 | tail -1 > gpurun_out/r06_eigh_standalone.txt; cat gpurun_out/r06_eigh_standalone.txt
python tools/pcie_time.py 2>&1 | grep -v amdgpu > gpurun_out/r06_pcie.txt; cat gpurun_out/r06_pcie.txt
timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -2 gpurun_out/r06_bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/r06_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'))
"
