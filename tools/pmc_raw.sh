#!/bin/bash
# Raw PMC medians of the kernels of ONE stand-alone python command, one rocprofv3 pass per counter group:
#   COUNTERS="SQ_INSTS_VALU SQ_WAVE_CYCLES;TCP_TCC_READ_REQ_sum" tools/pmc_raw.sh "<kernel substrings |-separated>" tools/haspi_ab.py 256
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; PAT=$1; shift
IFS=';' read -ra SETS <<< "${COUNTERS:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES;SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM;SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum;TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum;GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE}"
i=0
for set in "${SETS[@]}"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcr_$i; mkdir -p $R/gpurun_out/pmcr_$i
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcr_$i -- python3 $R/"$@" > $R/gpurun_out/pmcr_$i/log.txt 2>&1)
  grep -i -m2 "error\|invalid\|not found" $R/gpurun_out/pmcr_$i/log.txt
done
cd $R; PAT="$PAT" python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
pats = os.environ['PAT'].split('|')
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmcr_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(t in n for t in pats):
            key = n.split("(")[0].replace("void ", "")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, c in sorted(acc.items()):
    print(k)
    for n, v in sorted(c.items()):
        v = sorted(v)
        print("   %-36s median %14.1f  (n=%d)" % (n, v[len(v) // 2], len(v)))
PY
