#!/bin/bash
# PMC summary of the kernels of ONE stand-alone python command (each kernel alone on the GPU):
#   tools/pmc_cmd.sh "<kernel-name substrings, |-separated>" tools/conv16_check.py 256 5f
# MFMA-busy, LDS-busy, bank-conflict share, wait fractions, and the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; PAT=$1; shift
for set in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcc_$tag; mkdir -p $R/gpurun_out/pmcc_$tag
  (cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_$tag -- python3 $R/"$@" > $R/gpurun_out/pmcc_$tag/log.txt 2>&1)
done
cd $R; PAT="$PAT" python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
pats = os.environ['PAT'].split('|')
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmcc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(t in n for t in pats):
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            key = n.split("(")[0].replace("void ", "")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["_dur_us:" + r["Counter_Name"]].append(dur / 1e3)
for k, c in sorted(acc.items()):
    a = {n: sorted(v)[len(v) // 2] for n, v in c.items()}          # medians over the launches
    d = lambda n: a.get("_dur_us:" + n, 0.0)
    ghz = a.get("GRBM_GUI_ACTIVE", 0) / 8.0 / max(d("GRBM_GUI_ACTIVE"), 1e-9) / 1e3
    clk = d("SQ_VALU_MFMA_BUSY_CYCLES") * 1e3 * (ghz if ghz > 0 else 2.4)          # cycles of the launch at the clock it held
    clk24 = d("SQ_VALU_MFMA_BUSY_CYCLES") * 2400.0
    lclk = d("SQ_LDS_IDX_ACTIVE") * 1e3 * (ghz if ghz > 0 else 2.4)
    print("%-44s %8.1f us  clock %.2f GHz  mfma_busy %.3f (at 2.4 GHz: %.3f)  lds_busy %.2f  conflict/lds %.2f  active/wave %.2f  wait_lds/wave %.2f  wait_any/wave %.2f  HBM rd %.1f MB wr %.1f MB" % (
        k[:44], d("SQ_VALU_MFMA_BUSY_CYCLES"), ghz, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * max(clk, 1)), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * max(clk24, 1)),
        4 * a.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * max(lclk, 1)), a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        a.get("SQ_ACTIVE_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), a.get("SQ_WAIT_INST_LDS", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1),
        a.get("SQ_WAIT_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), a.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6, a.get("WRITE_SIZE", 0) * 1024 / 1e6))
PY
