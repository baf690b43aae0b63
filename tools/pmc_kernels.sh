#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmck_$tag; mkdir -p $R/gpurun_out/pmck_$tag
  cd /tmp && NELE_SERIAL=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-utts 0 --companions 0 > /dev/null 2>&1
done
cd $R; python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmck_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if any(t in n for t in ("conv1d_tile16","conv_tile16_kernel","conv_span16","conv_wgrad_tile16","conv_wgrad16","siib_proj_kernel<2>","haspi_ihc_fir","haspi_bank_scan_kernel<true, true, false, true>","haspi_mod_slide_kernel<1>","eigh_invit")):
            dur=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
            key=n.split("(")[0].replace("void ","")
            if "conv1d" in n: key+=("_big" if dur>60e3 else "_small")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"])); acc[key]["_dur_us"].append(dur/1e3)
for k,c in sorted(acc.items()):
    a={n: sum(v)/len(v) for n,v in c.items()}
    clk=a["_dur_us"]*2400.0
    print("%-52s %8.1f us  mfma %.3f  lds_busy %.2f  conflict/lds %.2f  active_inst/wavecyc %.2f  wait_lds/wavecyc %.2f  wait_any/wavecyc %.2f" % (k[:52], a["_dur_us"], a.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(1024*clk), 4*a.get("SQ_LDS_IDX_ACTIVE",0)/(256*clk), a.get("SQ_LDS_BANK_CONFLICT",0)/max(a.get("SQ_LDS_IDX_ACTIVE",1),1), a.get("SQ_ACTIVE_INST_ANY",0)/max(a.get("SQ_WAVE_CYCLES",1),1), a.get("SQ_WAIT_INST_LDS",0)/max(a.get("SQ_WAVE_CYCLES",1),1), a.get("SQ_WAIT_ANY",0)/max(a.get("SQ_WAVE_CYCLES",1),1)))
PY
