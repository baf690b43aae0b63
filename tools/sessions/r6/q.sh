#!/bin/bash
# r6 call q: float64 instruction counts of the HASPI filter-bank kernels (are they bound by float64 issue, as DESIGN says?)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6q; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*F64[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/avail.txt; cat $O/avail.txt; echo
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d $O/f64 -- python3 $R/tools/haspi_ab.py 256 > $O/f64.log 2>&1
tail -2 $O/f64.log
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/valu -- python3 $R/tools/haspi_ab.py 256 > $O/valu.log 2>&1
ls $O/f64/*/ | head
