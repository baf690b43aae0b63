export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag; mkdir -p $R/gpurun_out/pmc_$tag
  cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-utts 0 2>&1 | tail -1 | cut -c1-80
done
ls $R/gpurun_out/
