cd $GRAFT_REPO_ROOT
export NELE_SERIAL=1
bash tools/pmc_raw.sh "_kernel" bench.py --steps 2 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated > gpurun_out/s3_pmc_step.txt 2>&1
python tools/pmc_table.py gpurun_out/s3_pmc_step.txt | head -70
