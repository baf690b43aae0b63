cd $GRAFT_REPO_ROOT
bash tools/pmc_raw.sh "${1:-siib_|eigh_}" ${2:-tools/siib_ab.py} ${3:-256} ${4:-63871} > gpurun_out/s3_pmc.txt 2>&1
python tools/pmc_table.py gpurun_out/s3_pmc.txt
