cd $GRAFT_REPO_ROOT
bash tools/pmc_raw.sh "siib_|eigh_bisect|eigh_wy|estoi_" tools/siib_ab.py 256 63871 > gpurun_out/siib_pmc3.txt 2>&1
python tools/pmc_table.py gpurun_out/siib_pmc3.txt
