#!/bin/bash
cd $GRAFT_REPO_ROOT
PMC=0 bash tools/prof_round.sh r05 2>&1 | tail -4
python tools/kstats.py gpurun_out/r05_prof/stats_serial 4 70
