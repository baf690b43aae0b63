#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/dbg_fused.py
