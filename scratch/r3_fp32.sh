#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
python3 scratch/screen_time.py 2>&1 | tail -1
MEDGP_LIB=/root/repo/scratch/libx_fp32emul.so python3 scratch/screen_time.py 2>&1 | tail -1
done
python3 scratch/qb.py 2>&1 | tail -1
MEDGP_LIB=/root/repo/scratch/libx_fp32emul.so python3 scratch/qb.py 2>&1 | tail -1
