#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -4
bash scratch/r3_ab.sh libmedgp_base.so default
MEDGP_LIB=/root/repo/scratch/libmedgp_base.so python3 scratch/qt.py 256 256 2 2>&1 | tail -1 | cut -c1-300
python3 scratch/qt.py 256 256 2 2>&1 | tail -1 | cut -c1-300
python3 scratch/qt.py 1 2048 24 2>&1 | tail -1 | cut -c1-300
