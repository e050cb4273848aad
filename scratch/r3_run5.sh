#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 scratch/qb.py 2>&1 | tail -1
python3 scratch/qb.py 2>&1 | tail -1
python3 scratch/qt.py 256 256 2 2>&1 | tail -1
timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -5
python3 scratch/qb.py 2>&1 | tail -1
