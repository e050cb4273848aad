#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_parity2_gpu.py tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -3
for sp in 0 512 256 1024; do
echo "== MEDGP_LA_SPLIT=$sp"
MEDGP_LA_SPLIT=$sp python3 scratch/qt.py 1 4096 64 2>&1 | tail -1 | cut -c1-330
MEDGP_LA_SPLIT=$sp python3 scratch/qt.py 16 2048 24 2>&1 | tail -1 | cut -c1-330
MEDGP_LA_SPLIT=$sp python3 scratch/qt.py 4 2048 24 2>&1 | tail -1 | cut -c1-330
MEDGP_LA_SPLIT=$sp python3 scratch/qt.py 64 512 24 2>&1 | tail -1 | cut -c1-330
done
MEDGP_LA_SPLIT=512 python3 scratch/qt.py 1 2048 24 2>&1 | tail -1 | cut -c1-330
python3 scratch/qt.py 256 256 2 2>&1 | tail -1 | cut -c1-330
