#!/bin/bash
# round-3 first measurements: headline kernel times, config-2 workgroup-shape sweep
cd $GRAFT_REPO_ROOT
python3 scratch/qb.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -1
for nw in 84 44 42 82; do
  MEDGP_CHOLINV_NW=$nw python3 scratch/qt.py 256 256 2 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -1 | sed "s/^/NW=$nw /"
done
MEDGP_MULTI_CU=1 python3 scratch/qt.py 256 256 2 2>&1 | tail -1 | sed "s/^/MC=1 /"
for nw in 84 44 42 82; do
  MEDGP_CHOLINV_NW=$nw python3 scratch/qt.py 512 256 2 2>&1 | tail -1 | sed "s/^/P512 NW=$nw /"
done
python3 scratch/qt.py 1 2048 24 2>&1 | tail -1
python3 scratch/qt.py 1 4096 64 2>&1 | tail -1
