#!/bin/bash
# A/B of library variants on the look-ahead shapes (configs 3 and 5, 16 x N=2048), interleaved rounds:
#   bash scratch/la_ab.sh lib1.so lib2.so ...   ("default" = in-tree); LIB= is read by scratch/qt.py
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for l in "$@"; do
    for shape in "1 2048 24" "1 4096 64" "16 2048 24"; do
      if [ "$l" = default ]; then python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round /" | cut -c1-330; else LIB=/root/repo/scratch/$l python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round /" | sed 's#/root/repo/scratch/##' | cut -c1-330; fi
    done
  done
done
