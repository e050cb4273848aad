#!/bin/bash
# A/B of library variants on the bench workload, three interleaved rounds: bash scratch/r3_ab.sh lib1.so lib2.so ...  ("default" = in-tree)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for l in "$@"; do
    if [ "$l" = default ]; then python3 scratch/qb.py 2>&1 | tail -1 | sed "s/^/r$round /"; else MEDGP_LIB=/root/repo/scratch/$l python3 scratch/qb.py 2>&1 | tail -1 | sed "s/^/r$round /" | sed 's#/root/repo/scratch/##'; fi
  done
done
