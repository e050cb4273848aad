#!/bin/bash
# round 4, GPU call 13: which of the three diag16 changes (loads / stores / scaling) slows k_cholinv in situ
cd $GRAFT_REPO_ROOT
for r in 1 2; do
 for v in v000 v100 v010 v001 v110 d16; do
  L=/root/repo/scratch/libmedgp_$v.so
  a=$(LIB=$L python3 scratch/qt.py 256 256 2 2>/dev/null | tail -1 | sed "s/.*'k_cholinv': \([0-9.]*\).*/\1/")
  b=$(MEDGP_LIB=$L python3 scratch/qb.py 2>/dev/null | tail -1 | sed "s/.*'k_cholinv': \([0-9.]*\).*/\1/")
  c=$(LIB=$L python3 scratch/qt.py 1 2048 24 2>/dev/null | tail -1 | sed "s/.*'k_la_step': \([0-9.]*\).*/\1/")
  echo "r$r $v  config2 k_cholinv $a   headline k_cholinv $b   N2048 k_la_step $c"
 done
done
