#!/bin/bash
# round 4, GPU call 44: PRICING build of a split of k_wgrad's k ranges over 2 / 4 workgroups per tile (results meaningless: every split writes the same slab entries)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in default split2 split4; do
    for shape in "1 2048 24" "1 4096 64" "4 1024 24" "2 1536 24"; do
      if [ $v = default ]; then timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z0-9]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/";
      else LIB=/root/repo/scratch/libmedgp_$v.so timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z0-9]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/"; fi
    done
  done
done
