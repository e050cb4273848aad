#!/bin/bash
# round 4, GPU call 40: one result copy instead of three on small host-pointer calls: wall per call A/B (prev = before), then the GPU suite
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prevlib; 
for round in 1 2 3; do
  for v in prev new; do
    for shape in "256 256 2" "1 2048 24" "8 64 2" "32 512 24"; do
      if [ $v = new ]; then timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z0-9]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'sum' \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 wall \4/" | sed "s/.*\(r[0-9] [a-z]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).* sum \([0-9.]*\) wall_ms_per_call \([0-9.]*\)/\1 \2 kernels \3 wall \4/";
      else LIB=/root/repo/scratch/libmedgp_prev.so timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).* sum \([0-9.]*\) wall_ms_per_call \([0-9.]*\)/\1 \2 kernels \3 wall \4/"; fi
    done
  done
done
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
