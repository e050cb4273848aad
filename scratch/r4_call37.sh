#!/bin/bash
# round 4, GPU call 37: k_wgrad tile rows pinned to XCDs for <= 8 patients: bits vs the shipped build + A/B (with / without eight waves)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c37; mkdir -p $O
for cfg in "1 2048 24" "3 700 24" "2 1100 8" "1 4096 64" "4 1024 24" "8 768 24" "5 130 3" "1 70 2" "7 300 24"; do
  set -- $cfg
  timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  for nw in 0 1; do
  LIB=/root/repo/scratch/libmedgp_xmap.so MEDGP_WGRAD_NW8=$nw timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg nw8 $nw: nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'status', a['st'][:3], b['st'][:3])"
  done
done
for round in 1 2 3; do
  for v in default x0 x1; do
    for shape in "1 2048 24" "1 4096 64" "4 1024 24" "8 768 24" "2 1536 24" "8 2048 24"; do
      if [ $v = default ]; then timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z0-9]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/";
      else LIB=/root/repo/scratch/libmedgp_xmap.so MEDGP_WGRAD_NW8=${v#x} timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round $v /" | sed "s/.*\(r[0-9] [a-z0-9]*\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/"; fi
    done
  done
done
