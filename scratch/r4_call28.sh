#!/bin/bash
# round 4, GPU call 28: k_wgrad with two-chunk prefetch + serpentine tile order on launches that do not fill the chip twice: bits + A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c28; mkdir -p $O
export LIB=/root/repo/scratch/libmedgp_deep.so
for cfg in "1 2048 24" "3 700 24" "2 1100 8" "1 300 3"; do
  set -- $cfg
  MEDGP_WGRAD_DEEP=0 timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  MEDGP_WGRAD_DEEP=1 timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'status', a['st'][:3], b['st'][:3])"
done
for round in 1 2 3; do
  for deep in 0 1; do
    for shape in "1 2048 24" "1 4096 64" "16 2048 24" "4 1024 24" "32 512 24" "8 256 2"; do
      MEDGP_WGRAD_DEEP=$deep timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round deep=$deep /" | sed "s/.*\(r[0-9] deep=[01]\).*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/"
    done
  done
done
