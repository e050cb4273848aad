#!/bin/bash
# round 4, GPU call 36: eight-wave k_wgrad on few-patient launches: bits vs the shipped build + A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c36; mkdir -p $O
for cfg in "1 2048 24" "3 700 24" "2 1100 8" "1 4096 64" "4 1024 24" "9 130 3" "1 70 2"; do
  set -- $cfg
  timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_nw8.so MEDGP_WGRAD_NW8=1 timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'status', a['st'][:3], b['st'][:3])"
done
export LIB=/root/repo/scratch/libmedgp_nw8.so
for round in 1 2 3; do
  for v in 0 1; do
    for shape in "1 2048 24" "1 4096 64" "4 1024 24" "8 768 24" "2 1536 24" "16 2048 24" "32 512 24"; do
      MEDGP_WGRAD_NW8=$v MEDGP_WGRAD_DEEP=2 timeout 300 python3 scratch/qt.py $shape 2>&1 | tail -1 | sed "s/^/r$round nw8=$v /" | sed "s/.*\(r[0-9] nw8=[01]\) .*\(P[0-9]* N[0-9]* D[0-9]*\).*'k_wgrad': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_wgrad \3 wall \4/"
    done
  done
done
