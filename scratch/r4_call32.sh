#!/bin/bash
# round 4, GPU call 32: final k_wgrad (prefetch depth 2 + serpentine order on few-patient launches): bits vs the previous build, full GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c32; mkdir -p $O
for cfg in "1 2048 24" "3 700 24" "2 1100 8" "64 512 24" "1 4096 64" "4 1024 24" "9 130 3"; do
  set -- $cfg
  timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_prev.so timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'status', a['st'][:3], b['st'][:3])"
done
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
