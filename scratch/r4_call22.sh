#!/bin/bash
# round 4, GPU call 22: two-wave diag16 on the look-ahead chain: bits + A/B + stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c22
O=gpurun_out/r4c22
for cfg in "1 2048 24" "3 700 24" "2 1100 8"; do
  set -- $cfg
  timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_pair.so timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'status', a['st'][:3], b['st'][:3])"
done
timeout 900 bash scratch/la_ab.sh default libmedgp_pair.so 2>&1 | grep -v amdgpu | sed "s/.*\(default\|libmedgp_[a-zA-Z0-9]*.so\) \(P[0-9]* N[0-9]* D[0-9]*\).*'k_la_step': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_la_step \3 wall \4/"
LASTAMP_LIB=/root/repo/scratch/lib_lastamps_pair.so timeout 300 python3 scratch/la_stamps.py 2>&1 | tail -40
