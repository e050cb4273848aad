#!/bin/bash
# round 4, GPU call 19: unrolled store tail of the chain: A/B + bits
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c19
O=gpurun_out/r4c19
for cfg in "1 2048 24" "3 700 24"; do
  set -- $cfg
  python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_unr.so python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']))"
done
bash scratch/la_ab.sh default libmedgp_unr.so 2>&1 | grep -v amdgpu | sed "s/.*\(default\|libmedgp_[a-zA-Z0-9]*.so\) \(P[0-9]* N[0-9]* D[0-9]*\).*'k_la_step': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_la_step \3 wall \4/"
