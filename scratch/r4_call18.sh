#!/bin/bash
# round 4, GPU call 18: look-ahead schedule with the H role + fast diag16 loads in the four-wave factor + lower-triangle head-start loads:
# bits vs the previous build, A/B (each change alone), stamps, GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c18
O=gpurun_out/r4c18
for cfg in "1 2048 24" "3 700 24" "1 4096 64" "5 200 3" "2 130 2" "16 1024 24"; do
  set -- $cfg
  python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_prev.so python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), ' status', a['st'].tolist(), b['st'].tolist())"
done > $O/bits.log 2>&1
cat $O/bits.log
bash scratch/la_ab.sh default libmedgp_prev.so 2>&1 | grep -v amdgpu | sed "s/.*\(default\|libmedgp_[a-zA-Z0-9]*.so\) \(P[0-9]* N[0-9]* D[0-9]*\).*'k_la_step': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_la_step \3 wall \4/" > $O/ab_la.log; cat $O/ab_la.log
python3 scratch/la_stamps.py 2048 24 > $O/stamps_2048.log 2>&1; grep -E "step  ?(1|4|16):|D phases|F key" $O/stamps_2048.log | sed -n 4,12p | cut -c1-230
(time python3 -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
