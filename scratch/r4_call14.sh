#!/bin/bash
# round 4, GPU call 14: final diag16 (whole-row stores only): GPU suite, bits vs the previous build, A/B, profile set, bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c14
O=gpurun_out/r4c14
(time python3 -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
for cfg in "4 512 24 0" "4 512 24 -1" "1 2048 24 0" "3 300 2 -1" "2 130 3 -1" "1 4096 64 0"; do
  set -- $cfg
  if [ "$4" = "0" ]; then unset MEDGP_MULTI_CU; else export MEDGP_MULTI_CU=$4; fi
  python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_prev.so python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), ' status', a['st'].tolist(), b['st'].tolist())"
done > $O/bits.log 2>&1
unset MEDGP_MULTI_CU
cat $O/bits.log
bash scratch/r3_ab.sh default libmedgp_prev.so > $O/ab_headline.log 2>&1; grep -v amdgpu $O/ab_headline.log | cut -c1-260
bash scratch/la_ab.sh default libmedgp_prev.so 2>&1 | grep -v amdgpu | grep "P1 " | sed "s/.*\(default\|libmedgp_prev.so\) \(P1 N[0-9]* D[0-9]*\).*'k_la_step': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_la_step \3 wall \4/" > $O/ab_la.log; cat $O/ab_la.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_pre_pmc.log 2>&1
bash scratch/gpurun_prof.sh r04 2>&1 | tail -2
