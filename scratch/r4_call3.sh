#!/bin/bash
# round 4, GPU call 3: early start of the chain's first tile (LA_D_EARLY): bits vs the previous schedule, A/B timing, stamps, tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c3
O=gpurun_out/r4c3
for shp in "1 2048 24" "3 700 24" "1 4096 64"; do
  set -- $shp
  python3 scratch/dump_eval.py $1 $2 $3 $O/a_$2.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_noearly.so python3 scratch/dump_eval.py $1 $2 $3 $O/b_$2.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a_$2.npz'); b=np.load('$O/b_$2.npz')
print('shape $shp: bit-identical nlml', np.array_equal(a['nl'],b['nl']), 'grad', np.array_equal(a['g'],b['g']), 'status', a['st'], b['st'], 'max rel grad diff', float(np.max(np.abs(a['g']-b['g'])/np.maximum(1e-300,np.abs(b['g']).max())))"
done > $O/bits.log 2>&1
bash scratch/la_ab.sh default libmedgp_noearly.so > $O/ab.log 2>&1
python3 scratch/la_stamps.py 2048 24 > $O/stamps_2048.log 2>&1
python3 scratch/la_stamps.py 4096 64 > $O/stamps_4096.log 2>&1
(time python3 -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
cat $O/bits.log; grep -v amdgpu $O/ab.log | cut -c1-330; grep -E "step  ?(1|4|16):|step  ?(4|16) factor|D phases" $O/stamps_2048.log | sed -n 1,12p | cut -c1-200; tail -4 $O/pytest.log
