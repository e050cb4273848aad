#!/bin/bash
# round 4, GPU call 11: diag16 with branch-free loads / whole-row stores / LDS-broadcast scaling: micro-benchmark, bits vs the previous build, A/B on the headline and look-ahead shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c11
O=gpurun_out/r4c11
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 scratch/diag_bench.hip -o /tmp/diag_bench_new 2>/dev/null && /tmp/diag_bench_new > $O/diag_bench.log 2>&1
cat $O/diag_bench.log
for cfg in "4 512 24 0" "4 512 24 -1" "1 2048 24 0" "3 300 2 -1" "2 130 3 -1"; do
  set -- $cfg
  if [ "$4" = "0" ]; then unset MEDGP_MULTI_CU; else export MEDGP_MULTI_CU=$4; fi
  python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  LIB=/root/repo/scratch/libmedgp_d16.so python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg : nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), ' status', a['st'].tolist(), b['st'].tolist())"
done > $O/bits.log 2>&1
unset MEDGP_MULTI_CU
cat $O/bits.log
bash scratch/r3_ab.sh default libmedgp_d16.so > $O/ab_headline.log 2>&1; grep -v amdgpu $O/ab_headline.log | cut -c1-260
bash scratch/la_ab.sh default libmedgp_d16.so > $O/ab_la.log 2>&1; grep -v amdgpu $O/ab_la.log | cut -c1-300
for r in 1 2; do python3 scratch/qt.py 256 256 2 | tail -1 | cut -c1-250; LIB=/root/repo/scratch/libmedgp_d16.so python3 scratch/qt.py 256 256 2 | tail -1 | cut -c1-250; done
