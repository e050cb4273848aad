#!/bin/bash
# round 4, GPU call 24: diag16 chain variants (pre-broadcast, writelane/LDS finish, cubic reciprocal step): standalone + in situ
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c24; mkdir -p $O
for b in scratch/diag_bench_DD16_PREBCAST0DD16_FIN0 scratch/diag_bench_DD16_PREBCAST1DD16_FIN0 scratch/diag_bench_DD16_PREBCAST0DD16_FIN1 scratch/diag_bench_DD16_PREBCAST1DD16_FIN1 scratch/diag_bench_DD16_PREBCAST1DD16_FIN1DD16_HALLEY1; do echo "== $b"; timeout 60 $b; done
for cfg in "1 2048 24" "64 512 24"; do
  set -- $cfg
  timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/a.npz > /dev/null 2>&1
  for l in p1f0 p0f1 p1f1 p1f1h; do
  LIB=/root/repo/scratch/libmedgp_$l.so timeout 300 python3 scratch/dump_eval.py $1 $2 $3 $O/b.npz > /dev/null 2>&1
  python3 -c "
import numpy as np
a=np.load('$O/a.npz'); b=np.load('$O/b.npz')
print('shape $cfg $l: nlml identical', np.array_equal(a['nl'],b['nl']), ' grad identical', np.array_equal(a['g'],b['g']), 'max rel nlml', np.max(np.abs(a['nl']-b['nl'])/np.abs(a['nl'])), 'grad', np.max(np.abs(a['g']-b['g']))/np.max(np.abs(a['g'])))"
  done
done
timeout 1200 bash scratch/la_ab.sh default libmedgp_p1f0.so libmedgp_p0f1.so libmedgp_p1f1.so libmedgp_p1f1h.so 2>&1 | grep -v amdgpu | sed "s/.*\(default\|libmedgp_[a-zA-Z0-9]*.so\) \(P[0-9]* N[0-9]* D[0-9]*\).*'k_la_step': \([0-9.]*\).*wall_ms_per_call \([0-9.]*\)/\1 \2 k_la_step \3 wall \4/"
timeout 1200 bash scratch/r3_ab.sh default libmedgp_p1f0.so libmedgp_p0f1.so libmedgp_p1f1.so libmedgp_p1f1h.so 2>&1 | grep -v amdgpu | cut -c1-300
