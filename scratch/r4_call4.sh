#!/bin/bash
# round 4, GPU call 4: two cheap experiments on the headline step: (a) MEDGP_STREAMS=2 (two half batches on two streams: one half's
# k_cholinv beside the other half's pair kernels), (b) k_cholinv<4,4> with the 512-register budget of one workgroup per CU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c4
O=gpurun_out/r4c4
for r in 1 2 3; do
  python3 bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default  ', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
  MEDGP_STREAMS=2 python3 bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=2', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done > $O/streams.log 2>&1
for r in 1 2 3; do
  MEDGP_MULTI_CU=-1 MEDGP_CHOLINV_NW=44 python3 scratch/qt.py 256 512 24 2>/dev/null | tail -1 | cut -c1-260
  MEDGP_MULTI_CU=-1 MEDGP_CHOLINV_NW=44 LIB=/root/repo/scratch/libmedgp_wps1.so python3 scratch/qt.py 256 512 24 2>/dev/null | tail -1 | cut -c1-260
done > $O/wps1.log 2>&1
cat $O/streams.log $O/wps1.log
