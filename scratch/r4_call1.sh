#!/bin/bash
# round 4, GPU call 1: sanity (tests, bench), small-N route probes (item 5), look-ahead factor stamps (item 1b)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c1
O=gpurun_out/r4c1
(time python3 -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
python3 bench.py > $O/bench.log 2>&1
for r in 1 2; do
  echo "== config2 default / multi-cu"; python3 scratch/qt.py 256 256 2 | tail -1; MEDGP_MULTI_CU=1 python3 scratch/qt.py 256 256 2 | tail -1
  echo "== config2 forced <8,4>"; MEDGP_CHOLINV_NW=84 python3 scratch/qt.py 256 256 2 | tail -1
  echo "== screening 1000xN512 nlml-only default"; python3 scratch/qt.py 1000 512 24 0 | tail -1
  echo "== 128 / 192 / 256 x N=512 default vs multi-cu vs not"; for P in 128 192 256; do python3 scratch/qt.py $P 512 24 | tail -1; MEDGP_MULTI_CU=1 python3 scratch/qt.py $P 512 24 | tail -1; MEDGP_MULTI_CU=-1 python3 scratch/qt.py $P 512 24 | tail -1; done
done > $O/routes.log 2>&1
for L in lib_lastamps.so lib_lastamps_noinv.so; do
  echo "=== $L N=2048"; LASTAMP_LIB=$L python3 scratch/la_stamps.py 2048 24
  echo "=== $L N=4096 D=64"; LASTAMP_LIB=$L python3 scratch/la_stamps.py 4096 64
done > $O/stamps.log 2>&1
tail -3 $O/pytest.log; tail -2 $O/bench.log | cut -c1-600; cat $O/routes.log | cut -c1-400; grep -E "^===|step  ?(1|4|16) " $O/stamps.log | cut -c1-300
