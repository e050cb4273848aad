#!/bin/bash
# round 4, GPU call 2: full GPU test suite on the new host code, cohort imputation timing, route table, bench line with the config-4 cohort leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c2
O=gpurun_out/r4c2
(time python3 -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
python3 scratch/impute_cohort_time.py 64 200 16 > $O/impute.log 2>&1
python3 scratch/route_sweep.py > $O/routes_d24.log 2>&1
python3 scratch/route_sweep.py --d2 > $O/routes_d2.log 2>&1
python3 bench.py > $O/bench.log 2>&1
tail -4 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head; cat $O/impute.log | cut -c1-300; cat $O/routes_d24.log $O/routes_d2.log; tail -1 $O/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac']); print(json.dumps(d['other_configs'].get('config4_full_4096xN512_D24'))); print({k:(v.get('ms_per_call'),v.get('frac_fp64_peak')) for k,v in d['other_configs'].items() if isinstance(v,dict)})"
