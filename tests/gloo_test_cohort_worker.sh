#!/bin/bash
# stand-in for medgp_test in the CPU test of medgp_amd.impute_cohort: writes the flag file of both passes per patient of the shard
# usage: ... --cfg CFG --pan-list FILE --fold K --kernclust-alg ALG --device D
while [ $# -gt 0 ]; do case "$1" in --cfg) CFG=$2; shift 2;; --pan-list) PL=$2; shift 2;; --fold) FOLD=$2; shift 2;; *) shift;; esac; done
python3 - "$CFG" "$PL" "$FOLD" <<'PY'
import json, sys, os
cfg = json.load(open(sys.argv[1]))
for pan in open(sys.argv[2]).read().split():
    for mode in ("mean_wo_update", "mean_w_update"):
        open(os.path.join(cfg["exp_test_dir"], f"test_{mode}_flag_{pan}.txt"), "w").write("1\n")
    if pan == "FAIL":
        sys.exit(3)
PY
