#!/bin/bash
# stand-in for medgp_train in the CPU tests of medgp_amd.train_cohort: writes a train_hyp file per patient of the shard.
# MEDGP_FAKE_UNIT=<seconds>: sleeps that long per patient, ten times as long for the patients listed in MEDGP_FAKE_HEAVY
# (space separated) -- evaluation budgets the scheduler's cost model cannot see (early stops, ref: c_optimizer_varEM.cpp:89-95).
# usage: ... --cfg CFG --pan-list FILE --device D --max-batch B
while [ $# -gt 0 ]; do case "$1" in --cfg) CFG=$2; shift 2;; --pan-list) PL=$2; shift 2;; *) shift;; esac; done
python3 - "$CFG" "$PL" <<'PY'
import json, sys, numpy as np, os, zlib, time
cfg = json.load(open(sys.argv[1]))
Q, D, R = int(cfg["Q"]), int(cfg["D"]), int(cfg["R"])
H = D + Q * (D * R + 2 + D)
unit = float(os.environ.get("MEDGP_FAKE_UNIT", "0"))
heavy = set(os.environ.get("MEDGP_FAKE_HEAVY", "").split())
for pan in open(sys.argv[2]).read().split():
    if unit > 0:
        time.sleep(unit * (10 if pan in heavy else 1))
    rng = np.random.default_rng(zlib.crc32(pan.encode()))
    rng.normal(size=H).tofile(os.path.join(cfg["exp_train_dir"], f"train_hyp_{pan}.bin"))
PY
