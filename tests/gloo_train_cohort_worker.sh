#!/bin/bash
# stand-in for medgp_train in the CPU tests of medgp_amd.train_cohort: writes a train_hyp file per patient it is given.
# Speaks the trainer's protocol: list lines are "PAN [n]", walked longest first (stable); with --queue FILE the patients are
# PULLED one by one from the shared flock-protected counter (medgp_train.cpp Tickets), otherwise the whole list is this process's.
# MEDGP_FAKE_UNIT=<seconds>: sleeps that long per patient, ten times as long for the patients listed in MEDGP_FAKE_HEAVY
# (space separated) -- evaluation budgets the scheduler's cost model cannot see (early stops, ref: c_optimizer_varEM.cpp:89-95).
# MEDGP_FAKE_KILL=<PAN>: the process kills itself (SIGKILL) when it reaches that patient -- a HIP fault / OOM kill stand-in.
# usage: ... --cfg CFG --pan-list FILE --device D --max-batch B [--resident R] [--queue FILE]
Q=""
while [ $# -gt 0 ]; do case "$1" in --cfg) CFG=$2; shift 2;; --pan-list) PL=$2; shift 2;; --queue) Q=$2; shift 2;; *) shift;; esac; done
exec python3 - "$CFG" "$PL" "$Q" <<'PY'
import fcntl, json, sys, numpy as np, os, signal, zlib, time
cfg = json.load(open(sys.argv[1]))
Q, D, R = int(cfg["Q"]), int(cfg["D"]), int(cfg["R"])
H = D + Q * (D * R + 2 + D)
unit = float(os.environ.get("MEDGP_FAKE_UNIT", "0"))
heavy = set(os.environ.get("MEDGP_FAKE_HEAVY", "").split())
kill = os.environ.get("MEDGP_FAKE_KILL", "")
rows = [ln.split() for ln in open(sys.argv[2]) if ln.strip()]
pans = [r[0] for r in rows]
ns = [int(r[1]) if len(r) > 1 else 0 for r in rows]
order = sorted(range(len(pans)), key=lambda i: -ns[i])       # stable: ties keep the list order


def take(path):
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        raw = os.read(fd, 32)
        k = int(raw) if raw.strip() else 0
        os.lseek(fd, 0, os.SEEK_SET)
        os.ftruncate(fd, 0)
        os.write(fd, str(k + 1).encode())
        return k
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


k = 0
while True:
    if sys.argv[3]:
        k = take(sys.argv[3])
    if k >= len(order):
        break
    pan = pans[order[k]]
    k += 1
    if pan == kill:
        os.kill(os.getpid(), signal.SIGKILL)
    if unit > 0:
        time.sleep(unit * (10 if pan in heavy else 1))
    rng = np.random.default_rng(zlib.crc32(pan.encode()))
    rng.normal(size=H).tofile(os.path.join(cfg["exp_train_dir"], f"train_hyp_{pan}.bin"))
    print(f"finish individual id: {pan} w/ {ns[order[k - 1]]} samples; flag = 1; final loss = 0", flush=True)
PY
