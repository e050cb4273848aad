"""bench.py's config3_patient_inside_a_lockstep_batch leg under different workgroup shapes of the bulk class (MEDGP_CHOLINV_NW=44|84):
can the look-ahead chain of the N = 2048 patient co-reside with the bulk's workgroups?  python scratch/config3_in_batch.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for nw in ("", "84", "44", "", "44"):
    if nw: os.environ["MEDGP_CHOLINV_NW"] = nw
    else: os.environ.pop("MEDGP_CHOLINV_NW", None)
    out = {}
    bench.config3_in_batch(out, 0, 2024, 5)
    r = out["config3_patient_inside_a_lockstep_batch"]
    print("NW", nw or "auto", {k: round(v["ms_per_call"], 4) for k, v in r.items() if isinstance(v, dict)}, "marginal", round(r["marginal_ms_of_the_N2048_patient"], 4), flush=True)
