"""medgp_amd -- MI355X-native hot path of bee-hive/MedGP (per-patient nlml + gradient).

The product is libmedgp_hip.so (hand-written HIP for gfx950 behind the C ABI in
include/medgp_hip.h).  This package is the thin Python host layer used by the tests and
bench.py: a ctypes binding (`capi`), the synthetic cohort generator (`synth`) and the
cohort sharding helper (`shard`).  There is no CPU fallback: importing works anywhere,
but every compute call needs the built library and a HIP device.
"""
from . import capi, synth, shard  # noqa: F401
from .capi import Context, MedgpError, lib_path, load  # noqa: F401

__all__ = ["capi", "synth", "shard", "Context", "MedgpError", "lib_path", "load"]
