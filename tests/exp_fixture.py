"""The experiment-directory generator lives in the package (bench.py's host legs use it too); the tests keep this import path."""
from medgp_amd.synth_experiment import OPT, make_experiment  # noqa: F401
