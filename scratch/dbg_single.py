import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd.synth_experiment import make_experiment
tmp = tempfile.mkdtemp()
ex = make_experiment(tmp, ["P001", "P002"], D=2, Q=3, R=2, N=[60, 75], prior_index=2)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host", "medgp_train")
for args in (["--pan", "P001"], ["--pan", "P001,P002"]):
    try:
        r = subprocess.run(["timeout", "-s", "QUIT", "20", exe, "--cfg", ex["cfg"], "--thread", "1"] + args, capture_output=True, text=True, timeout=40)
        print(args, "rc", r.returncode)
        print(r.stdout[-1500:])
        print(r.stderr[-500:])
    except subprocess.TimeoutExpired as e:
        print("TIMEOUT", (e.stdout or b"")[-1500:])
