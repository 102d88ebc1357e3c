"""Why is the README recipe slower as a CHILD of bench.py than alone?  Runs the CLI three times as a child of a parent that holds
(a) nothing, (b) a HIP context + 30 GB of device memory, (c) + 6 GB of page-locked host memory, (d) + a one-rank RCCL group, and
prints the child's wall time and its feed line (what the feeder thread waited for).   python tools/cli_child_probe.py DIR REF.npz"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d, ref = sys.argv[1], sys.argv[2]
cmd = [sys.executable, "-m", "tise_toolbox_amd.fid_score", "--batch-size", "50", "--path1", ref, "--path2", d, "--synthetic-weights"]
env = dict(os.environ, TISE_TIMING="1", PYTHONPATH=ROOT)


def child(tag, n=3):
    for i in range(n):
        t0 = time.perf_counter()
        extra_argv = env.pop("__argv", "").split()
        r = subprocess.run(cmd + extra_argv, env=env, cwd=ROOT, capture_output=True, text=True)
        if extra_argv:
            env["__argv"] = " ".join(extra_argv)
        dt = time.perf_counter() - t0
        feed = [ln for ln in r.stderr.splitlines() if ln.startswith("[tise] png feed")]
        line = feed[-1] if feed else r.stderr[-300:]
        tail = line[line.find("feeder waited"):] if "feeder waited" in line else line[-200:]
        ph = " ".join(ln.split(": ")[-1].split(" s after")[0] for ln in r.stderr.splitlines() if "[tise timing]" in ln)
        print(f"{tag} run {i}: {dt:.2f} s | {line[17:60]} | phases {ph}", flush=True)


variants = [("fast exit (default)", {}), ("TISE_FAST_EXIT=0", {"TISE_FAST_EXIT": "0"})]
base = dict(env)
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 5):
    for name, extra in variants:
        env.clear(); env.update(base); env.update(extra)
        child(f"[{name}] round {rnd}", 1)
sys.exit(0)
