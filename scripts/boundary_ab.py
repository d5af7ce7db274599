"""A/B of the host boundary's unit (set_matrix + solve + solution through the C shim) under environment variants, each in a
fresh process, alternating: python scripts/boundary_ab.py "" "HIPFACT_UPLOAD_PLAIN=1" """
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, json
sys.path.insert(0, %r)
import bench
J, N, cp, ri, vx, b = bench.make_problem("banded_n1e5_m5e4", 0)
r = bench.boundary_bench(J, N, cp, ri, vx, b, 30, 0)
print(json.dumps({k: r[k] for k in ("rate", "ms_per_unit", "set_matrix_ms", "solve_plus_solution_ms")}))
""" % ROOT
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for v in (sys.argv[1:] or [""]):
        env = dict(os.environ)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            env[k] = val
        out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"[{rnd}] {v or 'default':30s}", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
