"""The vtable boundary (shim/fact_hipfact.c) under the variants of the host staging: round-3 path, kernel-read or
copy-engine upload of the right-hand side, copy-engine or kernel download of the solution.  One JSON line each."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import bench

    J, N, cp, ri, vx, b = bench.make_problem("banded_n1e5_m5e4", 0)
    out = bench.boundary_bench(J, N, cp, ri, vx, b, 60, 0)
    out.pop("note", None)
    print(json.dumps(out))
    sys.exit(0)
for env in ({"HIPFACT_BOUNDARY_FAST": "0"}, {"HIPFACT_BOUNDARY_H2D": "0", "HIPFACT_BOUNDARY_D2H": "0"},
            {"HIPFACT_BOUNDARY_H2D": "1", "HIPFACT_BOUNDARY_D2H": "0"}, {"HIPFACT_BOUNDARY_H2D": "0", "HIPFACT_BOUNDARY_D2H": "1"},
            {"HIPFACT_BOUNDARY_H2D": "1", "HIPFACT_BOUNDARY_D2H": "1"}):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, capture_output=True, text=True)
    print(json.dumps(env), (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)
