"""A/B on one box: bench.py's unit (refactor_device + solve_device) and per-kernel event times under environment
variants given as NAME=VALUE,NAME=VALUE;... on the command line (each variant runs in a fresh child process so that the
knobs read at hipfact_create apply).  Example: python scripts/ab_bench.py "HIPFACT_PANEL64_MIN=0" "HIPFACT_PANEL64_MIN=192" """
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sys.argv[1:] or [""]
rounds = int(os.environ.get("AB_ROUNDS", "2"))
for rnd in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            env[k] = val
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-ceilings", "--no-extras",
                              "--steps", "200", "--warmup", "20"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = {a: round(b["avg_launch_us"], 1) for a, b in d["kernels"].items()}
            print(f"[{rnd}] {v or 'default':40s} value {d['value']:.1f} ms {d['ms_per_step']:.4f} factor {d['factor_only_ms']:.4f} solve {d['solve_only']['ms_per_solve']:.4f} {k}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(v, "failed", e, out.stderr[-400:])
