import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f %s  ms/step %.3f  resid %.1e  factor_ms %.3f  solve %s" % (d["value"], d["unit"], d["ms_per_step"], d["scaled_residual"], d["factor_only_ms"], {k: round(v, 3) for k, v in d["solve_only"].items()}))
print("roofline", d["roofline"])
for k, v in d["kernels"].items():
    print("  %-8s %8.3f ms/step  %5.1f launches  %8.1f us avg" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_launch_us"]))
if "cpu_baseline" in d: print("cpu", d["cpu_baseline"])
