"""Summarises a rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass
(counter_collection.csv): per kernel, fp64 MFMA flops per launch and the share of SIMD cycles the
matrix pipes were busy.  GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, the SQ counters over all
1024 SIMDs."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches, dur = collections.Counter(), collections.Counter()
per = collections.defaultdict(dict)  # (kernel, dispatch) -> counters of that launch
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("hipfact::", "").replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = per[(k, r.get("Dispatch_Id", ""))]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        launches[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("%-26s %8s %14s %12s %10s %12s" % ("kernel", "launches", "MFMA flop/call", "us/call", "TFLOP/s", "MFMA busy %"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0))[:8]:
    n = max(launches[k], 1)
    flops = v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512 / n
    us = dur[k] / n / 1e3
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    act = v.get("GRBM_GUI_ACTIVE", 0) / n / 8.0
    if flops == 0:
        continue
    print("%-26s %8d %14.3e %12.1f %10.2f %12.1f" % (k[:26], n, flops, us, flops / (us * 1e-6) / 1e12 if us else 0,
                                                   100.0 * busy / (act * 1024) if act else 0))

# the launch with the most matrix work of every kernel (the dense chain of config 3: the Schur update of its first front)
print()
print("%-26s %14s %12s %10s %12s   (largest launch)" % ("kernel", "MFMA flop", "us", "TFLOP/s", "MFMA busy %"))
best = {}
for (k, disp), d in per.items():
    fl = d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512
    if fl > best.get(k, (0, None))[0]:
        best[k] = (fl, d)
for k, (fl, d) in sorted(best.items(), key=lambda kv: -kv[1][0])[:4]:
    act = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-26s %14.3e %12.1f %10.2f %12.1f" % (k[:26], fl, d["us"], fl / (d["us"] * 1e-6) / 1e12 if d["us"] else 0,
                                               100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (act * 1024) if act else 0))
