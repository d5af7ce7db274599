"""Average kernel durations of a rocprofv3 kernel trace directory (only kernels with >= 20 calls)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 20:
        print(r["Name"][:50].ljust(52), r["Calls"].rjust(6), "%9.1f us  min %7.1f  max %7.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
