"""Prints the kernel sequence of the last factor+solve step from a rocprofv3 --kernel-trace CSV
(start offset, duration, gap to the previous kernel, grid, name)."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_mvals_prod" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[max(i0 - 1, 0):i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("hipfact::", "").replace("void ", "")
    print("%9.1f us  dur %7.1f  gap %6.1f  grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
                                                          r.get("Grid_Size_X", r.get("Grid_Size", "?")), name[:60]))
    prev_end = e
