"""Solve-phase kernel timeline of the last step in a rocprofv3 kernel trace (see profiles/README.md)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_rhs_saddle" in r["Kernel_Name"] or "k_perm_in" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("hipfact::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = max(int(r["Workgroup_Size_X"]), 1)
    print(f"{n[:36]:36s} grid {int(r['Grid_Size_X']) // wg:5d} start {(s - t0) / 1000:8.1f} dur {(e - s) / 1000:7.2f}")
