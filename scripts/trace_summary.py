"""Summarise a rocprofv3 kernel trace: per-kernel durations and inter-kernel gaps for one solve / one factorisation."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0]
    return n.replace("hipfact::", "")
# find the last complete step: take the last 200 dispatches
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -160:]
prev_end = None
tot_k = tot_gap = 0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-28s grid %6d  dur %8.2f us  gap %7.2f us  lds %6s vgpr %s" % (short(r["Kernel_Name"])[:28], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), (e - s) / 1e3, gap, r["LDS_Block_Size"], r["VGPR_Count"]))
    tot_k += (e - s) / 1e3; tot_gap += max(gap, 0)
    prev_end = e
print("total kernel %.1f us, total gaps %.1f us" % (tot_k, tot_gap))
