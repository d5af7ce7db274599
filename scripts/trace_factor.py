"""Per-level durations of the factorisation kernels from a rocprofv3 kernel trace (last complete factorisation)."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_mvals" in n]
start = idx[-1]
tot = {}
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("hipfact::", "")
    if n.startswith("k_rhs") or n.startswith("k_fwd"):
        break
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0) + d
    print("%-22s grid %6d  dur %8.2f us  lds %6s" % (n[:22], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), d, r["LDS_Block_Size"]))
print({k: round(v, 1) for k, v in tot.items()}, "total", round(sum(tot.values()), 1))
