"""HBM traffic per launch of the three sparse products (J x, J^T y, symmetric H x) from the per-product PMC
passes of scripts/refresh_profiles.sh (`bench.py --spmv-only NAME` under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).
Counters in KiB; FETCH_SIZE doubled on gfx950 for wide coalesced reads (upper bound), as in pmc_summary.py."""
import csv, glob, hashlib, json, os, sys

out_dir, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]


def per_launch(d, counter):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    n, tot = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_spmv_csr" in r["Kernel_Name"]:
            n += 1
            tot += float(r["Counter_Value"])
    return n, (tot / n * 1024 if n else 0.0)


res = {}
print("%-10s %6s %16s %16s %16s" % ("product", "calls", "fetch B/launch", "x2 (gfx950)", "write B/launch"))
for op in ("J_x", "JT_y", "H_sym_x"):
    nf, f = per_launch(os.path.join(out_dir, f"pmc_spmv_fetch_{op}_{tag}"), "FETCH_SIZE")
    nw, w = per_launch(os.path.join(out_dir, f"pmc_spmv_write_{op}_{tag}"), "WRITE_SIZE")
    res[op] = {"calls": nf, "fetch_bytes_per_launch_raw": f, "fetch_bytes_per_launch_x2": 2 * f, "write_bytes_per_launch": w}
    print("%-10s %6d %16.0f %16.0f %16.0f" % (op, nf, f, 2 * f, w))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from sleqp_amd._lib import kernel_sources_sha16  # noqa: E402

res["_kernels_sha16"] = kernel_sources_sha16()
res["_workload"] = os.environ.get("HIPFACT_PROFILE_WORKLOAD", "banded_n1e5_m5e4")  # the passes ran bench.py's default workload
json.dump(res, open(dst, "w"), indent=1)
