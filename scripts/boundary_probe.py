import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from bench import make_problem, boundary_bench
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
for k in range(int(sys.argv[1]) if len(sys.argv)>1 else 2):
    o = boundary_bench(J, N, cp, ri, vx, b, 30, 0)
    print(os.environ.get("HIPFACT_HINT_PEEK","default"), round(o["rate"],1), round(o["ms_per_unit"],4), o["blocks_ms_per_unit"], round(o["set_matrix_ms"],4), round(o["solve_plus_solution_ms"],4), flush=True)
