"""N>1 path on the CPU: two gloo processes, each with its own problem (seed = rank),
no data-path collective, max-over-ranks timing, whole-job aggregate."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np
    from sleqp_amd import _lib, synth
    from sleqp_amd.replicas import Replicas
    from plan_emul import Plan, EmulFactor
    rep = Replicas(backend="gloo")
    seed = rep.problem_seed()
    J = synth.banded_jacobian(400, 200, 8, 60, seed)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    lib = _lib.load()
    rep.barrier()
    t0 = time.perf_counter()
    P = Plan(lib, N, cp, ri, vx)                   # host part of the hot path (no GPU in this test)
    z = EmulFactor(P, vx).solve(np.ones(N))
    t_local = time.perf_counter() - t0 + 0.01 * rep.rank   # rank 1 is deliberately slower
    t_max = rep.max_over_ranks(t_local)
    out = {"rank": rep.rank, "world": rep.world, "seed": seed, "checksum": float(np.abs(vx).sum()),
           "resid": float(np.abs(synth.kkt_full_matrix(N, cp, ri, vx) @ z - 1.0).max()),
           "t_local": t_local, "t_max": t_max, "rate": rep.aggregate_rate(3, t_max)}
    print("RESULT " + json.dumps(out), flush=True)
    rep.close()
""")


def test_two_replicas_gloo(tmp_path, hipfact_lib):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, os.path.join(ROOT, "tests")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    import json

    res = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        res.append(json.loads(line[7:]))
    res.sort(key=lambda r: r["rank"])
    assert [r["world"] for r in res] == [2, 2] and [r["seed"] for r in res] == [0, 1]
    assert res[0]["checksum"] != res[1]["checksum"]  # independent problems
    assert all(r["resid"] < 1e-9 for r in res)
    t_max = max(r["t_local"] for r in res)
    assert all(abs(r["t_max"] - t_max) < 1e-12 for r in res)  # MAX over ranks
    assert all(abs(r["rate"] - 2 * 3 / t_max) < 1e-9 for r in res)  # whole-job aggregate
