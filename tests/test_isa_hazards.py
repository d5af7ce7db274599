"""Static check of the hand-written DPP instructions in the diagonal-block kernel.

The hardware needs two wait states between a vector-ALU write of a register and its use as the
DPP-shuffled source of a following instruction, and the compiler's hazard recogniser does not look
into inline assembly (sleqp_amd/csrc/kernels_front_pivot.inc: rowb_f64 / fmac_rowb_f64 rely on the source
order instead).  This test compiles the device code to assembly and checks every `*_dpp`
instruction against the two instructions in front of it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sleqp_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)\b")


def _regs(tok):
    m = REG.fullmatch(tok.strip().lstrip("-|").rstrip("|"))
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def check_stream(lines):
    """Returns the number of DPP instructions checked; raises AssertionError on a hazard."""
    insts, checked = [], 0
    for line in lines:
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            if line.endswith(":") and not line.startswith(".LBB"):
                insts = []  # new function
            continue
        parts = line.split(None, 1)
        mnem, ops = parts[0], (parts[1] if len(parts) > 1 else "")
        ops = [o.strip() for o in re.split(r",\s*", ops.split(" row_")[0].split(" quad_perm")[0])]
        if mnem.endswith("_dpp"):
            src0 = _regs(ops[1])
            assert src0, line
            # walk back over two wait states: every instruction is one, s_nop N is N + 1
            states, j = 0, len(insts) - 1
            while states < 2 and j >= 0:
                pm, pops = insts[j]
                if pm == "s_nop":
                    states += int(pops[0], 0) + 1
                else:
                    if pm.startswith("v_") and not pm.startswith("v_cmp") and pops:
                        assert not (_regs(pops[0]) & src0), f"DPP hazard: `{pm} {', '.join(pops)}` then `{line}`"
                    states += 1
                j -= 1
            checked += 1
        insts.append((mnem, ops))
    return checked


def test_hazard_checker_itself():
    ok = ["f:", "v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]", "s_nop 1",
          "v_mov_b64_dpp v[8:9], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert check_stream(ok) == 1
    far = ["f:", "v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]", "v_add_f64 v[10:11], v[4:5], v[6:7]",
           "v_mul_f64 v[12:13], v[4:5], v[6:7]", "v_fmac_f64_dpp v[8:9], v[2:3], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf"]
    assert check_stream(far) == 1  # src1 may be fresh, only the shuffled source counts
    for bad in (["f:", "v_fmac_f64_e32 v[2:3], v[4:5], v[6:7]",
                 "v_mov_b64_dpp v[8:9], v[2:3] row_newbcast:3 row_mask:0xf bank_mask:0xf"],
                ["f:", "v_mov_b32_e32 v3, v9", "s_nop 0",
                 "v_fmac_f64_dpp v[8:9], v[2:3], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf"]):
        with pytest.raises(AssertionError):
            check_stream(bad)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_dpp_sources_are_two_wait_states_away_from_their_producers(tmp_path):
    out = tmp_path / "hipfact.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                           "--cuda-device-only", "-S", os.path.join(CSRC, "kernels_factor.hip"), "-o", str(out)],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    checked = check_stream(open(out))
    assert checked > 300  # the diagonal block alone has ~170 of them per instantiation
    shutil.rmtree(tmp_path, ignore_errors=True)
