"""CPU: static check of the inline-asm DPP instructions of the pipeline kernels' column role.

k_indirect_pipe8 / k_indirect_pipe48 (col_dpp_stage, pipe_common.hpp) issue their coefficient x column products as `v_fmac_f64_dpp ... row_newbcast:n` through inline asm
(the compiler has no pattern that folds a 64-bit DPP move into an FMA).  The compiler's hazard recognizer does not look
inside inline asm, so the two gfx9 DPP hazards are checked here on the generated assembly of every instantiation:
  * a VALU instruction that writes a VGPR read by a DPP instruction through its DPP operand (src0) needs 2 wait states
    in between -- by construction the DPP source registers are only ever written by LDS loads;
  * a VALU write of EXEC needs 5 wait states before a DPP instruction.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lowthrustopt_amd", "csrc")


def regs(tok):
    """'v[50:51]' / 'v7' / '-v[2:3]' -> set of VGPR numbers."""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_inline_asm_dpp_sources_are_never_written_by_a_valu_instruction_nearby(tmp_path):
    instrs = []          # (mnemonic, operand string) in stream order, one kernel after the other
    n_dpp = 0
    for src in ("kernels_indirect_pipe8.hip", "kernels_indirect_pipe48.hip"):
        out = str(tmp_path / (src + ".s"))
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-exceptions", "--cuda-device-only", "-S", "-I" + os.path.join(CSRC, "hooks"),
                               os.path.join(CSRC, src), "-o", out])
        for line in open(out):
            line = line.split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":") or line.startswith("//"):
                continue
            parts = line.split(None, 1)
            instrs.append((parts[0], parts[1] if len(parts) > 1 else ""))
    for i, (mn, ops) in enumerate(instrs):
        if mn != "v_fmac_f64_dpp":       # DPP instructions the compiler itself emits (wave reductions) are its business
            continue
        n_dpp += 1
        assert "row_newbcast" in ops, (mn, ops)
        src0 = regs(ops.split(",")[1])
        assert len(src0) == 2
        for back in (1, 2):
            pmn, pops = instrs[i - back]
            if pmn.startswith("v_") and not pmn.startswith("v_cmp"):
                dst = regs(pops.split(",")[0])
                assert not (dst & src0), "VALU write of the DPP source %d instruction(s) before %s %s: %s %s" % (back, mn, ops, pmn, pops)
        for back in range(1, 6):
            pmn, pops = instrs[i - back]
            if pmn.startswith("v_") and re.match(r"\s*exec", pops):
                raise AssertionError("VALU write of EXEC %d instruction(s) before a DPP instruction" % back)
    # per RK stage 44 / 45 DPP FMAs for ND = 14 (without / with lambda_m on the chain), 37 for ND = 12; 4 stages; the eight-wave
    # kernel carries the column step in four places (both steps of a phase, the alternating job's two waves), the 48-segment form in one
    assert n_dpp >= 4 * 4 * (44 + 37), n_dpp
