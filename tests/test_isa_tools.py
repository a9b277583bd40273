"""The CPU-side analysis tools of round 6 on a small instantiation (gfx950 is cross-compiled here): tools/isa_step_mix.py counts what one step asks
of the matrix pipe and of the vector-memory path on the code object, tools/cu_issue_sim.py replays the step loop against the model's rules."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def test_step_mix_and_issue_model_on_the_32_row_split3_kernel():
    import isa_gate
    import isa_step_mix as mixm
    import cu_issue_sim as sim
    s = isa_gate.compile_asm([], only=["F3S12(1)"])
    bodies = [(n, b) for n, b in mixm.kernel_bodies(s) if "gru_layer12_f3s_kernel" in n]
    assert len(bodies) == 1
    ins = mixm.step_loop(bodies[0][1])
    c, mfma_cyc, req_bytes, lds_cyc = mixm.mix(ins)
    # 32 rows: 3 passes x (512 + 256) / 32 k-blocks x 3 gates x 4 unit tiles of 16 = 864 MFMAs of 4 passes per wave and step
    assert c["v_mfma_f32_16x16x32_f16"] == 864 and mfma_cyc == 864 * 16
    assert 250 * 1024 < req_bytes < 400 * 1024 and lds_cyc > 0            # the weight stream (288 KiB per wave and step) + its part of the transfers
    prog = sim.program(bodies[0][1])
    assert sum(1 for i in prog if i.kind == "mfma") == 864 and sum(1 for i in prog if i.kind == "barrier") == 32
    per_step, stat = sim.simulate(prog, 3, sim.Params())
    assert len(per_step) == 2 and per_step[0] == pytest.approx(per_step[1], rel=0.02)
    # never faster than the busier of the two units it asks for, never slower than everything serialised
    assert max(2 * mfma_cyc, 8 * req_bytes / 64) <= per_step[1] <= 2 * mfma_cyc + 8 * req_bytes / 64 + 8 * 4000 * 4
