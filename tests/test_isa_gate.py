"""The ISA gate (tools/isa_gate.py) on HEAD and on deliberately broken builds.

The 16-wide GRU kernels (ccsm_gru_f3s.hip, ccsm_gru_mx16.hip) issue their MFMAs through asm statements, so hipcc pads none of their hazards
and counts none of the waits they need; what keeps them correct is checked on the code object (gfx950 is cross-compiled here: no GPU needed).
The full gate (all twelve product instantiations, ~1 min) runs in __graft_entry__.build() whenever the library is rebuilt; here: the 96-row
forms on HEAD, and one small instantiation per rule with the protection removed - the gate must go red on each."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def test_gate_is_green_on_the_product_kernels_of_full_launches():
    import isa_gate
    bad, lines = isa_gate.run(only=["F3S12(3)", "MX16(false, 3)", "MX16(true, 3)"], quiet=True)
    assert bad == 0, "\n".join(lines)
    assert sum("32 counted waits checked (32 exact)" in ln for ln in lines) == 2, "\n".join(lines)
    assert all("scratch 0" in ln for ln in lines if "MFMAs" in ln), "\n".join(lines)


@pytest.mark.parametrize("define,only,needle", [
    ("-DCCSM_F3S_NO_FIRST_NOP", "F3S12(2)", "behind a VALU write"),                 # VALU write -> MFMA read without its two wait states
    ("-DCCSM_F3S_NO_DRAINED", "F3S12(1)", "behind the MFMA that wrote it"),         # XDL write -> VALU read inside the drain's wait states
    ("-DCCSM_MX16_BAD_WAIT", "MX16(false, 1)", "of the awaited transfer"),  # a counted wait that lets one load too many stay in flight
])
def test_gate_goes_red_on_a_broken_build(define, only, needle):
    import isa_gate
    bad, lines = isa_gate.run(defines=[define], only=[only], quiet=True)
    assert bad > 0, "\n".join(lines)
    assert any(needle in ln for ln in lines), "\n".join(lines)


def test_shipped_library_has_no_scratch_in_any_gru_kernel():
    """The BUILT libccsm.so (what travels to the GPU box), read from the code object inside it: every GRU instantiation - the compiler-scheduled
    32-wide ones included, which the gate's translation unit does not recompile - without scratch and without spills; the one kernel that has
    any (the fp8 pool: three registers saved once per launch, outside its loops) does not grow."""
    import isa_gate
    lib = os.path.join(ROOT, "ccsmeth_amd", "lib", "libccsm.so")
    if not os.path.exists(lib):
        pytest.skip("libccsm.so is not built")
    k = isa_gate.shipped_kernels(lib)
    gru = {n: v for n, v in k.items() if "gru_" in n}
    assert len(gru) >= 45, sorted(k)                                  # mx (32-wide) x forms x arithmetics, f3, f3s, mx16, v2
    for fam in ("gru_layer12_mx_kernel", "gru_layer0_mx_kernel", "gru_layer12_f3s_kernel", "gru_layer0_f3s_kernel", "gru_layer12_mx16_kernel"):
        assert any(fam in n for n in gru), fam
    bad = {n: v for n, v in gru.items() if v["private_segment_fixed_size"] or v["vgpr_spill_count"]}        # (SGPRs parked in VGPR lanes are not scratch)
    assert not bad, bad
    other = {n: v["private_segment_fixed_size"] for n, v in k.items() if v["private_segment_fixed_size"] and n not in gru}
    assert all("attn_fc_f8_kernel" in n and b <= 16 for n, b in other.items()), other
