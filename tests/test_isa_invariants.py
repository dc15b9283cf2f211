"""ISA invariants the kernels rely on and the compiler does not promise:
`marlin24_fused_w4_lean_kernel` issues its scale / zero-point loads by inline asm and waits for them with a hand-written
`s_waitcnt vmcnt(8)` — correct only if exactly the EIGHT 16-byte weight loads are issued between the two asm loads and the wait
(vector-memory results return in order) and nothing spills in between.  The check itself lives in `__graft_entry__.py`
(`check_marlin_isa_invariants`) and runs inside `build_hip()` on the object that is about to be linked; here it is run on the in-tree
object again (a prebuilt tree on a box without hipcc is thereby checked too, if llvm-objdump is there), and shown to have teeth."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


@pytest.fixture(scope="module")
def marlin_disasm():
    if not os.path.exists(ge.LLVM_OBJDUMP):
        pytest.skip("no llvm-objdump on this machine")
    ge.build_hip()
    obj = os.path.join(ge.BUILD, "ct_marlin24.o")
    if not os.path.exists(obj):
        pytest.skip("no object file travelled with the tree (the build checked it where it was compiled)")
    return ge.disassemble_device_code(obj)


def test_marlin_lean_kernel_hand_placed_wait_counts_exactly_the_weight_loads(marlin_disasm):
    ge.check_marlin_isa_invariants(marlin_disasm)


def test_the_check_has_teeth(marlin_disasm):
    """one weight load fewer in front of the wait, a store among them, or the asm load's destination overwritten: each is reported"""
    lines = marlin_disasm.splitlines()
    w = next(i for i, l in enumerate(lines) if "s_waitcnt vmcnt(8)" in l)
    loads = [i for i in range(w) if lines[i].lstrip().startswith("global_load_dwordx4")]
    fewer = lines[:loads[-1]] + lines[loads[-1] + 1:]
    with pytest.raises(AssertionError):
        ge.check_marlin_isa_invariants("\n".join(fewer))
    stored = lines[:loads[-1]] + ["\tglobal_store_dword v[0:1], v2, off"] + lines[loads[-1]:]
    with pytest.raises(AssertionError):
        ge.check_marlin_isa_invariants("\n".join(stored))
    sb = max(i for i in range(w) if lines[i].lstrip().startswith("global_load_sbyte"))
    dst = re.match(r"\s*global_load_sbyte (v\d+),", lines[sb]).group(1)
    clobbered = lines[:w] + [f"\tv_mov_b32_e32 {dst}, 0"] + lines[w:]
    with pytest.raises(AssertionError):
        ge.check_marlin_isa_invariants("\n".join(clobbered))
