"""ISA invariants the kernels rely on and the compiler does not promise: `marlin24_fused_w4_lean_kernel` issues its
scale / zero-point loads by inline asm and waits for them with a hand-written `s_waitcnt vmcnt(8)` — correct only if exactly the EIGHT
16-byte weight loads are issued between the two asm loads and the wait (vector-memory results return in order) and nothing spills in
between.  The GPU parity tests would catch a wrong result on today's toolchain; this test catches the cause at build time on any hipcc:
it compiles the source to gfx950 assembly (seconds, no GPU) and reads the instruction stream of every instantiation."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


@pytest.fixture(scope="module")
def marlin_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "ct_marlin24.s"
    src = os.path.join(ge.CSRC, "ct_marlin24.hip")
    r = subprocess.run([ge.HIPCC, *ge.HIP_FLAGS, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


def _functions(asm, needle):
    """{mangled name: [instruction lines]} of the functions whose name contains `needle`"""
    out, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1) if needle in m.group(1) else None
            if name:
                out[name] = []
            continue
        if name and line.startswith("\t") and not line.startswith("\t.") and not line.lstrip().startswith(";"):
            if line.strip().startswith("s_endpgm"):
                name = None
                continue
            out[name].append(line.strip())
    return out


def test_marlin_lean_kernel_hand_placed_wait_counts_exactly_the_weight_loads(marlin_asm):
    fns = _functions(marlin_asm, "marlin24_fused_w4_lean_kernel")
    assert len(fns) == 4, sorted(fns)  # <bf16|fp16 weights> x <bf16|fp16 scales>
    vmem = re.compile(r"^(global_|flat_|buffer_|scratch_)")
    for name, ins in fns.items():
        waits = [i for i, l in enumerate(ins) if l.startswith("s_waitcnt vmcnt(8)")]
        assert len(waits) == 1, (name, len(waits))
        w = waits[0]
        mem = [(i, l) for i, l in enumerate(ins[:w]) if vmem.match(l)]
        # the last ten vector-memory instructions before the wait: the two asm loads, then the eight weight loads — nothing else in between
        tail = [l.split()[0] for _, l in mem[-10:]]
        assert tail[:2] == ["global_load_ushort", "global_load_sbyte"], (name, tail)
        assert tail[2:] == ["global_load_dwordx4"] * 8, (name, tail)
        first = mem[-10][0]
        between = ins[first:w]
        assert not any(l.startswith(("scratch_", "buffer_store", "global_store", "flat_store")) for l in between), name
        # the registers the asm loads write are not touched (copied, spilled, overwritten) before the wait
        dst = [ins[mem[-10][0]].split()[1].rstrip(","), ins[mem[-9][0]].split()[1].rstrip(",")]
        for l in ins[mem[-9][0] + 1:w]:
            ops = re.findall(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", l)
            used = set()
            for a, lo, hi in ops:
                used |= {int(a)} if a else set(range(int(lo), int(hi) + 1))
            assert not ({int(d[1:]) for d in dst} & used), (name, l, dst)
