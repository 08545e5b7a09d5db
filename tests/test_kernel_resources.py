"""The occupancy the kernels were measured at, held without a GPU: registers, scratch and LDS of every gfx950 kernel in the built
library are read from the code object's metadata (tools/kernel_resources.py) and compared with the budgets DESIGN.md states
(wavefronts per SIMD by registers, workgroups per CU by LDS: 160 KB per CU, 512 registers per SIMD lane).  A change that makes the
compiler spill, or that takes a kernel over its register / LDS step, fails here before it costs a GPU run."""
import os

import pytest

from tools import kernel_resources as KR

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qoi_amd", "lib", "libqoi_mi355x.so")
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    ks = KR.kernels(LIB)
    assert len(ks) >= 36, sorted(ks)
    return ks


def _one(ks, needle):
    hits = [k for k in ks if needle in k]
    assert len(hits) == 1, (needle, hits)
    return ks[hits[0]]


def test_no_kernel_spills(kernels):
    import re
    for name, k in kernels.items():
        if re.search(r"enc_setsILi\dELi\dELi\dELb\dELb1EE", name):      # the PIPE experiment (QOIMI_ENC_PIPE=1): measured, not the default
            continue
        if re.search(r"enc_setsILi\dELi\dELi3ELb\dELb\dEE", name):       # the one-pass experiment (QOIMI_ENC_UNI=1): six spills outside the step loop
            continue
        assert k["scratch"] == 0 and k["vgpr_spills"] == 0, (name, k)


@pytest.mark.parametrize("ch", [3, 4])
@pytest.mark.parametrize("entry", [0, 1])
def test_encoder_hot_kernel_keeps_six_wavefronts_per_simd(kernels, ch, entry):
    k = _one(kernels, f"enc_setsILi{ch}ELi1ELi{entry}ELb0ELb0EE")              # exchange probe: the default
    if (ch, entry) == (3, 0):                                          # flat 3-channel images only: five wavefronts per SIMD
        assert k["vgpr"] <= 96 and k["agpr"] == 0, k
    else:
        assert k["vgpr"] <= 80 and k["agpr"] == 0, k                   # 512 / 6 = 85 -> 80 at the allocation granule
    assert 6 * k["lds"] <= LDS_PER_CU, k                               # six workgroups of four wavefronts per CU


def test_decoder_passes_keep_their_workgroups_per_cu(kernels):
    tr = _one(kernels, "dec_transcodeILi0ELb0ELb0ELb0EE")
    assert tr["vgpr"] <= 80 and 4 * tr["lds"] <= LDS_PER_CU, tr        # four workgroups of four wavefronts
    p3 = _one(kernels, "dec_summarize_recILb0E")
    assert 8 * p3["lds"] <= LDS_PER_CU and p3["vgpr"] <= 128, p3       # eight wavefronts (one per workgroup) per CU
    for och in (3, 4):
        p4 = _one(kernels, f"dec_segments_recILi{och}ELb0E")
        assert 6 * p4["lds"] <= LDS_PER_CU and p4["vgpr"] <= (128 if och == 4 else 168), p4   # six per CU by LDS: two per SIMD at most, 256 registers each would do
        p4f = _one(kernels, f"dec_segments_recILi{och}ELb1E")         # flat images (run descriptors): LDS-bound all the same
        assert 6 * p4f["lds"] <= LDS_PER_CU and p4f["vgpr"] <= 168, p4f


@pytest.mark.parametrize("ch", [3, 4])
def test_encoder_keeps_its_pixel_prefetch(ch):
    """enc_sets asks for a group's sixteen pixel pairs a group ahead and waits for them one by one as the steps reach them:
    `s_waitcnt vmcnt(N)` with N counting down through the group.  Whether the compiler keeps that shape has hung on unrelated
    code twice (round 3: a refill at the top of the loop; round 4: a branch with vector-memory operations in the group loop made
    every 3-channel load wait for itself, -31 % on 3-channel batches, invisible to every parity test).  The disassembly of the
    built library must show the staggered waits."""
    import re
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(LIB) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    dis = KR.disassembly(LIB, objdump)
    name = [k for k in dis if f"enc_setsILi{ch}ELi1ELi1ELb0ELb0EE" in k]
    assert len(name) == 1, [k for k in dis if "enc_sets" in k]
    waits = [int(m.group(1)) for l in dis[name[0]] for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
    deep = [w for w in waits if w >= 3]
    assert len(deep) >= 16, (len(waits), sorted(set(waits)))          # two groups of eight steps in the pipelined loop


def _written_vgprs(line: str):
    """VGPR numbers an instruction line writes (its first operand where the mnemonic has a destination) - good enough for the check below."""
    import re
    m = re.match(r"^(\S+)\s+(.*)$", line)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("s_", "ds_write", "ds_or_b", "buffer_store", "global_store", "flat_store", "scratch_store", "v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane")) \
            or op in ("ds_wrxchg_rtn_b32",):
        return set()
    first = rest.split(",")[0].strip()
    r = re.match(r"^v(\d+)$", first)
    if r:
        return {int(r.group(1))}
    r = re.match(r"^v\[(\d+):(\d+)\]$", first)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    return set()


def test_exchange_results_are_not_overwritten_in_flight():
    """`ds_wrxchg_rtn_b32` sits in asm blocks: the compiler does not know that the instruction writes its result LATER, when the LDS
    answers, and is free to hand the result register to something else behind the block if the C++ side drops the value (round 5: the
    state look-back's first build lost half of an address that way - a memory fault on the GPU, 152 parity tests green before it).
    Held statically: in every kernel of the built library, between an exchange and the next `s_waitcnt` that waits for LDS results
    no instruction may write the exchange's result register (another exchange into the same register excepted: LDS results return in order)."""
    import re
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(LIB) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    dis = KR.disassembly(LIB, objdump)
    seen = 0
    for name, lines in dis.items():
        for i, l in enumerate(lines):
            m = re.match(r"^ds_wrxchg_rtn_b32\s+v(\d+),", l)
            if not m:
                continue
            seen += 1
            dest = int(m.group(1))
            for l2 in lines[i + 1:i + 400]:
                if re.match(r"^s_waitcnt\b.*lgkmcnt\(0\)", l2) or l2.startswith(("s_endpgm", "s_branch", "s_cbranch", "s_setpc")):
                    break                                     # waited for (or control leaves the straight line: not followed)
                assert dest not in _written_vgprs(l2), (name, l, l2)
    assert seen >= 20, seen                                   # the encoder's probes and replays are there at all
