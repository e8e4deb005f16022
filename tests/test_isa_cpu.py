"""CPU-only: checks on the gfx950 ISA the build leaves in multimodal-idbn_amd/build/ (hipcc -save-temps).

The streaming kernels wait for their LDS-DMA with HAND-COUNTED ``s_waitcnt vmcnt(N)`` (the compiler does not count asm
loads).  In the weight-update kernel K3 the count is "the register loads of the weight prefetch may stay in flight":
N must equal the number of ``global_load_dwordx4`` the compiler REALLY emitted between the last LDS-DMA and the wait --
in round 1 the middle pass of a multi-chunk batch waited vmcnt(32) with 16 loads emitted (its W loads were dead code),
i.e. it did not wait for its LDS-DMA at all.  This test would have caught it, for every instantiation.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "multimodal-idbn_amd", "build", "engine-hip-amdgcn-amd-amdhsa-gfx950.s")


@pytest.fixture(scope="module")
def functions():
    import __graft_entry__ as ge
    if not (os.path.exists(ISA) and ge._fresh(ISA)):
        ge.build()                       # compiles (about 3 minutes) only when the ISA is missing or older than the sources
    if not os.path.exists(ISA):
        pytest.skip("no ISA listing (library built by an older build())")
    fn, out = None, {}
    for line in open(ISA):
        m = re.match(r"^(_ZN5imdbn\w+):\s", line)
        if m:
            fn = m.group(1)
            out[fn] = []
        elif fn is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                fn = None
            else:
                out[fn].append(line.strip())
    return out


def _hand_waits(body):
    """(index, N) of every s_waitcnt vmcnt(N) inside an inline-asm block."""
    res, in_asm = [], False
    for i, l in enumerate(body):
        if l.startswith(";;#ASMSTART"):
            in_asm = True
        elif l.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l)
            if m:
                res.append((i, int(m.group(1))))
    return res


def test_k3_counted_waits_match_the_emitted_prefetch_loads(functions):
    k3 = {k: v for k, v in functions.items() if "assoc_update_planesILi" in k}
    assert len(k3) >= 8, sorted(functions)[:5]
    checked = 0
    for name, body in k3.items():
        for idx, n in _hand_waits(body):
            if n == 0:
                continue
            # walk back to the most recent LDS-DMA; count the register loads issued after it
            loads, j = 0, idx - 1
            while j >= 0 and "global_load_lds_dwordx4" not in body[j]:
                if re.match(r"global_load_dwordx4\b", body[j]):
                    loads += 1
                j -= 1
            assert j >= 0, f"{name}: counted wait vmcnt({n}) without a preceding LDS-DMA"
            assert loads == n, f"{name}: s_waitcnt vmcnt({n}) but {loads} global_load_dwordx4 between the last LDS-DMA and the wait"
            checked += 1
    updating = [k for k in k3 if "assoc_update_planesILi0E" in k]      # MODE 0: the kernels that stream W / W_m
    assert len(updating) >= 4 and checked >= 2 * len(updating)          # prologue + tile loop of each


def _check_term_loop_counts(name, body, na):
    """k1_stream's loop over bf16 terms (kernels_stream.hpp): per K16 step 2 LDS-DMA of weights and 2 NA LDS-DMA of A fragments, issued
    after the step's MFMAs as W(i+2) A(i+2); its counted wait leaves the younger W(i+1) A(i+1) in flight = 2 NA + 2 operations.  Checked in the emitted code: between two consecutive counted waits of the steady state lie exactly 2 NA + 2
    LDS-DMA (one step's worth), and no register-destination load at all (the compiler would wait vmcnt(0) for it)."""
    steady = 2 * na + 2
    in_asm, seen, dma, plain, counting = False, 0, 0, 0, False
    for l in body:
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if "global_load_lds_dwordx4" in l:
            dma += 1
        elif re.match(r"(global|buffer|flat)_load_", l):
            plain += 1
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", l) if in_asm else None
        if m:
            # the step that started with the steady-state wait: its refills up to the next hand-written vmcnt wait (the prologue's
            # wait for the bit plane can have the same count: such segments are not steps and are skipped)
            if counting and dma == 2 * na + 2 and plain == 0:
                seen += 1
            counting = int(m.group(1)) == steady
            dma = plain = 0
    assert seen >= 1, f"{name}: no steady-state step (vmcnt({steady})) found"


def test_streaming_kernels_have_no_scratch_and_only_the_planned_counts(functions):
    ks = {k: v for k, v in functions.items() if "k1_streamILi" in k}
    assert len(ks) >= 5
    for name, body in ks.items():
        na = int(re.search(r"k1_streamILi\dELi(\d)E", name).group(1))      # bf16 terms the kernel can read per element (0: bit planes only)
        counts = {n for _, n in _hand_waits(body)}
        planned = {0, 4, 8}                                   # 0, 4 (D - 1), 4 D  (kernels_stream.hpp K1S_D = 2): the bit-plane loop
        if na:
            planned |= {2 + 2 * na}                                       # loop over bf16 terms: W(i+1) A(i+1) may stay in flight
            planned |= {4 + 2, 8 + 2}                                     # bit-plane loop with the 2 exactness-map loads issued behind the first ring slots
        assert counts <= planned, (name, counts)
        assert not any(l.startswith("scratch_") for l in body), f"{name} spills to scratch"
        if na:
            _check_term_loop_counts(name, body, na)
        # everything these kernels keep in flight is LDS-DMA: a register-destination load written in inline asm is regarded as ready at
        # once by the compiler, which then copies its destination around before the wait that makes it valid (seen twice in round 3)
        in_asm = False
        for l in body:
            in_asm = l.startswith(";;#ASMSTART") or (in_asm and not l.startswith(";;#ASMEND"))
            assert not (in_asm and re.match(r"(global|buffer|flat)_load_(dword|ushort|ubyte)", l)), f"{name}: register load in inline asm: {l}"
    for name, body in functions.items():
        if "k2_streamILi" in name:
            assert not any(l.startswith("scratch_") for l in body), f"{name} spills to scratch"
