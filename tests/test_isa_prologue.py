"""The per-iteration kernels are latency chains for a lone pair, and every dependent round of scalar loads in front of
a block's row loop is ~0.4 us of it.  Their prologues request everything they branch on in ONE burst of scalar loads
(cvo_kernels.h: the empty-asm pins in k_assoc / k_coeff).  The register allocator undid that twice in round 4 without
any source change near it (a dead register of a 16-byte load handed to another load of the burst -> a wait in the
middle: +0.6 us per iteration, found in the ISA).  This test compiles the device code (CPU only, hipcc cross-compiles)
and checks the shape of the prologues, so that the next such accident shows up here and not in a profile.
"""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("isa_loops", os.path.join(ROOT, "scripts", "isa_loops.py"))
isa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(isa)


@pytest.fixture(scope="module")
def device_code():
    lines = isa.device_asm([])
    return lines, isa.kernels(lines)


def _scalar_rounds_before_first_exit(lines, span):
    """Mnemonics from the first s_load that is not a kernel-argument load up to the first conditional branch after it:
    returns (number of scalar loads, number of lgkmcnt waits) in that stretch."""
    a, b = span
    ins = [(i, lines[i].split(";")[0].strip()) for i in range(a, b + 1)]
    ins = [(i, s) for i, s in ins if s and not s.endswith(":") and not s.startswith(".")]
    # kernel arguments: s_load from s[0:1] at the very top
    k = 0
    while k < len(ins) and not (ins[k][1].startswith("s_load") and "s[0:1]" not in ins[k][1]):
        k += 1
    loads = waits = 0
    for _, s in ins[k:]:
        if s.startswith("s_load"):
            loads += 1
        elif s.startswith("s_waitcnt") and "lgkmcnt" in s:
            waits += 1
        elif s.startswith("s_cbranch") and waits:
            break
    return loads, waits


@pytest.mark.parametrize("kernel,max_waits", [("k_assoc<unsigned short, 64, 0, false>", 1),   # FEAT_GEO
                                              ("k_assoc<unsigned short, 64, 1, false>", 1),   # FEAT_ALL
                                              ("k_assoc<unsigned short, 64, 2, false>", 1),   # FEAT_COL
                                              ("k_assoc<unsigned short, 64, 3, false>", 1),   # FEAT_HOT
                                              # k_coeff sits at the SGPR limit (the 42 floats of the twist matrices live in
                                              # scalar registers through its row loop): the compiler pulls parameter loads of
                                              # the update into the burst and spills them to a VGPR, a wait each time.  Measured
                                              # against the one-wait build (scripts/exp_time.py, interleaved): nothing - the row
                                              # heads' vector loads, requested before the burst, take longer than all of it.
                                              ("k_coeff<false>", 3)])
def test_prologue_is_one_burst_of_scalar_loads(device_code, kernel, max_waits):
    lines, ks = device_code
    assert kernel in ks, sorted(ks)
    loads, waits = _scalar_rounds_before_first_exit(lines, ks[kernel])
    assert loads >= 10        # the burst is there ...
    assert 1 <= waits <= max_waits, f"{kernel}: {waits} waits inside the first burst of {loads} scalar loads"


def test_no_scratch_in_the_per_iteration_kernels(device_code):
    lines, ks = device_code
    for kernel in ("k_assoc<unsigned short, 64, 0, false>", "k_assoc<unsigned short, 64, 2, false>",
                   "k_assoc<unsigned short, 64, 3, false>", "k_coeff<false>", "k_assoc_dense<0, 4, false>"):
        a, b = ks[kernel]
        assert not any(re.search(r"\bscratch_(load|store)", lines[i]) for i in range(a, b + 1)), kernel
