"""The C-ABI library loads and exports every symbol include/cvo_hip.h (the drop-in boundary) and
include/cvo_hip_debug.h (test / profiling hooks) declare (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import cases
from unified_cvo_amd import _capi


def _declared(headers=("cvo_hip.h", "cvo_hip_debug.h")):
    out = set()
    for h in headers:
        text = open(os.path.join(cases.ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(cvo_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_debug_hooks_are_not_part_of_the_boundary_header():
    assert not [s for s in _declared(("cvo_hip.h",)) if s.startswith("cvo_debug_")]
    assert all(s.startswith("cvo_debug_") for s in _declared(("cvo_hip_debug.h",)))


def test_header_and_binding_lists_agree():
    assert _declared() == sorted(_capi.EXPORTED)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_capi.LIB_PATH), "build the HIP extension first (python -m unified_cvo_amd.build)"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH], text=True)
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    for sym in _declared():
        assert sym in exported, sym
    L = _capi.lib()
    assert b"gfx950" in L.cvo_version()


def test_params_struct_layout_matches_reference_order():
    names = [n for n, _ in _capi.cvo_params_t._fields_]
    assert names[:5] == ["ell_init_first_frame", "ell_init", "ell_min", "min_ell_iter_limit", "ell_max"]
    assert names[-1] == "multiframe_min_nonzeros" and len(names) == 52
    assert ctypes.sizeof(_capi.cvo_params_t) == 224  # 2 doubles force 8-byte alignment, as in the C++ struct
    p = _capi.cvo_params_t()
    _capi.lib().cvo_params_default(ctypes.byref(p))
    assert abs(p.ell_init - 0.5) < 1e-7 and p.nearest_neighbors_max == 512 and p.MAX_ITER == 10000
    assert abs(p.max_step - 0.8) < 1e-7


def test_code_object_targets_gfx950():
    out = subprocess.run(["strings", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(cases.ROOT, "unified_cvo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "pyoracle" not in text and "cvo_oracle" not in text and "libcvo_oracle" not in text, f
    ldd = subprocess.run(["ldd", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
