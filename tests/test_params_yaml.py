"""The cvo_params yaml surface (read_CvoParams_yaml, CvoParams.hpp:193-303) incl. the quirks of the shipped files."""
import glob
import os
import warnings

import pytest
import yaml

import cases
from unified_cvo_amd import CvoParams, parse_cvo_yaml_text, read_cvo_params_yaml
from unified_cvo_amd.params import _YAML_KEYS

REF_PARAMS = "/root/reference/cvo_params"


def test_defaults_match_reference_constructor():
    p = CvoParams()
    assert (p.ell_init, p.ell_min, p.sigma, p.sp_thres, p.c, p.d) == (0.5, 0.05, 0.1, 0.0006, 7.0, 7.0)
    assert (p.MAX_ITER, p.nearest_neighbors_max, p.indicator_window_size) == (10000, 512, 15)
    assert (p.is_using_geometry, p.is_using_intensity, p.is_using_semantics, p.is_using_kdtree) == (1, 0, 0, 0)
    assert p.min_step == 2e-5 and p.eps_2 == 0.000012 and p.ell_decay_rate == 0.9 and p.ell_decay_start == 30


def test_config_files():
    p = cases.load_params("geometric_gpu")
    assert (p.ell_init, p.MAX_ITER, p.min_step, p.max_step, p.eps_2) == (0.3, 2000, 0.0001, 0.8, 0.000006)
    assert p.nearest_neighbors_max == 512 and p.is_using_intensity == 0  # absent keys keep defaults
    p = cases.load_params("outdoor")
    assert (p.nearest_neighbors_max, p.is_using_geometric_type, p.sp_thres, p.MAX_ITER) == (256, 1, 0.007, 100000)
    assert p.ell_decay_start_first_frame == 100 and p.ell_decay_rate_first_frame == 0.99
    p = cases.load_params("semantic_img_gpu0")
    assert (p.is_using_semantics, p.s_ell, p.indicator_window_size, p.max_step) == (1, 1.0, 30, 0.01)


def test_tolerant_reader_quirks():
    text = """%YAML:1.0
---
<<<<<<< HEAD
ell_init: 0.45
=======
ell_init: 0.2
>>>>>>> origin/range_ell
sigma: 0.1   # trailing comment
nearest_neighbors_max: 256
is_ell_adaptive: False
is_dense_kernel: 0
nearest_neighbors_max: 512
<<<<<<< HEAD
MAX_ITER: 5000
=======
MAX_ITER: 1500
indicator_window_size: 10
>>>>>>> origin/range_ell
"""
    p = parse_cvo_yaml_text(text)
    assert p.ell_init == 0.45 and p.MAX_ITER == 5000       # HEAD side of both conflict blocks
    assert p.indicator_window_size == 15                  # the 'theirs' side is dropped entirely
    assert p.nearest_neighbors_max == 256                 # first occurrence wins (yaml-cpp linear lookup)
    assert p.is_ell_adaptive == 0                         # never read by read_CvoParams_yaml
    assert any("conflict" in w for w in p.warnings) and any("duplicate" in w for w in p.warnings)


def test_bad_value_raises():
    with pytest.raises(ValueError):
        parse_cvo_yaml_text("MAX_ITER: lots\n")


@pytest.mark.skipif(not os.path.isdir(REF_PARAMS), reason="reference checkout not present")
def test_every_shipped_file_parses_like_pyyaml():
    """All 16 reference yaml files: our reader == PyYAML on the keys read_CvoParams_yaml asks for."""
    files = sorted(glob.glob(os.path.join(REF_PARAMS, "*.yaml")))
    assert len(files) == 16
    for path in files:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = read_cvo_params_yaml(path)
        lines, side = [], None
        for line in open(path).read().splitlines():
            if line.startswith("<<<<<<<"):
                side = "ours"
            elif line.startswith("=======") and side:
                side = "theirs"
            elif line.startswith(">>>>>>>") and side:
                side = None
            elif side != "theirs" and not line.startswith("%"):
                lines.append(line)
        seen = {}
        for line in lines:  # first occurrence wins
            body = line.split("#")[0]
            if ":" in body:
                k = body.split(":")[0].strip()
                seen.setdefault(k, yaml.safe_load(body.split(":", 1)[1]))
        for k, v in seen.items():
            if k in _YAML_KEYS:
                assert getattr(p, k) == pytest.approx(float(v)), (path, k)
        d = CvoParams()
        for k in _YAML_KEYS:
            if k not in seen:
                assert getattr(p, k) == getattr(d, k), (path, k)


def test_c_struct_cache_follows_the_members():
    """CvoParams.to_ctypes() hands out one cached cvo_params_t until a member changes (filling 52 fields per call was a
    sixth of a one-launch inner product); copies, deep copies and pickles do not share it."""
    import copy
    import pickle
    p = CvoParams()
    a = p.to_ctypes()
    assert a is p.to_ctypes()
    p.ell_init = 0.7
    b = p.to_ctypes()
    assert b is not a and abs(b.ell_init - 0.7) < 1e-6 and abs(a.ell_init - 0.5) < 1e-6
    p.warnings = ["x"]  # (not a member of the struct: the cache stays)
    assert p.to_ctypes() is b
    for q in (p.copy(), copy.deepcopy(p), pickle.loads(pickle.dumps(p))):
        q.nearest_neighbors_max = 7
        assert q.to_ctypes().nearest_neighbors_max == 7 and p.to_ctypes().nearest_neighbors_max == 512
        assert abs(q.to_ctypes().ell_init - 0.7) < 1e-6
