"""CvoParams (include/UnifiedCvo/cvo/CvoParams.hpp) and its yaml surface, host side.

`read_cvo_params_yaml` mirrors read_CvoParams_yaml (CvoParams.hpp:193-303): every key is optional,
unknown keys are ignored, defaults come from CvoParams::CvoParams() (CvoParams.hpp:75-126).
The reader is deliberately tolerant of what the reference's shipped files contain
("config quirks"): the OpenCV-style `%YAML:1.0` directive, `#` comments,
duplicate keys (first occurrence wins, as yaml-cpp's linear lookup does) and unresolved git
conflict markers (the HEAD side is taken and a warning recorded).
"""
import ctypes as C
import warnings

from ._capi import cvo_params_t

# keys recognised by read_CvoParams_yaml, CvoParams.hpp:196-297
_YAML_KEYS = [
    "ell_init_first_frame", "ell_init", "ell_min", "min_ell_iter_limit", "ell_max", "dl", "dl_step", "sigma",
    "sp_thres", "c", "d", "c_ell", "c_sigma", "s_ell", "s_sigma", "MAX_ITER", "eps", "eps_2", "min_step",
    "max_step", "ell_decay_rate", "ell_decay_rate_first_frame", "ell_decay_start",
    "ell_decay_start_first_frame", "indicator_window_size", "indicator_stable_threshold",
    "is_pcl_visualization_on", "is_using_least_square", "is_full_ip_matrix", "is_using_geometry",
    "is_using_intensity", "is_using_semantics", "is_using_range_ell", "is_using_kdtree",
    "is_using_geometric_type", "is_exporting_association", "nearest_neighbors_max", "multiframe_using_cpu",
    "multiframe_ell_init", "multiframe_max_iters", "multiframe_ell_min", "multiframe_ell_decay_rate",
    "multiframe_iterations_per_ell", "multiframe_iterations_per_solve", "multiframe_downsample_voxel_size",
    "multiframe_expected_points", "multiframe_num_neighbors", "multiframe_min_nonzeros",
    "multiframe_least_squares_num_threads",
]

_FIELD_TYPES = {name: typ for name, typ in cvo_params_t._fields_}

# CvoParams::CvoParams(), CvoParams.hpp:75-126.  max_step and step are NOT initialised upstream;
# they get the documented stand-ins 0.8 / 0 (DESIGN.md "Config quirks").
_DEFAULTS = dict(
    ell_init_first_frame=0.5, ell_init=0.5, ell_min=0.05, min_ell_iter_limit=1, ell_max=1.2, dl=0.0,
    dl_step=0.3, sigma=0.1, sp_thres=0.0006, c=7.0, d=7.0, c_ell=0.15, c_sigma=0.6, s_ell=0.1, s_sigma=0.8,
    MAX_ITER=10000, min_step=2e-5, eps=0.00005, eps_2=0.000012, max_step=0.8, step=0.0, ell_decay_rate=0.9,
    ell_decay_rate_first_frame=0.99, ell_decay_start=30, ell_decay_start_first_frame=300,
    indicator_window_size=15, indicator_stable_threshold=0.2, is_pcl_visualization_on=0,
    is_using_least_square=0, is_ell_adaptive=0, is_full_ip_matrix=0, is_using_geometry=1,
    is_using_intensity=0, is_using_semantics=0, is_using_range_ell=0, is_using_kdtree=0,
    is_using_geometric_type=0, is_exporting_association=0, multiframe_using_cpu=1, multiframe_max_iters=200,
    nearest_neighbors_max=512, multiframe_ell_init=0.15, multiframe_ell_min=0.05, multiframe_iter_per_ell=10,
    multiframe_ell_decay_rate=0.7, multiframe_iterations_per_ell=50, multiframe_iterations_per_solve=8,
    multiframe_downsample_voxel_size=0.5, multiframe_expected_points=1000, multiframe_num_neighbors=128,
    multiframe_min_nonzeros=300, multiframe_least_squares_num_threads=24,
)


class CvoParams:
    """Plain attribute bag with the members of cvo::CvoParams; floats are stored at float32 precision."""

    def __init__(self, **overrides):
        for k, v in _DEFAULTS.items():
            setattr(self, k, v)
        self.warnings = []
        for k, v in overrides.items():
            if k not in _DEFAULTS:
                raise KeyError(k)
            setattr(self, k, v)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in _FIELD_TYPES:
            object.__setattr__(self, "_c", None)  # (the cached C struct is stale)

    def __getstate__(self):  # (the cache does not travel)
        d = dict(self.__dict__)
        d.pop("_c", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def to_ctypes(self):
        """The C struct the library takes (read-only for the caller: it is cached until a member changes - filling 52
        fields costs 7 us, a sixth of a one-launch inner product)."""
        p = self.__dict__.get("_c")
        if p is None:
            p = cvo_params_t()
            for name, _ in cvo_params_t._fields_:
                setattr(p, name, getattr(self, name))
            object.__setattr__(self, "_c", p)
        return p

    def as_dict(self):
        return {k: getattr(self, k) for k in _DEFAULTS}

    def copy(self):
        q = CvoParams(**self.as_dict())
        q.warnings = list(self.warnings)
        return q


def _parse_scalar(text, typ):
    text = text.strip()
    if len(text) >= 2 and text[0] == text[-1] and text[0] in "\"'":
        text = text[1:-1]
    if typ is C.c_int:
        return int(text)
    return float(text)


def parse_cvo_yaml_text(text, params=None):
    """Parses the `key: value` subset of YAML the reference's parameter files use."""
    p = params if params is not None else CvoParams()
    seen = set()
    side = None  # None = outside a conflict block, "ours", "theirs"
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = raw.rstrip()
        if line.startswith("<<<<<<<"):
            side = "ours"
            p.warnings.append(f"line {lineno}: unresolved git conflict marker; taking the HEAD side")
            continue
        if line.startswith("=======") and side is not None:
            side = "theirs"
            continue
        if line.startswith(">>>>>>>") and side is not None:
            side = None
            continue
        if side == "theirs":
            continue
        hash_pos = line.find("#")
        if hash_pos >= 0:
            line = line[:hash_pos]
        line = line.strip()
        if not line or line.startswith("%") or line == "---" or line == "...":
            continue
        if ":" not in line:
            p.warnings.append(f"line {lineno}: ignored ({raw.strip()!r})")
            continue
        key, _, value = line.partition(":")
        key = key.strip()
        if key not in _YAML_KEYS:
            continue  # unknown keys are ignored, as `if (fs["k"])` never asks for them
        if key in seen:
            p.warnings.append(f"line {lineno}: duplicate key {key!r}; keeping the first value")
            continue
        seen.add(key)
        try:
            setattr(p, key, _parse_scalar(value, _FIELD_TYPES[key]))
        except ValueError as e:
            raise ValueError(f"line {lineno}: cannot parse {key}: {value.strip()!r}") from e
    p.yaml_keys = seen
    return p


def read_cvo_params_yaml(path, params=None):
    with open(path, "r") as f:
        p = parse_cvo_yaml_text(f.read(), params)
    for w in p.warnings:
        warnings.warn(f"{path}: {w}")
    return p
