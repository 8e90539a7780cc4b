"""Shared builders of the BASELINE.json configurations for tests and bench.py."""
import os

import numpy as np

from unified_cvo_amd import CvoPointCloud, read_cvo_params_yaml, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_params(name):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return read_cvo_params_yaml(os.path.join(CONFIGS, name + ".yaml"))


def config2(n=5000, pair_id=0, m=None):
    """Synthetic xyz-only clouds, cvo_geometric_params_gpu.yaml, identity init."""
    p = load_params("geometric_gpu")
    src, tgt, _ = synth.geometric_pair(n, pair_id, m=m)
    return p, CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt), np.eye(4, dtype=np.float32)


def scene(n=10000, pair_id=0):
    """A clustered street-scene pair (synth.scene_pair: ground, facades, small dense objects; local density varies by
    more than 100x), cvo_geometric_params_gpu.yaml, identity init.  Not a BASELINE.json config: the evidence that the
    candidate-list machinery does not depend on the uniform density of the synthetic slab."""
    p = load_params("geometric_gpu")
    src, tgt, _ = synth.scene_pair(n, pair_id)
    return p, CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt), np.eye(4, dtype=np.float32)


def scene_colour(n=10000, pair_id=0):
    """The clustered street scene with colour features (synth.scene_colour_pair), cvo_intensity_params_gpu.yaml, identity
    init.  Not a BASELINE.json config: clustered density through the colour instantiations of the kernels."""
    p = load_params("intensity_gpu")
    src, fsrc, tgt, ftgt = synth.scene_colour_pair(n, pair_id)
    geo = np.tile(np.array([[0.0, 1.0]], np.float32), (n, 1))
    return (p, CvoPointCloud.from_arrays(src, fsrc, None, geo), CvoPointCloud.from_arrays(tgt, ftgt, None, geo),
            np.eye(4, dtype=np.float32))


def config3(n=10000, pair_id=0):
    """Colour clouds, cvo_intensity_params_gpu.yaml (HEAD side + documented overrides), identity init."""
    p = load_params("intensity_gpu")
    src, fsrc, tgt, ftgt, _, _ = synth.colour_pair(n, pair_id)
    geo = np.tile(np.array([[0.0, 1.0]], np.float32), (n, 1))
    return (p, CvoPointCloud.from_arrays(src, fsrc, None, geo), CvoPointCloud.from_arrays(tgt, ftgt, None, geo),
            np.eye(4, dtype=np.float32))


def config4(n=10000, pair_id=0):
    """Semantic clouds, cvo_semantic_params_img_gpu0.yaml, warm start T_gt o delta."""
    p = load_params("semantic_img_gpu0")
    src, fsrc, lsrc, tgt, ftgt, ltgt = synth.semantic_pair(n, pair_id)
    geo = np.tile(np.array([[0.0, 1.0]], np.float32), (n, 1))
    init = (synth.gt_motion() @ synth.warm_start_delta()).astype(np.float32)
    return (p, CvoPointCloud.from_arrays(src, fsrc, lsrc, geo), CvoPointCloud.from_arrays(tgt, ftgt, ltgt, geo),
            init)


def demo_clouds():
    """The two README demo clouds (parsed fixture, tests/golden/demo_clouds.npz)."""
    d = np.load(os.path.join(GOLDEN, "demo_clouds.npz"))
    return d["src_xyz"], d["src_rgb"], d["tgt_xyz"], d["tgt_rgb"]


def config1(geometric_only=True):
    """README demo: cvo_outdoor_params.yaml + the demo driver's preprocessing
    (main_cvo_gpu_align_two_color_pcd.cpp:27-81)."""
    p = load_params("outdoor")
    sx, sr, tx, tr = demo_clouds()
    src = CvoPointCloud.from_xyzrgb(sx, sr)
    tgt = CvoPointCloud.from_xyzrgb(tx, tr)

    def get_pc_mean(pc):  # float accumulation in index order, then / n (lines 27-33)
        m = np.zeros(3, np.float32)
        for q in pc.positions():
            m = (m + q).astype(np.float32)
        return (m / np.float32(pc.num_points())).astype(np.float32)

    d = get_pc_mean(src) - get_pc_mean(tgt)
    p.ell_init = float(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])))
    p.ell_decay_rate = p.ell_decay_rate_first_frame
    p.ell_decay_start = p.ell_decay_start_first_frame
    if geometric_only:
        p.is_using_intensity = 0
    return p, src, tgt, np.eye(4, dtype=np.float32)


def max_abs_diff(A, B):
    return float(np.max(np.abs(np.asarray(A, np.float64) - np.asarray(B, np.float64))))
