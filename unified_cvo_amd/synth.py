"""Deterministic synthetic frame pairs of SURVEY.md section 8(d) (numpy PCG64, fixed seeds).

Conventions: target y = R_gt x + t_gt (+ noise), so the expected *returned* transform of align()
is T_gt^-1 and the expected internal state / `init` argument is T_gt.
"""
import numpy as np

FEATURE_DIMENSIONS = 5
NUM_CLASSES = 19


def rot_axis_angle(axis, deg):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    th = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def gt_motion():
    """rot(axis=(0.2,1,0.1), 1.5 deg) + trans (0.05, 0.02, 0.40) m as a 4x4 float64."""
    T = np.eye(4)
    T[:3, :3] = rot_axis_angle((0.2, 1.0, 0.1), 1.5)
    T[:3, 3] = (0.05, 0.02, 0.40)
    return T


def warm_start_delta():
    """Config 4's perturbation: rot(x, 0.2 deg), trans (0.01, 0.005, 0.03) m."""
    T = np.eye(4)
    T[:3, :3] = rot_axis_angle((1.0, 0.0, 0.0), 0.2)
    T[:3, 3] = (0.01, 0.005, 0.03)
    return T


def _box(n):
    # 10k points live in x[-10,10] y[-2,2] z[2,30]; other sizes scale x and z by the cube root of
    # n/10k (5k points -> x[-7.9,7.9], z[2,24.2])
    s = (n / 10000.0) ** (1.0 / 3.0)
    return (-10.0 * s, 10.0 * s), (-2.0, 2.0), (2.0, 2.0 + 28.0 * s)


def geometric_pair(n, pair_id=0, m=None, noise=0.01):
    """Source/target xyz (float32) of one config-2/5 style pair."""
    m = n if m is None else m
    rs = np.random.default_rng(1000 + pair_id)
    (x0, x1), (y0, y1), (z0, z1) = _box(max(n, m))
    nn = max(n, m)
    pts = np.stack([rs.uniform(x0, x1, nn), rs.uniform(y0, y1, nn), rs.uniform(z0, z1, nn)], axis=1)
    src = pts[:n]
    rt = np.random.default_rng(2000 + pair_id)
    T = gt_motion()
    tgt = pts[:m] @ T[:3, :3].T + T[:3, 3] + rt.normal(0.0, noise, (m, 3))
    perm = rt.permutation(m)
    return src.astype(np.float32), tgt[perm].astype(np.float32), perm


def scene_pair(n, pair_id=0, m=None, noise=0.01):
    """A clustered frame pair (not part of BASELINE.json: the uniform slab of geometric_pair has 2-3 neighbours per point
    everywhere).  A street scene as a forward-looking depth sensor samples it: a ground plane (55 % of the points, image
    sampling: density falls with the square of the depth), two facades (25 %), a dozen small dense objects - poles,
    boxes (20 %); local density varies by more than 100x between a nearby object and the far ground.  Same motion,
    noise and target permutation as geometric_pair."""
    m = n if m is None else m
    nn = max(n, m)
    rs = np.random.default_rng(6000 + pair_id)
    n_ground, n_wall = int(0.55 * nn), int(0.25 * nn)
    n_obj = nn - n_ground - n_wall
    # ground y = -1.6: uniform in the IMAGE (u, v) -> depth z = h / v, lateral x = u z
    v = rs.uniform(1.6 / 40.0, 1.6 / 2.5, n_ground)
    z = 1.6 / v
    u = rs.uniform(-0.6, 0.6, n_ground)
    ground = np.stack([u * z, np.full(n_ground, -1.6) + rs.normal(0, 0.02, n_ground), z], axis=1)
    # facades x = +-7 (depth again image-sampled), 0 .. 6 m high
    zw = 1.0 / rs.uniform(1.0 / 40.0, 1.0 / 4.0, n_wall)
    side = np.where(rs.random(n_wall) < 0.5, -7.0, 7.0)
    wall = np.stack([side + rs.normal(0, 0.03, n_wall), rs.uniform(-1.6, 4.4, n_wall), zw], axis=1)
    # objects: a dozen boxes / poles of 0.3 .. 1.5 m, points on them ~ uniformly
    k = 12
    centres = np.stack([rs.uniform(-5.5, 5.5, k), np.full(k, -1.6), rs.uniform(4.0, 25.0, k)], axis=1)
    sizes = np.stack([rs.uniform(0.15, 0.8, k), rs.uniform(0.8, 2.5, k), rs.uniform(0.15, 0.8, k)], axis=1)
    which = rs.integers(0, k, n_obj)
    obj = centres[which] + np.stack([rs.uniform(-1, 1, n_obj), rs.uniform(0, 1, n_obj), rs.uniform(-1, 1, n_obj)], axis=1) * sizes[which]
    pts = np.concatenate([ground, wall, obj], axis=0)[rs.permutation(nn)]
    src = pts[:n]
    rt = np.random.default_rng(7000 + pair_id)
    T = gt_motion()
    tgt = pts[:m] @ T[:3, :3].T + T[:3, 3] + rt.normal(0.0, noise, (m, 3))
    perm = rt.permutation(m)
    return src.astype(np.float32), tgt[perm].astype(np.float32), perm


def colour_features(xyz, rng, noise=0.0):
    """5 channels: rgb = 0.5 + 0.5 sin(w_c . xyz + phi_c), 2 gradient channels ~ N(0.5, 0.05)."""
    w = np.array([[0.9, 0.3, 0.2], [0.2, 1.1, 0.4], [0.5, 0.6, 0.8]])
    phi = np.array([0.3, 1.1, 2.0])
    rgb = 0.5 + 0.5 * np.sin(xyz.astype(np.float64) @ w.T + phi)
    grad = np.clip(rng.normal(0.5, 0.05, (xyz.shape[0], 2)), 0, 1)
    f = np.concatenate([rgb, grad], axis=1)
    if noise > 0:
        f = np.clip(f + rng.normal(0, noise, f.shape), 0, 1)
    return f


def colour_pair(n, pair_id=0):
    """Config 3: geometry of geometric_pair + 5-channel colour features."""
    rs = np.random.default_rng(1000 + pair_id)
    (x0, x1), (y0, y1), (z0, z1) = _box(n)
    pts = np.stack([rs.uniform(x0, x1, n), rs.uniform(y0, y1, n), rs.uniform(z0, z1, n)], axis=1)
    fsrc = colour_features(pts, np.random.default_rng(3000 + pair_id))
    rt = np.random.default_rng(2000 + pair_id)
    T = gt_motion()
    tgt = pts @ T[:3, :3].T + T[:3, 3] + rt.normal(0.0, 0.01, (n, 3))
    perm = rt.permutation(n)
    ftgt = np.clip(fsrc + np.random.default_rng(4000 + pair_id).normal(0, 0.01, fsrc.shape), 0, 1)
    return (pts.astype(np.float32), fsrc.astype(np.float32), tgt[perm].astype(np.float32),
            ftgt[perm].astype(np.float32), pts, perm)


def scene_colour_pair(n, pair_id=0):
    """The clustered street scene of scene_pair with the 5-channel colour features of config 3 on it (what a stereo camera
    frame looks like to the colour kernel: dense near objects AND per-point appearance).  Not a BASELINE.json config."""
    src, tgt, perm = scene_pair(n, pair_id)
    fsrc = colour_features(src, np.random.default_rng(8000 + pair_id))
    # the target's features: the source point's (same surface point seen again) + sensor noise, in the target's order
    ftgt = np.clip(fsrc + np.random.default_rng(9000 + pair_id).normal(0, 0.01, fsrc.shape), 0, 1)[perm]
    return src, fsrc.astype(np.float32), tgt, ftgt.astype(np.float32)


def checkerboard_labels(xyz, cell=2.0, flip=0.0, rng=None):
    """19-class one-hot labels from a 3-D checkerboard of `cell` m cells hashed to a class."""
    c = np.floor(xyz.astype(np.float64) / cell).astype(np.int64)
    h = (c[:, 0] * 73856093) ^ (c[:, 1] * 19349663) ^ (c[:, 2] * 83492791)
    cls = np.mod(h, NUM_CLASSES)
    if flip > 0 and rng is not None:
        sel = rng.random(xyz.shape[0]) < flip
        cls = np.where(sel, rng.integers(0, NUM_CLASSES, xyz.shape[0]), cls)
    onehot = np.zeros((xyz.shape[0], NUM_CLASSES), dtype=np.float32)
    onehot[np.arange(xyz.shape[0]), cls] = 1.0
    return onehot


def semantic_pair(n, pair_id=0):
    """Config 4: config-3 clouds + one-hot labels (2 % flips in the target)."""
    src, fsrc, tgt, ftgt, pts, perm = colour_pair(n, pair_id)
    lsrc = checkerboard_labels(pts)
    ltgt = checkerboard_labels(pts, flip=0.02, rng=np.random.default_rng(5000 + pair_id))[perm]
    return src, fsrc, lsrc, tgt, ftgt, ltgt


def to_colmajor16(T):
    """4x4 (row, col) array -> 16 floats in Eigen's column-major order."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def from_colmajor16(v):
    return np.asarray(v, dtype=np.float64).reshape(4, 4).T
