"""The dependency-free C++ host veneer (include/UnifiedCvo, host/): yaml reader on CPU; demo driver on GPU."""
import glob
import os
import subprocess
import warnings

import numpy as np
import pytest

import cases
from unified_cvo_amd import CvoParams, read_cvo_params_yaml
from unified_cvo_amd._capi import cvo_params_t

HOST = os.path.join(cases.ROOT, "host")
DUMP = os.path.join(HOST, "cvo_params_dump")
DEMO = os.path.join(HOST, "cvo_align_gpu_two_color_pcd")


def _dump(path=None):
    assert os.path.exists(DUMP), "build the host library first (make -C host)"
    out = subprocess.check_output([DUMP] + ([path] if path else []), text=True)
    kv = {}
    warns = []
    for line in out.splitlines():
        k, _, v = line.partition("=")
        if k == "warning":
            warns.append(v)
        else:
            kv[k] = float(v)
    return kv, warns


def test_cpp_defaults_match_python_defaults():
    kv, _ = _dump()
    d = CvoParams()
    for name, _ in cvo_params_t._fields_:
        assert kv[name] == pytest.approx(getattr(d, name), rel=1e-6, abs=1e-12), name


def test_cpp_reader_matches_python_reader_on_config_files():
    files = sorted(glob.glob(os.path.join(cases.CONFIGS, "*.yaml"))) + sorted(glob.glob("/root/reference/cvo_params/*.yaml"))
    assert len(files) >= 4
    for path in files:
        kv, warns = _dump(path)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = read_cvo_params_yaml(path)
        for name, _ in cvo_params_t._fields_:
            assert kv[name] == pytest.approx(getattr(p, name), rel=1e-6, abs=1e-12), (path, name)
        assert len(warns) == len(p.warnings), path


def _write_pcd(path, xyz, rgb):
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F U\n"
                f"COUNT 1 1 1 1\nWIDTH {len(xyz)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(xyz)}\nDATA ascii\n")
        for p, c in zip(xyz, rgb):
            u = (255 << 24) | (int(c[0]) << 16) | (int(c[1]) << 8) | int(c[2])
            f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {u}\n")


@pytest.mark.gpu
def test_demo_driver_matches_python_api(tmp_path):
    """README demo (config 1) through the C++ veneer == the same call through the Python mirror, bitwise."""
    from unified_cvo_amd import CvoGPU
    sx, sr, tx, tr = cases.demo_clouds()
    _write_pcd(tmp_path / "source.pcd", sx, sr)
    _write_pcd(tmp_path / "target.pcd", tx, tr)
    out = subprocess.check_output([DEMO, str(tmp_path / "source.pcd"), str(tmp_path / "target.pcd"),
                                   os.path.join(cases.CONFIGS, "outdoor.yaml")], text=True, cwd=tmp_path)
    rows = out.split("Transform is")[1].strip().splitlines()[:4]
    T_cpp = np.array([[float(v) for v in r.split()] for r in rows])
    P, src, tgt, init = cases.config1(geometric_only=False)
    g = CvoGPU(params=P).align(src, tgt, init)
    assert "ret 0" in out and g.ret == 0
    assert np.max(np.abs(T_cpp - g.transform)) < 5e-8  # printed with 8 decimals
    assert os.path.exists(tmp_path / "after_align.pcd")


@pytest.mark.gpu
def test_multiframe_edge_driver_matches_python_mirror(tmp_path):
    """cvo::CvoFrameGPU + cvo::BinaryStateGPU through the C++ veneer == the Python mirror."""
    from unified_cvo_amd import CvoGPU, CvoPointCloud, CvoFrameGPU, BinaryStateGPU
    edge_bin = os.path.join(HOST, "cvo_multiframe_edge")
    sx, sr, tx, tr = cases.demo_clouds()
    _write_pcd(tmp_path / "f1.pcd", sx, sr)
    _write_pcd(tmp_path / "f2.pcd", tx, tr)
    yaml = os.path.join(cases.CONFIGS, "outdoor.yaml")
    pose2 = np.array([[0.9998, -0.0175, 0.01, 0.2], [0.0175, 0.9998, 0.0, -0.1], [-0.01, 0.0002, 0.99995, 0.05]])
    K, ell = 64, 0.5
    out = subprocess.check_output([edge_bin, str(tmp_path / "f1.pcd"), str(tmp_path / "f2.pcd"), yaml, str(K), str(ell)] +
                                  [repr(float(v)) for v in pose2.reshape(-1)], text=True)
    rows = {l.split()[0]: l.split() for l in out.strip().splitlines()}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P = read_cvo_params_yaml(yaml)
    gpu = CvoGPU(params=P)
    c1, c2 = CvoPointCloud.from_xyzrgb(sx, sr), CvoPointCloud.from_xyzrgb(tx, tr)
    f1 = CvoFrameGPU(gpu, c1, np.hstack([np.eye(3), np.zeros((3, 1))]))
    f2 = CvoFrameGPU(gpu, c2, pose2)
    st = BinaryStateGPU(f1, f2, K, ell)

    def summary():
        m, i = st.mat, st.ind
        w = np.arange(1, i.shape[1] + 1)
        valid = i >= 0
        return int(st.nonzero_sum), i.shape[1], float(m[valid].astype(np.float64).sum()), int((i * w * valid).sum())

    st.update_inner_product()
    for tag in ("first", "second"):
        if tag == "second":
            st.update_inner_product()
        r = rows[tag]
        nz, k, vs, cs = summary()
        assert (int(r[2]), int(r[4]), int(r[8])) == (nz, k, cs), tag
        assert float(r[6]) == pytest.approx(vs, rel=1e-6)
    st.ell = st.ell * P.multiframe_ell_decay_rate if st.ell > P.multiframe_ell_min else st.ell
    f2.pose_vec[3] += 0.25
    f2.transform_pointcloud()
    st.update_inner_product()
    nz, k, vs, cs = summary()
    r = rows["moved"]
    assert (int(r[2]), int(r[4]), int(r[8])) == (nz, k, cs)
    assert float(r[10]) == pytest.approx(st.ell, rel=1e-6)


PCIO = os.path.join(HOST, "cvo_pointcloud_io")


def _read_pcd(path):
    lines = open(path).read().splitlines()
    hdr = {l.split()[0]: l.split()[1:] for l in lines[:11] if l and not l.startswith("#")}
    data = [l.split() for l in lines[lines.index("DATA ascii") + 1:]]
    return hdr, data


def test_pointcloud_text_format_and_writers(tmp_path):
    """the upstream text constructor (N F C header, 55 m filter) and the write_to_* family."""
    rs = np.random.default_rng(3)
    n, F, C = 40, 5, 19
    xyz = rs.uniform(-20, 20, (n, 3)).astype(np.float32)
    xyz[5] = (60, 0, 0)          # dropped: norm > 55
    xyz[17] = (33, 33, 30.5)     # dropped too (norm 55.7)
    feat = rs.uniform(0, 1, (n, F)).astype(np.float32)
    lab = rs.uniform(0, 1, (n, C)).astype(np.float32)
    src = tmp_path / "cloud_in.txt"
    with open(src, "w") as f:
        f.write(f"{n} {F} {C}\n")
        for i in range(n):
            f.write(" ".join(repr(float(v)) for v in list(xyz[i]) + list(feat[i]) + list(lab[i])) + "\n")
    out = subprocess.check_output([PCIO, str(src), str(tmp_path)], text=True).split()
    kv = dict(zip(out[0::2], out[1::2]))
    keep = np.linalg.norm(xyz.astype(np.float64), axis=1) <= 55
    assert int(kv["points"]) == int(keep.sum()) == n - 2
    assert (int(kv["features"]), int(kv["classes"]), int(kv["geometric_types"])) == (F, C, 0)
    assert float(kv["sum_xyz"]) == pytest.approx(float(xyz[keep].astype(np.float64).sum()), rel=1e-6)
    assert float(kv["sum_features"]) == pytest.approx(float(feat[keep].astype(np.float64).sum()), rel=1e-6)
    assert float(kv["sum_labels"]) == pytest.approx(float(lab[keep].astype(np.float64).sum()), rel=1e-6)

    hdr, data = _read_pcd(tmp_path / "xyz.pcd")
    assert hdr["FIELDS"] == ["x", "y", "z"] and hdr["POINTS"] == [str(n - 2)] and len(data) == n - 2
    assert np.allclose(np.array(data, np.float64), xyz[keep], rtol=1e-6)
    hdr, data = _read_pcd(tmp_path / "label.pcd")
    assert hdr["FIELDS"] == ["x", "y", "z", "label"]
    assert [int(d[3]) for d in data] == list(np.argmax(lab[keep], axis=1))
    hdr, data = _read_pcd(tmp_path / "intensity.pcd")
    assert hdr["FIELDS"] == ["x", "y", "z", "intensity"]
    assert np.allclose([float(d[3]) for d in data], feat[keep][:, 0], rtol=1e-6)
    hdr, data = _read_pcd(tmp_path / "color.pcd")
    packed = np.array([int(d[3]) for d in data], np.uint64)
    q = lambda v: np.minimum(255, (v.astype(np.float32) * np.float32(255)).astype(np.int32))  # noqa: E731
    assert np.array_equal((packed >> 16) & 255, q(feat[keep][:, 2]))   # r <- feature 2, as upstream
    assert np.array_equal(packed & 255, q(feat[keep][:, 0]))           # b <- feature 0
    txt = open(tmp_path / "cloud.txt").read().split()
    assert (int(txt[0]), int(txt[1])) == (n - 2, C)                    # upstream's two-number header
    assert len(txt) == 2 + (n - 2) * (3 + F + C)

    # the intensity PCD read back: one feature = the cvo_gpu_lidar_lib flavour
    (tmp_path / "again").mkdir()
    out2 = subprocess.check_output([PCIO, str(tmp_path / "intensity.pcd"), str(tmp_path / "again")], text=True).split()
    kv2 = dict(zip(out2[0::2], out2[1::2]))
    assert (int(kv2["points"]), int(kv2["features"]), int(kv2["classes"])) == (n - 2, 1, 0)
    assert float(kv2["sum_features"]) == pytest.approx(float(feat[keep][:, 0].astype(np.float64).sum()), rel=1e-5)


def test_pointcloud_raw_image_format(tmp_path):
    """read_cvo_pointcloud_from_file: "u v idepth features xyz labels" per point, no distance filter."""
    n, F, C = 7, 5, 3
    rs = np.random.default_rng(4)
    rows = rs.uniform(0, 100, (n, 3 + F + 3 + C))
    src = tmp_path / "raw.txt"
    with open(src, "w") as f:
        f.write(f"{n} {F} {C}\n")
        for r in rows:
            f.write(" ".join(f"{v:.6f}" for v in r) + "\n")
    out = subprocess.check_output([PCIO, "--raw", str(src), str(tmp_path)], text=True).split()
    kv = dict(zip(out[0::2], out[1::2]))
    assert (int(kv["points"]), int(kv["features"]), int(kv["classes"])) == (n, F, C)
    assert float(kv["sum_xyz"]) == pytest.approx(rows[:, 3 + F:3 + F + 3].sum(), rel=1e-5)
    assert float(kv["sum_features"]) == pytest.approx(rows[:, 3:3 + F].sum(), rel=1e-5)
