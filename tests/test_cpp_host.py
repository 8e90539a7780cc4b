"""The dependency-free C++ host veneer (include/UnifiedCvo, host/): yaml reader on CPU; demo driver on GPU."""
import glob
import os
import subprocess
import sys
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


@pytest.mark.gpu
def test_api_surface_driver(tmp_path):
    """The rest of the cvo::CvoGPU surface through the C++ veneer: pcl (192-byte CvoPoint array) overloads of align /
    inner_product_gpu / function_angle, align(..., Association*) = the last executed iteration's matrix,
    inner_product_cpu and function_angle(is_gpu=false) (the reference's HOST function, against a dense numpy form)."""
    import np_reference as npr
    from unified_cvo_amd import CvoGPU, CvoPointCloud
    surf = os.path.join(HOST, "cvo_api_surface")
    sx, sr, tx, tr = cases.demo_clouds()
    _write_pcd(tmp_path / "source.pcd", sx, sr)
    _write_pcd(tmp_path / "target.pcd", tx, tr)
    yaml = os.path.join(cases.CONFIGS, "outdoor.yaml")
    max_iter, ell = 400, 0.9
    out = subprocess.check_output([surf, str(tmp_path / "source.pcd"), str(tmp_path / "target.pcd"), yaml, str(max_iter),
                                   str(ell), "2.5"], text=True)
    rows = {l.split()[0]: l.split()[1:] for l in out.strip().splitlines()}
    assert rows["ret"] == ["0", "0"] and rows["aos_equals_soa"] == ["1"]
    assert rows["stream_equals_align"] == ["1"]   # CvoGPU::align_stream (batch queue): poses bit-identical to align()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P = read_cvo_params_yaml(yaml)
    P.MAX_ITER = max_iter
    P.is_exporting_association = 1
    P.ell_init = 2.5       # (the yaml's 0.2 leaves the two demo clouds, 5.8 m apart, without a single pair)
    src, tgt = CvoPointCloud.from_xyzrgb(sx, sr), CvoPointCloud.from_xyzrgb(tx, tr)
    gpu = CvoGPU(params=P)
    g = gpu.align(src, tgt, np.eye(4))
    T_cpp = np.array([float(v) for v in rows["transform"]], np.float32).reshape(4, 4).T
    assert np.array_equal(T_cpp, g.transform)
    rp, col, val, kw, kr = gpu.align_association(src.num_points())
    a = dict(zip(rows["association"][0::2], rows["association"][1::2]))
    assert int(a["nnz"]) == len(col) > 100 and int(a["rows"]) == src.num_points() and int(a["cols"]) == tgt.num_points()
    assert int(a["source_inliers"]) == int((np.diff(rp) > 0).sum()) and int(a["target_inliers"]) == len(col)
    assert float(a["value_sum"]) == pytest.approx(float(val.astype(np.float64).sum()), rel=1e-7)
    assert int(a["col_checksum"]) == int((col.astype(np.int64) * (np.arange(len(col)) % 97 + 1)).sum())

    ip = gpu.inner_product_gpu(src, tgt, np.eye(4), ell)
    assert [np.float32(v) for v in rows["inner_product_gpu"]] == [np.float32(ip)] * 2
    fa = [gpu.function_angle(src, tgt, np.eye(4), ell, True), gpu.function_angle(src, tgt, np.eye(4), ell, False)]
    got = [np.float32(v) for v in rows["function_angle_gpu"]]
    assert got[0] == got[1] == np.float32(fa[0]) and got[2] == got[3] == np.float32(fa[1])

    xs, fs, _, _ = src.device_arrays()
    xt, ft, _, _ = tgt.device_arrays()
    ref0 = npr.inner_product_cpu(P, xs, xt, np.eye(4), ell, fs, ft)
    ref1 = npr.inner_product_cpu(P, xs, xt, np.linalg.inv(g.transform.astype(np.float64)), ell, fs, ft)
    got = [float(v) for v in rows["inner_product_cpu"]]
    assert got[0] == pytest.approx(ref0, rel=2e-5) and got[1] == pytest.approx(ref1, rel=2e-5) and ref1 > ref0 > 0
    fxx = npr.inner_product_cpu(P, xs, xs, np.eye(4), ell, fs, fs)
    fyy = npr.inner_product_cpu(P, xt, xt, np.eye(4), ell, ft, ft)
    got = [float(v) for v in rows["function_angle_cpu"]]
    assert got[0] == pytest.approx(ref0 / np.sqrt(len(xs)) / np.sqrt(len(xt)), rel=2e-5)
    assert got[1] == pytest.approx(ref0 / np.sqrt(fxx) / np.sqrt(fyy), rel=2e-5)


def _write_xyz_pcd(path, xyz):
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
                f"COUNT 1 1 1\nWIDTH {len(xyz)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(xyz)}\nDATA ascii\n")
        for p in xyz:
            f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")


@pytest.mark.gpu
def test_sharded_cpp_host_matches_batch(tmp_path):
    """cvo::CvoGPUSharded (host/cvo_gpu_sharded.cpp): one context + host thread per device, contiguous blocks of pairs,
    ONE ncclAllGather (RCCL) of the poses and one of the return codes.  On the single GPU of the test box the
    communicator has one rank; the gathered result must equal cvo_align_batch bit for bit (the C++ path north_star
    names for the 8-GPU mode; the driver's scaling run uses the Python harness)."""
    from unified_cvo_amd import CvoGPU, CvoPointCloud, synth
    shard = os.path.join(HOST, "cvo_align_sharded")
    yaml = os.path.join(cases.CONFIGS, "geometric_gpu.yaml")
    pairs, args = [], []
    for p in range(5):
        src, tgt, _ = synth.geometric_pair(900 + 150 * p, p)
        _write_xyz_pcd(tmp_path / f"s{p}.pcd", src)
        _write_xyz_pcd(tmp_path / f"t{p}.pcd", tgt)
        args += [str(tmp_path / f"s{p}.pcd"), str(tmp_path / f"t{p}.pcd")]
        pairs.append((CvoPointCloud.from_xyz(src), CvoPointCloud.from_xyz(tgt)))
    out = subprocess.check_output([shard, yaml, "250", "1"] + args, text=True)
    lines = [l.split() for l in out.strip().splitlines() if l.startswith(("pair ", "devices "))]  # (RCCL prints a banner)
    assert lines[-1][:4] == ["devices", "1", "pairs", "5"]
    P = cases.load_params("geometric_gpu")
    P.MAX_ITER = 250
    res = CvoGPU(params=P).align_batch([a for a, _ in pairs], [b for _, b in pairs], [np.eye(4)] * 5)
    for p, (l, r) in enumerate(zip(lines[:5], res)):
        assert l[:6] == ["pair", str(p), "device", "0", "ret", str(r.ret)]
        T = np.array([float(v) for v in l[7:23]], np.float32).reshape(4, 4).T
        assert np.array_equal(T, r.transform), p


def test_sharded_block_assignment():
    """pair p -> device p / ceil(n / n_devices): 512 pairs on 8 devices = 64 contiguous pairs per GPU (configs[4])."""
    per = lambda n, D: (n + D - 1) // D  # noqa: E731
    assert [p // per(512, 8) for p in (0, 63, 64, 511)] == [0, 0, 1, 7]
    from unified_cvo_amd import sharding
    for n, D in ((512, 8), (100, 8), (5, 2), (7, 1)):
        for r in range(D):
            lo, hi = sharding.shard_range(n, D, r)
            assert all(p // per(n, D) == r for p in range(lo, hi))


def test_cmake_package_builds_installs_and_is_consumable(tmp_path):
    """The drop-in packaging itself (CMakeLists.txt: targets cvo_gpu_img_lib / cvo_gpu_lidar_lib exported as
    UnifiedCvo::*, headers under include/UnifiedCvo-0.1): configure + build + install, then a 10-line consumer with
    find_package(UnifiedCvo) links UnifiedCvo::cvo_gpu_img_lib.  hipcc cross-compiles gfx950 without a GPU."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not (shutil.which("cmake") and os.path.exists(hipcc)):
        pytest.skip("cmake / hipcc not available")
    build, prefix = tmp_path / "build", tmp_path / "prefix"
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.check_call(["cmake", "-S", cases.ROOT, "-B", str(build), f"-DCMAKE_CXX_COMPILER={hipcc}",
                           f"-DCMAKE_INSTALL_PREFIX={prefix}"] + gen, stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(build), "-j", "8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--install", str(build)], stdout=subprocess.DEVNULL)
    assert (prefix / "include" / "UnifiedCvo-0.1" / "cvo" / "CvoGPU.hpp").exists()
    assert (prefix / "include" / "UnifiedCvo-0.1" / "cvo_hip.h").exists()
    cons = tmp_path / "consumer"
    cons.mkdir()
    (cons / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.18)\nproject(consumer LANGUAGES CXX)\nset(CMAKE_CXX_STANDARD 17)\n"
        "find_package(UnifiedCvo REQUIRED)\nadd_executable(consumer main.cpp)\n"
        "target_link_libraries(consumer PRIVATE UnifiedCvo::cvo_gpu_img_lib)\n")
    (cons / "main.cpp").write_text(
        '#include <cstdio>\n#include "cvo/CvoGPU.hpp"\n'
        "int main(int argc, char** argv) {\n"
        "  cvo::CvoParams p;\n  cvo_params_default(&p);\n"
        '  std::printf("%d %d %d %zu\\n", p.nearest_neighbors_max, NUM_CLASSES, FEATURE_DIMENSIONS, sizeof(cvo::CvoPoint));\n'
        "  if (argc > 1) { cvo::CvoGPU g(argv[1]); return g.get_params().MAX_ITER > 0 ? 0 : 1; }\n  return 0;\n}\n")
    subprocess.check_call(["cmake", "-S", str(cons), "-B", str(cons / "b"), f"-DCMAKE_CXX_COMPILER={hipcc}",
                           f"-DCMAKE_PREFIX_PATH={prefix}"] + gen, stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", str(cons / "b")], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([str(cons / "b" / "consumer")], text=True).split()
    assert out == ["512", "19", "5", "192"]      # defaults of CvoParams() + the PUBLIC compile definitions of the target


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


def test_shard_plan_partition_and_gather_arithmetic():
    """cvo::CvoGPUSharded's partition / gather index maths (include/UnifiedCvo/cvo/ShardPlan.hpp) driven WITHOUT devices
    for D = 8 and n in {512, 100, 5} (BASELINE.json configs[4]: 512 pairs -> 64 per GPU; a ragged tail; fewer pairs than
    GPUs), plus D = 1 / 3: contiguous blocks, every pair on exactly one device, read back from its own gather slot."""
    exe = os.path.join(HOST, "cvo_shard_plan_check")
    assert os.path.exists(exe), "build the host tools first (make -C host)"
    out = subprocess.check_output([exe, "8", "512", "100", "5", "0", "8", "9"], text=True).splitlines()
    assert out[0] == "D=8 n=512 per=64 counts=64,64,64,64,64,64,64,64"
    assert out[1] == "D=8 n=100 per=13 counts=13,13,13,13,13,13,13,9"
    assert out[2] == "D=8 n=5 per=1 counts=1,1,1,1,1,0,0,0"
    assert out[3] == "D=8 n=0 per=0 counts=0,0,0,0,0,0,0,0"
    for D in ("1", "3"):
        subprocess.check_call([exe, D, "1", "7", "64", "100"], stdout=subprocess.DEVNULL)
    # the Python host shards the same way (unified_cvo_amd/sharding.py)
    from unified_cvo_amd import sharding
    for n in (512, 100, 5):
        counts = [hi - lo for lo, hi in (sharding.shard_range(n, 8, r) for r in range(8))]
        line = [l for l in out if l.startswith(f"D=8 n={n} ")][0]
        assert line.endswith("counts=" + ",".join(map(str, counts)))


def test_library_load_sets_the_hardware_queue_hint():
    """The hardware-queue contract (include/cvo_hip.h, cvo_ctx_advice): loading the library puts GPU_MAX_HW_QUEUES=8 into
    the process environment - HIP reads it at the process's first HIP call - unless the variable is already set or
    CVO_NO_HW_QUEUE_HINT is.  No GPU needed: nothing here calls HIP."""
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from unified_cvo_amd import _capi; _capi.lib(); "
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; print(g(b'GPU_MAX_HW_QUEUES'))" % cases.ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "CVO_NO_HW_QUEUE_HINT")}
    run = lambda e: subprocess.check_output([sys.executable, "-c", code], env=e, text=True).strip()  # noqa: E731
    assert run(env) == "b'8'"
    assert run(dict(env, GPU_MAX_HW_QUEUES="3")) == "b'3'"          # an explicit choice is left alone
    assert run(dict(env, CVO_NO_HW_QUEUE_HINT="1")) == "None"


@pytest.mark.gpu
def test_hardware_queue_advice_channel():
    """cvo_ctx_advice: silent when the process has 8 hardware queues, a text (and one stderr line) below that."""
    code = ("import sys; sys.path.insert(0, %r); from unified_cvo_amd import CvoGPU; print('ADVICE[' + CvoGPU().advice() + ']')"
            % cases.ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "CVO_NO_HW_QUEUE_HINT", "CVO_QUIET")}
    ok = subprocess.run([sys.executable, "-c", code], env=env, text=True, capture_output=True)
    assert "ADVICE[]" in ok.stdout and "[cvo] advice" not in ok.stderr          # the load-time hint did its job
    low = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="2"), text=True, capture_output=True)
    assert "ADVICE[GPU_MAX_HW_QUEUES is 2" in low.stdout and "[cvo] advice: GPU_MAX_HW_QUEUES is 2" in low.stderr


@pytest.mark.gpu
def test_cpp_host_runs_the_headline_batch_at_the_c_abi_speed(tmp_path):
    """the C++ host of the multi-GPU mode (cvo::CvoGPUSharded, RCCL communicator alive, one
    device here) solves BASELINE.json's per-GPU headline workload - 64 resident 10k x 10k pairs, 2000 iterations - within
    10 % of cvo_align_batch called directly, in a process that sets NO environment variable (the library's load-time
    hardware-queue hint is all it gets), and returns bit-identical poses."""
    import time
    from unified_cvo_amd import CvoGPU
    shard = os.path.join(HOST, "cvo_align_sharded")
    yaml = os.path.join(cases.CONFIGS, "geometric_gpu.yaml")
    NP = 64
    pairs = [cases.config2(n=10000, pair_id=p) for p in range(NP)]
    args = []
    for p, (_, src, tgt, _) in enumerate(pairs):
        _write_xyz_pcd(tmp_path / f"s{p}.pcd", src.device_arrays()[0])
        _write_xyz_pcd(tmp_path / f"t{p}.pcd", tgt.device_arrays()[0])
        args += [str(tmp_path / f"s{p}.pcd"), str(tmp_path / f"t{p}.pcd")]
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "CVO_NO_HW_QUEUE_HINT")}
    out = subprocess.check_output([shard, "--bench", "5", yaml, "0", "1"] + args, text=True, env=env)
    lines = [l.split() for l in out.strip().splitlines()]
    bench = [l for l in lines if l and l[0] == "bench"][0]
    assert bench[:7] == ["bench", "devices", "1", "pairs", str(NP), "reps", "5"]
    ms_cpp = float(bench[bench.index("min") + 1])
    assert not [l for l in lines if l and l[0] == "advice"]
    P = pairs[0][0]
    gpu = CvoGPU(params=P)
    both = gpu.upload_many([a[1] for a in pairs] + [a[2] for a in pairs])
    inits = [np.eye(4, dtype=np.float32)] * NP
    gpu.align_batch(both[:NP], both[NP:], inits)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        res = gpu.align_batch(both[:NP], both[NP:], inits)
        best = min(best, time.perf_counter() - t0)
    ms_py = best * 1e3
    print(f"headline batch: C++ host (CvoGPUSharded, RCCL) {ms_cpp:.2f} ms, cvo_align_batch via ctypes {ms_py:.2f} ms")
    assert ms_cpp <= 1.10 * ms_py, (ms_cpp, ms_py)
    plines = [l for l in lines if l and l[0] == "pair"]
    assert len(plines) == NP
    for p, (l, r) in enumerate(zip(plines, res)):
        T = np.array([float(v) for v in l[7:23]], np.float32).reshape(4, 4).T
        assert np.array_equal(T, r.transform), p
