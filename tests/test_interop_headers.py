"""The Eigen / PCL adaptor headers of the drop-in boundary go through a compiler ("the real signatures never
compile anywhere").  Eigen and PCL are not in this image: 30-line mocks under tests/mock_include stand in for the few
names the adaptors touch - it is the repo's own adaptor code that is being compiled and executed, nothing about upstream is
pinned by this.  Also: cvo::CvoPointCloud::label_at / feature_at / geometry_type_at (upstream CvoPointCloud.hpp:141-143)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import cases
from unified_cvo_amd import CvoPointCloud

INC = [os.path.join(cases.ROOT, "tests", "mock_include"), os.path.join(cases.ROOT, "include"),
       os.path.join(cases.ROOT, "include", "UnifiedCvo")]
CXX = shutil.which("g++") or shutil.which("c++")
FLAGS = ["-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror"] + [f"-I{p}" for p in INC]


@pytest.mark.skipif(CXX is None, reason="no host C++ compiler")
def test_eigen_interop_compiles_and_converts(tmp_path):
    exe = tmp_path / "interop_eigen_check"
    subprocess.check_call([CXX] + FLAGS + ["-o", str(exe), os.path.join(cases.ROOT, "tests", "cpp", "interop_eigen_check.cpp"),
                                           os.path.join(cases.ROOT, "host", "cvo_pointcloud.cpp")])
    out = subprocess.check_output([str(exe)], text=True)
    assert "interop ok" in out, out


@pytest.mark.skipif(CXX is None, reason="no host C++ compiler")
def test_pcl_interop_templates_instantiate(tmp_path):
    obj = tmp_path / "interop_pcl_check.o"
    subprocess.check_call([CXX] + FLAGS + ["-c", "-o", str(obj), os.path.join(cases.ROOT, "tests", "cpp", "interop_pcl_check.cpp")])
    syms = subprocess.check_output(["nm", "-C", str(obj)], text=True)
    # the three adaptor templates were instantiated and forward to the CvoPoint-array overloads of cvo::CvoGPU
    for name in ("cvo::CvoGPU::align(cvo::CvoPoint const*", "cvo::CvoGPU::inner_product_gpu(cvo::CvoPoint const*",
                 "cvo::CvoGPU::function_angle(cvo::CvoPoint const*"):
        assert name in syms, name


def test_python_mirror_row_accessors():
    pc = CvoPointCloud(5, 19)
    pc.reserve(3, 5, 19)
    assert pc.add_point(1, [1, 2, 3], np.arange(5) / 10, np.eye(19)[4], [0, 1]) == 0
    la, fa, ga = pc.label_at(1), pc.feature_at(1), pc.geometry_type_at(1)
    assert la.shape == (19,) and la[4] == 1 and la.sum() == 1 and np.allclose(fa, np.arange(5) / 10)
    assert np.array_equal(ga, [0, 1]) and np.array_equal(pc.geometry_type_at(0), [0, 0])
    la[4] = 7  # by value, as upstream's Eigen::VectorXf return
    assert pc.labels()[1, 4] == 1
