"""Second opinion on the LOOP-LEVEL logic of align_impl (CvoGPU.cu:1387-1531).

tests/np_reference.py::align_loop re-derives the whole optimiser loop from the reference text in float64 numpy /
scipy (dense N x M kernel matrix, expm / logm for the Lie-group steps, numpy.roots for the cubic, deques for the
indicator windows) and shares no code with oracle/.  The oracle's discrete events must coincide with it: the K
sequence (neighbour adaptation), the iterations at which ell decays (three-`if` indicator windows interacting with
ell_decay_start), the order of the eps / eps_2 exits and the stop iteration; the poses agree to float32 level.
A shared misreading of the reference's control flow by the oracle AND the HIP update kernel (which mirror each
other) would show up here.
"""
import numpy as np
import pytest

import cases
import np_reference as npr


def _arrays(pc):
    x, f, l, g = pc.device_arrays()
    return x, f, l, g


def _compare(oracle, P, src, tgt, init, max_iterations=0, nnz_rel=2e-3, pose_tol=1e-4):
    xs, fs, ls, gs = _arrays(src)
    xt, ft, lt, gt = _arrays(tgt)
    r = npr.align_loop(P, xs, xt, init, max_iterations, fx=fs, fy=ft, lx=ls, ly=lt, gx=gs, gy=gt)
    o = oracle.align(oracle.params_from(P), oracle.Cloud.from_pointcloud(src), oracle.Cloud.from_pointcloud(tgt), init,
                     trace_capacity=len(r["events"]) + 8, trace_dense=1 << 30, max_iterations=max_iterations)
    assert o["ret"] == r["ret"]
    assert o["iterations"] == r["iterations"], (o["iterations"], r["iterations"])
    assert len(o["trace"]) == len(r["events"])
    decays_np = [e["k"] for e in r["events"] if e["decayed"]]
    decays_or = [a.k for a, b in zip(o["trace"], o["trace"][1:]) if b.ell != a.ell]
    assert decays_or == decays_np[:len(decays_or)] and len(decays_np) - len(decays_or) <= 1  # (a decay at the last k shows in no later row)
    for e, t in zip(r["events"], o["trace"]):
        assert (e["k"], e["K"]) == (t.k, t.K), (e["k"], e["K"], t.K)
        assert np.float32(e["ell"]) == t.ell, (e["k"], e["ell"], t.ell)
        assert abs(int(e["nnz"]) - int(t.nnz)) <= max(3, nnz_rel * t.nnz), (e["k"], e["nnz"], t.nnz)
        # (the step is the root of a cubic whose constant term B is a cancelling sum near the optimum: the float32
        # per-pair terms of the reference against float64 here move it by percents, and once two trajectories jitter around
        # the optimum on different phases their steps are unrelated - single-iteration parity on shared state is what
        # tests/test_oracle_numpy.py checks; here the steps are compared over the well-conditioned start only)
        if e["k"] < 15:
            assert e["step"] == pytest.approx(t.step, rel=2e-3, abs=1e-7), (e["k"], e["step"], t.step)
    assert cases.max_abs_diff(r["transform"], o["transform"]) <= pose_tol
    return r, o


def test_whole_loop_config4_semantic_warm_start(oracle):
    """Config 4 (semantic, warm start) to its own dist < eps_2 stop: K adaptation, three ell decays, stop iteration."""
    P, src, tgt, init = cases.config4(n=500)
    r, o = _compare(oracle, P, src, tgt, init)
    assert 100 < r["iterations"] < P.MAX_ITER
    assert sum(e["decayed"] for e in r["events"]) >= 2
    assert len({e["K"] for e in r["events"]}) >= 2


def test_whole_loop_config1_300_point_prefix(oracle):
    """The demo pair cut to its first 300 points (rows sit on the K cap: ordered truncation) over a prefix that covers
    the first-frame decay start (ell_decay_start = 300) and the decays after it."""
    P, src, tgt, init = cases.config1()
    from unified_cvo_amd import CvoPointCloud
    sx, sf, _, sg = src.device_arrays()
    tx, tf, _, tg = tgt.device_arrays()
    a = CvoPointCloud.from_arrays(sx[:300], sf[:300], None, sg[:300])
    b = CvoPointCloud.from_arrays(tx[:300], tf[:300], None, tg[:300])
    P.nearest_neighbors_max = 64       # the cap bites on 300 targets as 256 does on 1080
    r, o = _compare(oracle, P, a, b, init, max_iterations=420, pose_tol=1e-4)
    assert r["events"][0]["max_nnz"] == 64
    assert any(e["decayed"] for e in r["events"])
    assert not any(e["decayed"] for e in r["events"] if e["k"] <= P.ell_decay_start)


def test_whole_loop_config2_clamped_steps(oracle):
    """Config 2 (geometric): the steps reach the min_step clamp and K collapses to a handful."""
    P, src, tgt, init = cases.config2(n=400)
    # (both runs end jittering by a few min_step around the optimum, on different phases: 5e-4)
    r, o = _compare(oracle, P, src, tgt, init, max_iterations=250, pose_tol=5e-4)
    assert any(abs(e["step"] - P.min_step) < 1e-9 for e in r["events"])
    assert r["events"][-1]["K"] < 20


def test_whole_loop_flow_vanishes(oracle):
    """Empty association: omega = v = 0 -> `break` with ret = -1 before any update (CvoGPU.cu:1454-1458)."""
    P, src, tgt, init = cases.config2(n=200)
    from unified_cvo_amd import CvoPointCloud
    far = CvoPointCloud.from_xyz(tgt.positions() + np.float32(100.0))
    r, o = _compare(oracle, P, src, far, init)
    assert r["ret"] == -1 and r["iterations"] == 0
    assert np.allclose(r["transform"], np.eye(4))
