"""Oracle vs the committed golden traces (tests/golden/oracle_traces.json, scripts/make_golden.py)."""
import json
import os

import numpy as np
import pytest

import cases

with open(os.path.join(cases.GOLDEN, "oracle_traces.json")) as f:
    GOLD = {c["name"]: c for c in json.load(f)["cases"]}

BUILDERS = {"config1_demo_geometric_k1000": cases.config1, "config2_n2000": cases.config2,
            "config3_n2000": cases.config3, "config4_n2000": cases.config4}


@pytest.mark.parametrize("name", ["config1_demo_geometric_k1000", "config2_n2000", "config4_n2000"])
def test_oracle_reproduces_golden(oracle, name):
    g = GOLD[name]
    P, src, tgt, init = BUILDERS[name](**g["kwargs"])
    op = oracle.params_from(P)
    x, y = oracle.Cloud.from_pointcloud(src), oracle.Cloud.from_pointcloud(tgt)
    cap = min(300, g["iterations"]) if name != "config4_n2000" else 0
    r = oracle.align(op, x, y, init, trace_capacity=400, trace_dense=50, trace_every=100,
                     max_iterations=cap if cap else g["max_iterations"])
    rows = [t for t in g["trace"] if (not cap or t["k"] < cap)]
    got = {t.k: t for t in r["trace"]}
    for t in rows:
        a = got[t["k"]]
        assert (a.K, a.nnz, a.max_nnz) == (t["K"], t["nnz"], t["max_nnz"]), t["k"]
        assert a.ell == pytest.approx(t["ell"], rel=1e-6) and a.step == pytest.approx(t["step"], rel=1e-5)
        assert np.allclose(list(a.omega) + list(a.v), t["omega"] + t["v"], atol=2e-6)
        assert a.B == pytest.approx(t["B"], rel=1e-6, abs=1e-9)
    if not cap:
        assert r["iterations"] == g["iterations"] and r["ret"] == g["ret"]
        assert np.max(np.abs(r["transform"] - np.array(g["transform"]))) < 1e-6


def test_golden_config1_hits_neighbour_cap():
    rows = GOLD["config1_demo_geometric_k1000"]["trace"]
    assert rows[0]["K"] == 256 and rows[0]["max_nnz"] == 256  # rows sit on K_max
    assert rows[0]["ell"] == pytest.approx(5.76, abs=0.01)


def test_golden_config2_recovers_motion():
    g = GOLD["config2_n2000"]
    with open(os.path.join(cases.GOLDEN, "oracle_traces.json")) as f:
        gt_inv = np.array(json.load(f)["gt_inverse"])
    assert g["iterations"] == 2000
    assert np.max(np.abs(np.array(g["transform"]) - gt_inv)) < 2e-3
