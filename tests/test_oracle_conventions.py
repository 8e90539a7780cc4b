"""What cannot be pinned must not matter: the oracle under every plausible floating-point convention.

The reference's exact rounding on its CUDA path is unknowable here (nvcc's fmad contraction choices, the association
of Eigen's unrolled 3-term reductions) and no reference build exists to pin it.  The default oracle - and, mirrored
statement for statement, the HIP kernels - fix one set of conventions (oracle/cvo_oracle.cpp header).  This test
rebuilds the oracle under the others (oracle/Makefile `variants`: fma in {dev, none, all} x sum order in {def, alt})
and shows, on every BASELINE.json configuration, that

  * the integer decisions of the loop (K, nnz, max_nnz, the ell schedule) are identical over the well-conditioned
    prefix of the trajectory,
  * the final pose agrees to the tolerance the north_star promises (1e-4; 2e-4 = 2 * min_step for the runs that
    end clamped at min_step),

i.e. the parity claim "GPU == oracle" carries over to "GPU == reference up to the stated tolerance" for any of these
choices the reference's compiler may have made.  scripts/convention_sweep.py prints the same numbers as a report
(profiles/r2/convention_sweep.json).
"""
import numpy as np
import pytest

import cases

N_TRACE = 400


def sweep(oracle, P, src, tgt, init, max_iterations=0):
    """Runs the align loop under the six conventions.  Returns {key: dict(transform, iterations, trace)}."""
    op = oracle.params_from(P)
    ox, oy = oracle.Cloud.from_pointcloud(src), oracle.Cloud.from_pointcloud(tgt)
    out = {}
    for fma in oracle.VARIANT_FMA:
        for s3 in oracle.VARIANT_SUM:
            with oracle.variant(fma, s3):
                out[f"{fma}_{s3}"] = oracle.align(op, ox, oy, init, trace_capacity=N_TRACE, trace_dense=N_TRACE,
                                                  max_iterations=max_iterations)
    return out


def summarize(res, base="dev_def"):
    """First iteration whose integer decisions differ from the default convention's, and the pose spread."""
    b = res[base]
    summary = {}
    for key, o in res.items():
        first = None
        for a, c in zip(o["trace"], b["trace"]):
            if (a.k, a.K, a.nnz, a.max_nnz) != (c.k, c.K, c.nnz, c.max_nnz) or a.ell != c.ell:
                first = a.k
                break
        summary[key] = dict(iterations=o["iterations"], first_decision_divergence=first,
                            pose_max_abs_vs_default=cases.max_abs_diff(o["transform"], b["transform"]))
    spread = max(cases.max_abs_diff(x["transform"], y["transform"]) for x in res.values() for y in res.values())
    return summary, spread


# (builder, kwargs, max_iterations, pose tolerance, iterations over which every convention must take the same integer
#  decisions).  Config 1 is compared at a fixed iteration count: its own stop is an accidental small step.
# The demo pair is the ill-conditioned one: at ell = 5.76 every row sits on K_max = 256 of 1080 targets, so which 256
# pairs a row keeps (first-K in index order) flips with single marginal hits, and once the decisions of two conventions
# separate (iteration 89-139) the trajectories drift apart by a few 1e-2 before the optimiser pulls them back together
# (7.8e-3 at their own stops, which themselves lie between iteration 6661 and 7312).  No two implementations that round
# differently anywhere can agree to 1e-4 on this pair beyond its first ~100 iterations - including the reference on two
# GPUs whose thrust reductions order their sums differently.  The 1e-4 claim for config 1 is therefore made at k = 100;
# the k = 1000 row only bounds the drift (and keeps this statement honest in profiles/r2/convention_sweep.json).
CASES = {
    "config1_demo_k100": (cases.config1, {}, 100, 1e-4, 80),
    "config1_demo_k1000": (cases.config1, {}, 1000, 0.1, 80),
    "config2_n1500": (cases.config2, dict(n=1500), 0, 2e-4, 10),
    "config3_n1000": (cases.config3, dict(n=1000), 0, 2e-4, 10),
    "config4_n1500": (cases.config4, dict(n=1500), 0, 1e-4, 10),
}

# The same sweep at BASELINE.json's literal sizes (scripts/convention_sweep.py --full -> profiles/r3/
# convention_sweep_full.json; minutes of CPU, so only config 2 at its literal 5k x 5k runs in the suite below).
CASES_FULL = {
    "config2_n5000": (cases.config2, dict(n=5000), 0, 2e-4, 10),
    "config2_shape_n10000": (cases.config2, dict(n=10000), 0, 2e-4, 10),
    "config3_n10000": (cases.config3, dict(n=10000), 0, 2e-4, 10),
    "config4_n10000": (cases.config4, dict(n=10000), 0, 1e-4, 10),
}
CASES["config2_n5000"] = CASES_FULL["config2_n5000"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_results_do_not_depend_on_the_unpinned_conventions(oracle, name):
    builder, kw, max_it, tol, n_same = CASES[name]
    P, src, tgt, init = builder(**kw)
    res = sweep(oracle, P, src, tgt, init, max_it)
    summary, spread = summarize(res)
    # the default build IS variant dev_def
    o = oracle.align(oracle.params_from(P), oracle.Cloud.from_pointcloud(src), oracle.Cloud.from_pointcloud(tgt), init,
                     max_iterations=max_it)
    assert np.array_equal(o["transform"], res["dev_def"]["transform"]) and o["iterations"] == res["dev_def"]["iterations"]
    for key, s in summary.items():
        assert s["first_decision_divergence"] is None or s["first_decision_divergence"] >= n_same, (name, key, s)
    assert spread <= tol, (name, spread, summary)
    # the conventions are really different builds: at least one of them changes some bit of the result
    assert any(s["pose_max_abs_vs_default"] > 0 or s["first_decision_divergence"] is not None for s in summary.values())
    # termination: runs that end by MAX_ITER do so under every convention; eps_2 stops may shift by a few iterations
    its = [s["iterations"] for s in summary.values()]
    if summary["dev_def"]["iterations"] in (P.MAX_ITER, max_it):
        assert len(set(its)) == 1
    else:
        assert max(its) - min(its) <= max(5, summary["dev_def"]["iterations"] // 50), its
