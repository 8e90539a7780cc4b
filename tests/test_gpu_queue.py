"""Batch queue (cvo_batch_open / _submit / _poll): a stream of pairs through a fixed number of in-flight slots, finished
pairs handing their slot to the next one.  Every pose must be bit-identical to a solo cvo_align of the same pair."""
import numpy as np
import pytest

import cases
from unified_cvo_amd import CvoGPU, CvoError

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (a.iterations, a.ret, a.final_ell, a.final_num_neighbors) == (b.iterations, b.ret, b.final_ell, b.final_num_neighbors) \
        and np.array_equal(a.transform, b.transform)


def test_queue_ragged_pairs_equal_solo():
    """14 ragged pairs through 4 slots (two sub-batch streams would need 8): every slot is refilled several times."""
    pairs = [cases.config2(n=1500 + 170 * p, pair_id=p) for p in range(14)]
    P = pairs[0][0]
    gpu = CvoGPU(params=P)
    src = [gpu.upload(p[1]) for p in pairs]
    tgt = [gpu.upload(p[2]) for p in pairs]
    limits = [150 + 20 * i for i in range(14)]   # every pair with its own iteration limit
    solo = [gpu.align(s, t, p[3], max_iterations=limits[i]) for i, (s, t, p) in enumerate(zip(src, tgt, pairs))]
    res = gpu.align_stream(src, tgt, [p[3] for p in pairs], slots=4, max_iterations=500, limits=limits)
    assert [r.ticket for r in res] == list(range(14))
    for a, b in zip(res, solo):
        assert _same(a, b)
    # the context is usable again after the queue has been closed
    again = gpu.align(src[0], tgt[0], pairs[0][3], max_iterations=limits[0])
    assert _same(again, solo[0])


def test_queue_incremental_submit_and_order():
    """Pairs submitted while others run; results come back in submission order whatever order they finish in (the
    semantic configuration: warm starts that end by themselves after a few hundred iterations among runs cut at 40)."""
    P, a, b, warm = cases.config4(n=3000)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(a), gpu.upload(b)
    lim = 900
    limits = [40 if (k % 3 == 0) else 0 for k in range(10)]
    solo = [gpu.align(da, db, warm, max_iterations=40), gpu.align(da, db, warm, max_iterations=lim)]
    q = gpu.open_queue(8, 3000, 3000, max_iterations=lim)
    with pytest.raises(CvoError):   # the queue owns the workspace
        gpu.align(da, db, warm, max_iterations=5)
    got = []
    for k in range(10):
        assert q.submit(da, db, warm, limits[k]) == k
        got.extend(q.poll(wait=0))
    while q.pending():
        got.extend(q.poll(wait=1))
    st = q.stats()
    q.close()
    assert [r.ticket for r in got] == list(range(10))
    assert st["refills"] == 10
    for k, r in enumerate(got):
        assert _same(r, solo[0 if k % 3 == 0 else 1]), k
    assert got[0].iterations == 40 < got[1].iterations < lim   # (the long ones end by themselves: dist < eps_2)


def test_queue_rejects_what_it_was_not_sized_for():
    P, a, b, init = cases.config2(n=1200)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(a), gpu.upload(b)
    q = gpu.open_queue(2, 1000, 1000)
    with pytest.raises(CvoError):
        q.submit(da, db, init)
    q.close()
    q = gpu.open_queue(2, 20000, 20000)   # sized for large clouds only: a small one would need more coefficient slices
    with pytest.raises(CvoError):
        q.submit(da, db, init)
    q.close()
    q = gpu.open_queue(2, 20000, 20000, min_source_points=1000)
    q.submit(da, db, init)
    r = q.poll(wait=2)
    q.close()
    assert len(r) == 1 and _same(r[0], gpu.align(da, db, init))
