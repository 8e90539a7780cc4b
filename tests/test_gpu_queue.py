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


def test_queue_random_mix_equals_solo():
    """A soak in miniature: 36 submissions of slab and clustered pairs of three sizes with random iteration limits through
    six slots (two sub-batch streams would need eight: one stream, refills all the time), polled at random - every result,
    in submission order, bit-identical to the same pair solved alone with the same limit."""
    rng = np.random.default_rng(7)
    kinds = [cases.config2(n=1800, pair_id=1), cases.config2(n=2600, pair_id=2), cases.scene(n=2200, pair_id=3)]
    P = kinds[0][0]
    gpu = CvoGPU(params=P)
    dev = [(gpu.upload(k[1]), gpu.upload(k[2]), k[3]) for k in kinds]
    jobs = [(int(rng.integers(0, 3)), int(rng.choice([15, 60, 140, 260]))) for _ in range(36)]
    solo = {}
    for kind, lim in sorted(set(jobs)):
        s, t, T = dev[kind]
        solo[(kind, lim)] = gpu.align(s, t, T, max_iterations=lim)
    q = gpu.open_queue(6, 2600, 2600, min_source_points=1800, max_iterations=300)
    got = []
    for kind, lim in jobs:
        s, t, T = dev[kind]
        q.submit(s, t, T, lim)
        if rng.random() < 0.5:
            got.extend(q.poll(wait=int(rng.integers(0, 2))))
    while q.pending():
        got.extend(q.poll(wait=1))
    q.close()
    assert [r.ticket for r in got] == list(range(36))
    for r, (kind, lim) in zip(got, jobs):
        assert _same(r, solo[(kind, lim)]), (r.ticket, kind, lim)



def test_queue_lifetime_edges():
    """a context destroyed under an open queue releases the queue's device side and orphans the handle (later
    calls fail cleanly, close() only frees the host object); options cannot change under an open queue; a queue dropped
    without close() unlocks its context; a with-block closes on the way out; an empty stream is an empty result."""
    P, a, b, init = cases.config2(n=1200)
    gpu = CvoGPU(params=P)
    da, db = gpu.upload(a), gpu.upload(b)
    solo = gpu.align(da, db, init, max_iterations=60)
    assert gpu.align_stream([], [], []) == []
    with gpu.open_queue(2, 1200, 1200, max_iterations=60) as q:
        with pytest.raises(CvoError):
            gpu.set_option("SKIN", "0.5")      # graphs with chunks in flight would be destroyed
        q.submit(da, db, init)
        assert _same(q.poll(wait=2)[0], solo)
    assert _same(gpu.align(da, db, init, max_iterations=60), solo)   # closed by the with-block
    gpu.set_option("SKIN", None)
    q = gpu.open_queue(2, 1200, 1200, max_iterations=60)
    q.submit(da, db, init)
    del q                                      # dropped without close(): __del__ closes it
    import gc
    gc.collect()
    assert _same(gpu.align(da, db, init, max_iterations=60), solo)
    # context destroyed under an open queue with work in flight
    gpu2 = CvoGPU(params=P)
    ea, eb = gpu2.upload(a), gpu2.upload(b)
    q2 = gpu2.open_queue(2, 1200, 1200, max_iterations=400)
    q2.submit(ea, eb, init)
    L, h = gpu2.L, q2.handle
    L.cvo_ctx_destroy(gpu2.ctx)                # the raw C-ABI call, behind the wrapper's back
    gpu2.ctx = None
    import ctypes as C
    n = C.c_int()
    from unified_cvo_amd import _capi
    buf = (_capi.cvo_batch_result_t * 1)()
    assert L.cvo_batch_poll(h, 2, 1, buf, C.byref(n)) != 0 and n.value == 0
    assert L.cvo_batch_pending(h) == 0
    q2.close()                                 # frees the orphaned host object only
    ea.free()                                  # (clouds outlive their context: cvo_cloud_free only needs the device ordinal)
    eb.free()
