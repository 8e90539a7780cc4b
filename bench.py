#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: frame-pair align()/s (and ms/iteration) on 10k x 10k
geometric clouds, `pairs_per_gpu` independent pairs per GPU (BASELINE.json configs[4]: 64 per GPU),
one process per GPU, poses all-gathered with RCCL at the end of every step.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 2 --warmup 1

A "step" = one pass of the hot path over one batch: every rank solves its `pairs_per_gpu` resident
pairs to completion (cvo_align_batch) and the poses are gathered.  Inputs are resident in HBM before
the timed region (the PCIe-inclusive number is printed to stderr and recorded in DESIGN.md, never in
`value`).  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

# Per-pair kernel durations inside the loop (roofline.avg_launch_ms) come from the instrumented instantiation of the
# per-iteration kernels (cvo_align_opts_t.kernel_clock, ~3 % slower): ONE EXTRA step AFTER the timed region runs it - every
# timed step runs the production kernels.  CVO_KERNEL_CLOCK=0 in the environment switches the instrumented step off (the
# variable itself is cleared: set, it would instrument every call).
CLOCK_EXTRA_STEP = os.environ.pop("CVO_KERNEL_CLOCK", "1") != "0"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector
C_CULL_FLOPS = 8               # flops per pair test
N_SIMD = 256 * 4               # MI355X: 256 CUs x 4 SIMDs, each issuing one VALU wave-instruction per >= 4 cycles
SHADER_CLOCK_HZ = 2.4e9        # peak engine clock


def available_cpus():
    """Logical CPUs this process may actually use (the GPU box has a cgroup quota below nproc)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def _getenv_c(name):
    """The C environment (what HIP and cvo_process_hint_hw_queues see; os.environ is Python's start-up snapshot)."""
    import ctypes
    g = ctypes.CDLL(None).getenv
    g.restype = ctypes.c_char_p
    v = g(name.encode())
    return v.decode() if v else None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    # ONE line on stdout: libraries print there too (RCCL's five-line version banner at communicator creation), so file
    # descriptor 1 is pointed at stderr for the run and the JSON line goes to a private copy of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=10000)
    ap.add_argument("--pairs-per-gpu", type=int, default=64)
    ap.add_argument("--max-iterations", type=int, default=0, help="debug only: cap the optimiser loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="REHEARSAL of the N>1 branch on a 1-GPU box: every rank solves its shard on device 0 and the poses are "
                         "gathered with the gloo backend (RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = one whole align() on the CPU (about 3 s)")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the PCIe-inclusive pipeline leg")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the early-phase, 20k, colour / semantic batch and batch-queue legs (they run after the timed region)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cases
    from unified_cvo_amd import CvoGPU, sharding, _capi
    # The hardware-queue contract lives in the library (include/cvo_hip.h, cvo_ctx_advice): _capi.lib() calls
    # cvo_process_hint_hw_queues(), which puts GPU_MAX_HW_QUEUES=8 into the environment unless the caller chose a value.  HIP reads the variable at the process's
    # first HIP call, so the library is loaded HERE - after `import torch` (the process must end up with ONE HIP runtime:
    # torch's), before torch touches the GPU; the bench line reports what the context found (config.hardware_queues).
    _capi.lib()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1")
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    rehearse = args.rehearse_one_device
    if rehearse:
        local_rank = 0  # (every rank on the one device of the box)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if rehearse else dev  # where the collectives' buffers live (gloo: host)
    # host threads this rank may use (uploads, CPU baseline): the box's usable CPUs are shared by the local ranks
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    n_threads = max(1, available_cpus() // local_world)

    n = args.points
    B = args.pairs_per_gpu
    total_pairs = B * world
    lo, hi = sharding.shard_range(total_pairs, world, rank)
    P = cases.load_params("geometric_gpu")
    gpu = CvoGPU(params=P, device=local_rank)
    t_up = time.time()
    pairs = [cases.config2(n=n, pair_id=p) for p in range(lo, hi)]
    host_clouds = [(q[1], q[2]) for q in pairs]
    t_gen = time.time() - t_up
    t_up = time.time()
    both = gpu.upload_many([a for a, _ in host_clouds] + [b for _, b in host_clouds], threads=n_threads)
    src, tgt = both[:len(host_clouds)], both[len(host_clouds):]
    t_h2d = time.time() - t_up
    inits = [q[3] for q in pairs]
    # (after the first upload: RCCL's start-up threads otherwise compete with the upload pool for the host cores and
    # the one-off `uploaded in` figure doubles)
    # The step ends with an RCCL all-gather of the poses whatever the world size is: a plain `python bench.py --gpus 1`
    # forms a one-rank process group too, so that the timed region contains what `config.parallelism` says it does.
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    use_dist = True
    try:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    except Exception as e:  # noqa: BLE001
        if world > 1:
            raise
        use_dist = False
        log(f"[bench] one-rank RCCL process group failed to initialise ({e}): the step runs without the all-gather")
    pose_buf = torch.zeros(hi - lo, 16, dtype=torch.float32, device=coll_dev)
    kw = dict(max_iterations=args.max_iterations) if args.max_iterations > 0 else {}

    def step(clocked=False):
        # (cvo_align_opts_t.kernel_clock: the library keeps the graphs of both kernel instantiations)
        res = gpu.align_batch(src, tgt, inits, kernel_clock=clocked, **kw)
        if rehearse:  # (host buffers for gloo; column-major 4 x 4 like cvo_batch_poses_to_device writes them)
            pose_buf[:] = torch.from_numpy(np.stack([np.ascontiguousarray(r.transform.T).reshape(16) for r in res]))
        else:
            gpu.poses_to_device(pose_buf.data_ptr(), hi - lo)
        status = torch.tensor([r.ret for r in res], dtype=torch.int32, device=coll_dev)
        poses, stat = sharding.gather_poses(pose_buf, status, total_pairs, world, rank)
        return res, poses, stat

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The harness holds ~10^5 live Python objects (clouds, ctypes arrays); a generation-2 garbage collection in the
    # middle of a step costs ~35 ms of pure interpreter time.  Collect now and park what exists.
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        res, poses, stat = step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, coll_dev)
    if CLOCK_EXTRA_STEP:  # outside the timed region: the instrumented kernels (device clock per launch)
        step(clocked=True)
        step(clocked=True)
    iters = [r.iterations for r in res]
    loop_s = res[0].seconds

    out = None
    if rank == 0:
        aligns = total_pairs * args.steps
        value = aligns / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        mean_iters = float(np.mean(iters))
        # ms per optimiser iteration of one frame pair, with `B` pairs in flight on each GPU
        ms_per_iter_pair = elapsed / args.steps / max(mean_iters, 1.0) / B * 1e3
        # ---- roofline of the dominant kernel, timed live with HIP events on the ctx stream.
        # One optimiser iteration of a sub-batch is two launches: k_assoc (pass 1 of SURVEY.md 8(d): K1+K2+K3 over the
        # cached candidate lists) and k_coeff (pass 2: K4+K5, plus the scalar update in its last block); k_scan only
        # runs when a pair's candidate list has expired.  k_coeff holds the largest share of GPU time.
        tiles, rpt, tpt = gpu.debug_scan_stats()        # over the last measured step (all pairs, all list builds)
        # Durations of the two per-iteration kernels inside the loop (the instrumented step after the timed region): every block reports its
        # entry on the device's constant-rate counter, the block that finishes a pair's work closes the interval
        # (CVO_KERNEL_CLOCK, rate calibrated against HIP events) - first block in to last block out, per pair and launch,
        # i.e. what rocprofv3 --kernel-trace --stats averages for the same launches.  The launches sit inside hipGraphs,
        # where HIP events cannot be timed; the HIP-event figures below are replays of the same launches alone on the GPU.
        try:
            assoc_ms, coeff_ms, clocked = gpu.debug_kernel_clock()
        except Exception:  # CVO_KERNEL_CLOCK=0: fall back to the replayed launches below
            assoc_ms = coeff_ms = None
            clocked = 0
        builds, iters_total, cand_evals = gpu.debug_list_builds()
        n_groups, ppl = gpu.debug_last_geometry()       # the batch runs as n_groups sub-batches of ppl pairs
        # Kernel times depend on the optimiser state (lists shrink as ell decays): replay them on the state half way
        # through the trajectory, which is close to the average over the run that rocprofv3 --stats reports.
        mid_iters = max(1, int(mean_iters) // 2)
        gpu.align_batch(src, tgt, inits, max_iterations=mid_iters)
        assoc_alone_ms, coeff_alone_ms = gpu.debug_time_kernels(20)
        if not clocked:
            assoc_ms, coeff_ms = assoc_alone_ms, coeff_alone_ms
        scan_ms = gpu.debug_time_scan(20)
        bytes_pass = (n * 12 + n * 12) * ppl            # one pass over one iteration's inputs, geometric payload
        pair_tests_iter = 2.0 * float(n) * float(n)     # two passes over N x M per iteration and pair
        pair_rate = pair_tests_iter * iters_total * args.steps / elapsed if args.steps else 0.0
        executed_tests = float(tiles) * rpt * tpt + 2.0 * float(cand_evals)   # scan tiles + both passes over the lists
        executed_frac = executed_tests / (pair_tests_iter * max(float(iters_total), 1.0))
        # HBM traffic per launch comes from PMC passes (rocprofv3 --pmc, scripts/profile_round.sh), which cannot run
        # inside this process: profiles/kernel_traffic.json carries them together with the hash of the kernel sources they
        # were measured on, and is only reported while that hash matches the library that runs here (else null).
        traffic, traffic_note = {}, "profiles/kernel_traffic.json missing"
        valu_insts_per_step = None
        valu_f64_per_step = None
        trace_avg_us = {}
        tpath = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                from unified_cvo_amd import build as hipbuild
                tj = json.load(open(tpath))
                if tj.get("source_sha") != hipbuild.source_hash():
                    traffic_note = (f"stale: measured on kernel sources {tj.get('source_sha')}, running {hipbuild.source_hash()} "
                                    "(re-run scripts/profile_round.sh)")
                elif tj.get("points") != n or tj.get("pairs") != ppl:
                    traffic_note = "measured on another workload shape"
                else:
                    traffic = tj.get("hbm_bytes_per_launch", {})
                    traffic_note = tj.get("source", "")
                    valu_insts_per_step = tj.get("valu_wave_insts_per_step")
                    valu_f64_per_step = tj.get("valu_f64_wave_insts_per_step")
                    trace_avg_us = tj.get("trace_avg_launch_us", {})
            except Exception as e:  # noqa: BLE001
                traffic_note = f"unreadable: {e}"
        if not traffic:
            log(f"[bench] roofline.traffic = null ({traffic_note})")

        def kernel_entry(name, ms, share, alone_ms=None):
            # `achieved` is priced with the LONGER of the two in-loop durations we have for the kernel: the device clock of
            # this run (first block in -> last block out) and, while its kernel-source hash matches, the average of the
            # rocprofv3 kernel trace of the same command (profiles/<round>/bench_kernel_stats.csv; it includes dispatch and
            # end-of-kernel release, ~1 us more per launch)
            tr = trace_avg_us.get(name.split("::")[-1])
            priced = max(ms, tr * 1e-3) if tr else ms
            gbs = bytes_pass / (priced * 1e-3) / 1e9
            e = {"kernel": name, "achieved": round(gbs, 3), "frac": round(gbs / HBM_PEAK_GBS, 6),
                 "avg_launch_ms": round(priced, 5), "device_clock_avg_launch_ms": round(ms, 5),
                 "rocprof_trace_avg_launch_ms": round(tr * 1e-3, 5) if tr else None,
                 "launches_per_iteration_and_subbatch": share,
                 "traffic": traffic.get(name.split("::")[-1])}
            if alone_ms is not None:
                e["alone_on_gpu_launch_ms"] = round(alone_ms, 5)
            return e

        # the dominant kernel: the one the rocprofv3 trace has longer (when the committed trace belongs to these kernel
        # sources), else the one this run's device clock has longer
        ent_coeff = kernel_entry("cvo_dev::k_coeff", coeff_ms, 1.0, coeff_alone_ms)
        ent_assoc = kernel_entry("cvo_dev::k_assoc", assoc_ms, 1.0, assoc_alone_ms)
        dom, other = (ent_coeff, ent_assoc) if ent_coeff["avg_launch_ms"] >= ent_assoc["avg_launch_ms"] else (ent_assoc, ent_coeff)
        # memory-side traffic of a whole step from the PMC bytes per launch: every iteration of every sub-batch launches
        # both kernels once (rebuild kernels: 2 % of the iterations, not counted)
        step_traffic_gbs = None
        if ent_coeff["traffic"] and ent_assoc["traffic"]:
            launches = n_groups * mean_iters
            step_traffic_gbs = round((ent_coeff["traffic"] + ent_assoc["traffic"]) * launches / (elapsed / args.steps) / 1e9, 1)
        roofline = {
            "kernel": dom["kernel"], "bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac"], "traffic": dom["traffic"], "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": bytes_pass, "avg_launch_ms": dom["avg_launch_ms"],
            "device_clock_avg_launch_ms": dom["device_clock_avg_launch_ms"],
            "rocprof_trace_avg_launch_ms": dom["rocprof_trace_avg_launch_ms"],
            "alone_on_gpu_launch_ms": dom["alone_on_gpu_launch_ms"], "launches_clocked": int(clocked),
            "avg_launch_source": (("the longer of: device clock over every launch of one instrumented step run AFTER the timed "
                                   "region (first block in -> last block out; all timed steps run the production kernels), and "
                                   "the rocprofv3 --kernel-trace average of the same command committed under profiles/ "
                                   "(kernel-source hash checked)") if clocked
                                  else "HIP events around replayed launches, alone on the GPU"),
            # what binds, as measured (profiles/r5/ell8_experiment.txt): the 64-pair step is four chains of dependent launches
            # whose kernels wait on memory - 16 extra bytes per row cost 3 % although they saved 35 VALU instructions per row,
            # halving the ELL stream gains 3 % only at 128 pairs in flight.  "hbm" is the closer of the two roofs this field
            # can name; the honest description is "memory-side latency and bytes, far from either roof".
            "binds": ("the 64-pair step sits on the knee between a chain's latency (dependent launches, memory-side round trips: "
                      "profiles/r5/ell8_experiment.txt, profiles/r6/scale_probe.txt) and the chip's throughput, where VALU issue is ~80 % of "
                      "the time (profiles/r6/pmc_summary.txt); far from the HBM roof either way"),
            "pairs_per_launch": ppl, "sub_batches": n_groups, "timed_at_iteration": mid_iters,
            "step_traffic_gbs": step_traffic_gbs,
            "other_kernels": [other, kernel_entry("cvo_dev::k_scan", scan_ms, round(builds / max(iters_total, 1), 5))],
            # The path is an all-pairs accumulation with O(N+M) compulsory bytes; which resource binds
            # is a measurement.  `valu_issue_utilisation`: VALU wave-instructions of ONE step (SQ_INSTS_VALU summed over every
            # kernel and launch of one cvo_align_batch of this workload: PMC pass of scripts/profile_round.sh, keyed to the
            # kernel-source hash like `traffic`) x 4 cycles (a wave64 instruction occupies its 16-lane SIMD for at least four)
            # / (1024 SIMDs x 2.4 GHz x the measured step time): the share of the chip's VALU issue slots the step uses, a
            # lower bound where instructions take more than one pass.  Bounding-box culling and candidate-list reuse skip
            # almost all of the 2 N M algorithmic pair tests: `executed_fraction_of_pair_tests` is the share really run.
            "valu": {"valu_issue_utilisation": (round(valu_insts_per_step * 4.0 / (N_SIMD * SHADER_CLOCK_HZ * (elapsed / args.steps)), 4)
                                                if valu_insts_per_step else None),
                     # ... with FP64 instructions (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64) counted at their half rate: 8 cycles
                     "valu_issue_utilisation_fp64_weighted": (
                         round((valu_insts_per_step + valu_f64_per_step) * 4.0 / (N_SIMD * SHADER_CLOCK_HZ * (elapsed / args.steps)), 4)
                         if valu_insts_per_step and valu_f64_per_step else None),
                     "valu_wave_insts_per_step": valu_insts_per_step, "valu_f64_wave_insts_per_step": valu_f64_per_step,
                     "simds": N_SIMD, "clock_hz": SHADER_CLOCK_HZ,
                     "algorithmic_pair_tests_per_s": pair_rate,
                     "executed_fraction_of_pair_tests": round(executed_frac, 6),
                     "list_builds_per_iteration": round(builds / max(iters_total, 1), 5)},
        }
        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import pyoracle as po
            threads = n_threads
            po.set_num_threads(threads)
            op = po.params_from(P)
            ox, oy = po.Cloud.from_pointcloud(host_clouds[0][0]), po.Cloud.from_pointcloud(host_clouds[0][1])
            it = P.MAX_ITER if args.cpu_iters <= 0 else min(args.cpu_iters, P.MAX_ITER)
            po.align(op, ox, oy, inits[0], max_iterations=5)  # warm-up (page-in, thread pool)
            po.scan_seconds(reset=True)
            o = po.align(op, ox, oy, inits[0], max_iterations=it)   # by default ONE WHOLE align(), timed, nothing extrapolated
            scan_share = po.scan_seconds(reset=True) / max(o["seconds"], 1e-12)
            sec_per_iter = o["seconds"] / max(o["iterations"], 1)
            whole = o["iterations"] == int(round(mean_iters))
            cpu_value = 1.0 / o["seconds"] if whole else 1.0 / (sec_per_iter * mean_iters)
            cpu_baseline = {
                "value": cpu_value, "unit": "align/s", "cores": threads, "kind": "port",
                "ms_per_iter": sec_per_iter * 1e3, "align_seconds": o["seconds"] if whole else None,
                "scan_share": round(scan_share, 4),  # the K2 (association scan) share of the CPU time
                "sample": (f"one whole align() of pair 0 ({n}x{n}, {o['iterations']} iterations) with the CPU oracle (dense scan, "
                           f"OpenMP over rows), timed end to end" if whole else
                           f"first {o['iterations']} optimiser iterations of pair 0 ({n}x{n}) with the CPU oracle "
                           f"(dense scan, OpenMP over rows), extrapolated to the {mean_iters:.0f} iterations of a full align()"),
            }
            # "best-effort CPU" (SURVEY.md 8(d)): the same oracle with a uniform grid instead of the dense scan (the reference's
            # own CPU code uses a kd-tree).  The grid is built ONCE over the initial targets and queried with the row's point
            # mapped into that frame (an isometry; candidates still go through the exact test against the transformed
            # targets in ascending j: identical results), so no O(M) structure is rebuilt per iteration.
            po.set_grid(True)
            try:
                og, og_threads = None, threads
                for t in sorted({min(4, threads), threads}):  # its serial parts stop scaling at a few threads
                    po.set_num_threads(t)
                    po.scan_seconds(reset=True)
                    cand = po.align(op, ox, oy, inits[0])
                    cand["scan_share"] = po.scan_seconds(reset=True) / max(cand["seconds"], 1e-12)
                    if og is None or cand["seconds"] < og["seconds"]:
                        og, og_threads = cand, t
            finally:
                po.set_grid(False)
                po.set_num_threads(threads)
            cpu_baseline["best_effort"] = {
                "value": 1.0 / max(og["seconds"], 1e-9), "unit": "align/s", "cores": og_threads,
                "scan_share": round(og["scan_share"], 4),
                "ms_per_iter": og["seconds"] * 1e3 / max(og["iterations"], 1),
                "speedup_over_dense": round(o["seconds"] / max(og["seconds"], 1e-9), 2) if whole else None,
                "sample": f"one full align() of pair 0 ({og['iterations']} iterations) with the oracle's persistent uniform-grid variant "
                          f"(grid over the initial targets, rebuilt only when ell has shrunk by 2.5x; identical results, "
                          f"tests/test_oracle_numpy.py)"}
            # cross-check of the measured batch against the oracle on the same sample: the pose the timed steps returned
            # for pair 0 (whole align) / a rerun cut at the sample's iteration count
            g_T = res[0].transform if whole else gpu.align(src[0], tgt[0], inits[0], max_iterations=it).transform
            d = float(np.max(np.abs(g_T - o["transform"])))
            log(f"[bench] parity of the sample ({o['iterations']} iterations): pose max|d| = {d:.2e}")
            cpu_baseline["sample_parity_max_abs"] = d
        # ---- what a single align() costs (the reference's own use: frame-to-frame tracking, one pair at a time):
        # BASELINE.json configs 2, 3, 4 and the 10k shape of config 2, one pair in flight, hipEvent time of the loop
        single_pair = []
        # (these legs run the production kernels, like every timed step but the last)
        if world == 1 and args.max_iterations <= 0 and not args.no_single_pair:
            for name, builder, kw2 in (("config2: 5k x 5k xyz", cases.config2, dict(n=5000)),
                                       ("config2 shape at 10k x 10k xyz", cases.config2, dict(n=10000)),
                                       ("config3: 10k x 10k + 5-channel colour", cases.config3, dict(n=10000)),
                                       ("config4: 10k x 10k + colour + 19-class semantics, warm start", cases.config4, dict(n=10000))):
                Pc, a_, b_, init_ = builder(**kw2)
                gsp = CvoGPU(params=Pc, device=local_rank)
                da, db = gsp.upload(a_), gsp.upload(b_)
                gsp.align(da, db, init_, max_iterations=50)          # graphs instantiated, workspace allocated
                best = None
                for _ in range(3):
                    r = gsp.align(da, db, init_)
                    if best is None or r.seconds < best.seconds:
                        best = r
                single_pair.append({"config": name, "iterations": best.iterations, "ret": best.ret,
                                    "align_ms": round(best.seconds * 1e3, 4),
                                    "ms_per_iter": round(best.seconds * 1e3 / max(best.iterations, 1), 6)})
                da.free()
                db.free()
                gsp.close()
                log(f"[bench] single pair, {name}: {best.iterations} iterations, {best.seconds*1e3:.2f} ms "
                    f"({best.seconds*1e6/max(best.iterations,1):.2f} us/iteration)")
        # ---- the other two entry points of the path: inner_product_gpu (CvoGPU.cu:1719-1778) and function_angle
        # (CvoGPU.cu:1814-1846; exact = three inner products) - the loop-closure / overlap queries of the reference drivers.
        # One launch of k_overlap over resident clouds (64-row blocks against the target tiles their bounding volumes reach,
        # no candidate structure; the list chain of the loop - k_prep, k_scan, k_list, k_assoc [+ k_assoc_dense] - when a
        # row exceeds nearest_neighbors_max, and `list_chain_inner_product_gpu_ms` for comparison); wall time per call
        # through the Python binding, median of 9.
        overlap_queries = []
        if world == 1 and args.max_iterations <= 0 and not args.no_single_pair:
            for name, builder, kw2, extra_bytes in (("config2 shape at 10k x 10k xyz", cases.config2, dict(n=10000), 0),
                                                    ("config3: 10k x 10k + 5-channel colour", cases.config3, dict(n=10000), 5 * 4),
                                                    ("config4: 10k x 10k + colour + 19-class semantics", cases.config4, dict(n=10000), (5 + 19) * 4)):
                Pc, a_, b_, init_ = builder(**kw2)
                gq = CvoGPU(params=Pc, device=local_rank)
                da, db = gq.upload(a_), gq.upload(b_)

                def med(fn, reps=9):
                    fn()
                    ts = []
                    for _ in range(reps):
                        t1 = time.perf_counter()
                        fn()
                        ts.append(time.perf_counter() - t1)
                    return sorted(ts)[len(ts) // 2] * 1e3

                ip_ms = med(lambda: gq.inner_product_gpu(da, db, init_, Pc.ell_init))
                fa_ms = med(lambda: gq.function_angle(da, db, init_, Pc.ell_init, True))
                fe_ms = med(lambda: gq.function_angle(da, db, init_, Pc.ell_init, False))
                gq.set_option("IP_CHAIN", "1")
                chain_ms = med(lambda: gq.inner_product_gpu(da, db, init_, Pc.ell_init))
                gq.set_option("IP_CHAIN", None)
                one_pass = (12 + extra_bytes) * (a_.num_points() + b_.num_points())   # the loop's per-iteration bytes / 2
                overlap_queries.append({"config": name, "inner_product_gpu_ms": round(ip_ms, 4),
                                        "function_angle_approximate_ms": round(fa_ms, 4), "function_angle_exact_ms": round(fe_ms, 4),
                                        "list_chain_inner_product_gpu_ms": round(chain_ms, 4), "algorithmic_bytes_one_pass": one_pass,
                                        "achieved_gbs": round(one_pass / (ip_ms * 1e-3) / 1e9, 3),
                                        "frac_of_hbm_peak": round(one_pass / (ip_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)})
                log(f"[bench] overlap queries, {name}: inner_product_gpu {ip_ms:.3f} ms, function_angle {fa_ms:.3f} (approximate) / "
                    f"{fe_ms:.3f} (exact) ms per call")
                da.free()
                db.free()
                gq.close()
        # ---- the first 256 iterations of the headline batch (a third of the step: rows have ~13 candidates there, ~3 later)
        early_phase = None
        extra = world == 1 and args.max_iterations <= 0 and not args.no_extra_legs
        if extra:
            gpu.align_batch(src, tgt, inits, max_iterations=256)
            t_e = []
            for _ in range(3):
                t1 = time.perf_counter()
                gpu.align_batch(src, tgt, inits, max_iterations=256)
                t_e.append(time.perf_counter() - t1)
            early_phase = {"iterations": 256, "ms": round(min(t_e) * 1e3, 3), "share_of_step": round(min(t_e) / (elapsed / args.steps), 3)}
            log(f"[bench] first 256 iterations of the batch: {min(t_e)*1e3:.2f} ms ({100*min(t_e)/(elapsed/args.steps):.0f}% of a step)")

        def timed_batch(g, s_, t_, i_, reps=2, **kw3):
            g.align_batch(s_, t_, i_, **kw3)
            best_t, best_r = None, None
            for _ in range(reps):
                t1 = time.perf_counter()
                r_ = g.align_batch(s_, t_, i_, **kw3)
                dt_ = time.perf_counter() - t1
                if best_t is None or dt_ < best_t:
                    best_t, best_r = dt_, r_
            return best_t, best_r

        # ---- north_star: "synthetic N = 5k-20k clouds": the 20k x 20k shape, one pair in flight and a 16-pair batch
        shapes_20k = None
        if extra:
            p20 = [cases.config2(n=20000, pair_id=p) for p in range(16)]
            g20 = CvoGPU(params=P, device=local_rank)
            c20 = g20.upload_many([q[1] for q in p20] + [q[2] for q in p20], threads=n_threads)
            g20.align(c20[0], c20[16], p20[0][3], max_iterations=50)
            r1 = min((g20.align(c20[0], c20[16], p20[0][3]) for _ in range(2)), key=lambda r_: r_.seconds)
            tb, rb = timed_batch(g20, c20[:16], c20[16:], [q[3] for q in p20])
            shapes_20k = {"single_pair": {"iterations": r1.iterations, "align_ms": round(r1.seconds * 1e3, 3),
                                          "ms_per_iter": round(r1.seconds * 1e3 / max(r1.iterations, 1), 6)},
                          "batch_of_16": {"ms": round(tb * 1e3, 3), "align_per_s": round(16 / tb, 2),
                                          "iterations": int(np.mean([r_.iterations for r_ in rb])),
                                          "equals_single_pair": bool(np.array_equal(rb[0].transform, r1.transform))}}
            log(f"[bench] 20k x 20k: one pair {r1.seconds*1e3:.1f} ms ({r1.seconds*1e6/max(r1.iterations,1):.1f} us/iteration), "
                f"16-pair batch {tb*1e3:.1f} ms = {16/tb:.1f} align/s")
            for h in c20:
                h.free()
            g20.close()

        # ---- colour / semantic batches: 64 x config 3 and 64 x config 4 (BASELINE.json configs[2], [3] as batches), align/s,
        # one sample of each checked against the CPU oracle
        def feature_batch(builder, label, sample_its=300):
            prs = [builder(n=10000, pair_id=p) for p in range(B)]
            gf = CvoGPU(params=prs[0][0], device=local_rank)
            cl = gf.upload_many([q[1] for q in prs] + [q[2] for q in prs], threads=n_threads)
            tb, rb = timed_batch(gf, cl[:B], cl[B:], [q[3] for q in prs])
            its = [r_.iterations for r_ in rb]
            ent = {"pairs": B, "ms": round(tb * 1e3, 3), "align_per_s": round(B / tb, 2), "mean_iterations": float(np.mean(its)),
                   "us_per_pair_iteration": round(tb * 1e6 / max(sum(its), 1), 4)}
            if not args.no_cpu_baseline:
                from oracle import pyoracle as po
                po.set_num_threads(n_threads)
                o_ = po.align(po.params_from(prs[0][0]), po.Cloud.from_pointcloud(prs[0][1]), po.Cloud.from_pointcloud(prs[0][2]),
                              prs[0][3], max_iterations=sample_its)
                g_ = gf.align(cl[0], cl[B], prs[0][3], max_iterations=sample_its)
                ent[f"sample_parity_max_abs_at_{sample_its}_iterations"] = float(np.max(np.abs(g_.transform - o_["transform"])))
                ent["sample_iterations"] = [int(g_.iterations), int(o_["iterations"])]
            log(f"[bench] batch of {B} x {label}: {tb*1e3:.1f} ms = {B/tb:.1f} align/s ({np.mean(its):.0f} iterations each)")
            for h in cl:
                h.free()
            gf.close()
            return ent

        batch_colour = feature_batch(cases.config3, "config 3 (10k x 10k + colour)") if extra else None
        batch_semantic = feature_batch(cases.config4, "config 4 (10k x 10k + colour + one-hot semantics, warm start)") if extra else None
        # ... and 64 clustered pairs (NOT a BASELINE shape: tests/synth.scene_pair - ground, facades, small dense objects, local
        # density varying by more than 100x; 13 x the pair tests of the uniform slab, most of them in the wave-per-row kernels)
        batch_clustered = feature_batch(cases.scene, "clustered street scene (10k x 10k xyz, not a BASELINE shape)", 100) if extra else None
        # ... the same scene with colour features (what "KITTI-stereo-shaped" means for density AND appearance)
        batch_clustered_colour = feature_batch(cases.scene_colour, "clustered street scene + 5-channel colour (not a BASELINE shape)", 100) if extra else None

        # ---- batch queue (cvo_batch_open / _submit / _poll): the 8-GPU headline's whole work list - 512 pairs - on ONE GPU
        # through 128 in-flight slots, and a mixed queue (three pairs in four stop after 300 iterations, like warm-started
        # tracking frames, the fourth runs its 2000) against the cost-weighted ideal of uniform fixed batches
        batch_queue = None
        if extra:
            def run_queue(slots, limits):
                q_ = gpu.open_queue(slots, n, n)
                t1 = time.perf_counter()
                for k_, lim_ in enumerate(limits):
                    q_.submit(src[k_ % B], tgt[k_ % B], inits[k_ % B], lim_)
                got = []
                while q_.pending():
                    got.extend(q_.poll(wait=2))
                dt_ = time.perf_counter() - t1
                st_ = q_.stats()
                q_.close()
                return dt_, got, st_
            run_queue(128, [0] * 128)  # graphs of the queue's sub-batch geometry
            tq, gq_, sq_ = run_queue(128, [0] * 512)
            same = all(np.array_equal(r_.transform, res[k_ % B].transform) and r_.iterations == res[k_ % B].iterations for k_, r_ in enumerate(gq_))
            t300, _ = timed_batch(gpu, src, tgt, inits, max_iterations=300)
            limits = [0 if k_ % 4 == 0 else 300 for k_ in range(256)]
            tm, gm_, sm_ = run_queue(128, limits)
            ideal = (64 * (elapsed / args.steps) + 192 * t300) / B     # uniform batches of 64 of either kind, back to back
            batch_queue = {"uniform_512_pairs_128_slots": {"ms": round(tq * 1e3, 2), "align_per_s": round(512 / tq, 1),
                                                           "bit_identical_to_fixed_batch": bool(same), **sq_},
                           "mixed_256_pairs_128_slots": {"ms": round(tm * 1e3, 2), "short_iterations": 300, "long_iterations": int(mean_iters),
                                                         "ideal_ms_uniform_batches_of_64": round(ideal * 1e3, 2),
                                                         "fraction_of_cost_weighted_ideal": round(ideal / tm, 3),
                                                         "fraction_of_iteration_weighted_ideal": round(
                                                             (sum(r_.iterations for r_ in gm_) / tm) / (B * mean_iters / (elapsed / args.steps)), 3),
                                                         **sm_}}
            log(f"[bench] batch queue: 512 pairs through 128 slots {tq*1e3:.1f} ms = {512/tq:.1f} align/s (poses "
                f"{'identical' if same else 'DIFFERENT'}); mixed 300 / {mean_iters:.0f}-iteration queue {tm*1e3:.1f} ms = "
                f"{ideal/tm:.2f} of the cost-weighted ideal")
        upload_ms_per_cloud = t_h2d * 1e3 / max(2 * len(host_clouds), 1)
        # PCIe-inclusive, as a frame pipeline runs it: while the GPU solves batch k the host threads order and upload
        # batch k + 1 (cvo_cloud_upload_many on its own streams); every step pays for fresh inputs, the timed region
        # is the steady state of that pipeline.  `sequential_value` is upload-then-solve without any overlap.
        pcie_inclusive = {"sequential_value": aligns / (elapsed + args.steps * t_h2d), "unit": "align/s",
                          "upload_ms_per_cloud": round(upload_ms_per_cloud, 4), "upload_threads": n_threads}
        if world == 1 and args.max_iterations <= 0 and not args.no_pipeline:
            import threading
            all_host = [a for a, _ in host_clouds] + [b_ for _, b_ in host_clouds]
            nxt = {}

            def prefetch():
                nxt["clouds"] = gpu.upload_many(all_host, threads=max(1, n_threads - 1))

            cur = gpu.upload_many(all_host, threads=n_threads)
            n_pipe = max(args.steps, 3)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(n_pipe):
                th = threading.Thread(target=prefetch)
                th.start()
                gpu.align_batch(cur[:B], cur[B:], inits)
                th.join()
                for h in cur:
                    h.free()
                cur = nxt["clouds"]
            torch.cuda.synchronize()
            t_pipe = time.perf_counter() - tp0
            for h in cur:
                h.free()
            pcie_inclusive.update({"value": B * n_pipe / t_pipe, "ms_per_step": round(t_pipe / n_pipe * 1e3, 3), "steps": n_pipe,
                                   "fraction_of_resident_rate": round(B * n_pipe / t_pipe / value, 4)})
            log(f"[bench] PCIe-inclusive pipeline: {B * n_pipe / t_pipe:.1f} align/s ({t_pipe / n_pipe * 1e3:.1f} ms per step, "
                f"{100.0 * B * n_pipe / t_pipe / value:.0f}% of the resident rate)")
        pcie_inclusive["note"] = ("every step uploads its 2 x pairs_per_gpu clouds afresh (spatial ordering + one H2D copy "
                                  "per cloud); `value` of the bench line has them resident, as registration_seconds of the "
                                  "reference excludes its H2D copies")
        out = {
            "metric": "frame-pair align()/sec, 10k x 10k geometric clouds", "value": value, "unit": "align/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_pair_iteration_amortised": ms_per_iter_pair, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4] shape: {B} independent {n}x{n} xyz-only frame pairs per GPU "
                                   f"(seeds 1000+p / 2000+p), cvo_geometric_params_gpu.yaml, identity init, "
                                   f"{mean_iters:.0f} optimiser iterations per align()",
                       "pairs_per_gpu": B, "points": n, "iterations_per_align": mean_iters,
                       "parallelism": (f"REHEARSAL, not a measurement: {world} ranks on ONE device, pairs sharded {B}/rank, poses gathered "
                                       f"with gloo (RCCL refuses two ranks on one device)" if rehearse else
                                       f"pairs sharded {B}/GPU over {world} GPU(s); one RCCL all-gather of poses per step "
                                       f"(process group of {world} rank(s))" if use_dist else
                                       f"{B} pairs on 1 GPU; NO collective ran (the one-rank process group failed to initialise)"),
                       "timed_steps": f"{args.steps} steps, production kernels (the instrumented step behind roofline.avg_launch_ms runs after the timed region)",
                       "host_threads_per_rank": n_threads,
                       "hardware_queues": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES") or _getenv_c("GPU_MAX_HW_QUEUES"),
                                           "advice": gpu.advice()}},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "single_pair": single_pair,
            "overlap_queries": overlap_queries, "pcie_inclusive": pcie_inclusive,
            "early_phase": early_phase, "shapes_20k": shapes_20k, "batch_colour": batch_colour, "batch_semantic": batch_semantic, "batch_clustered": batch_clustered, "batch_clustered_colour": batch_clustered_colour,
            "batch_queue": batch_queue,
        }
        if rehearse:
            # the gathered table against this rank's own results and against a solo solve of the LAST pair (another rank's)
            mine = np.stack([np.ascontiguousarray(r.transform.T).reshape(16) for r in res])
            pl = cases.config2(n=n, pair_id=total_pairs - 1)
            solo = gpu.align(pl[1], pl[2], pl[3], **kw)
            out["rehearsal"] = {"backend": "gloo", "ranks": world, "device_of_every_rank": 0,
                                "gathered_rows": int(poses.shape[0]),
                                "own_shard_in_place": bool(np.array_equal(poses[lo:hi].cpu().numpy(), mine)),
                                "last_pair_of_last_rank_equals_solo_solve": bool(np.array_equal(
                                    poses[total_pairs - 1].cpu().numpy(), np.ascontiguousarray(solo.transform.T).reshape(16)))}
        h2d_rate = (2 * n * 16 * 1.0) * B / max(t_h2d, 1e-9) / 1e9
        log(f"[bench] inputs: generated in {t_gen:.2f}s, uploaded in {t_h2d:.3f}s ({h2d_rate:.2f} GB/s incl. host-side k-d ordering on {n_threads} threads); "
            f"PCIe-inclusive rate = {aligns / (elapsed + args.steps * t_h2d):.2f} align/s")
        log(f"[bench] loop {loop_s:.4f}s/step on rank 0; per launch of {ppl} pairs: k_assoc {assoc_ms*1e3:.1f} us "
            f"({assoc_alone_ms*1e3:.1f} alone on the GPU), k_coeff {coeff_ms*1e3:.1f} us ({coeff_alone_ms*1e3:.1f} alone), k_scan {scan_ms*1e3:.1f} us (runs in {100.0*builds/max(iters_total,1):.1f}% of the "
            f"iterations); {pair_rate/1e12:.1f} T algorithmic pair-tests/s, {100*executed_frac:.3f}% of them executed")
        assert int(stat.abs().sum().item()) == 0, "some align() returned -1"
        print(json.dumps(out), file=result_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
