"""Installs the outputs of scripts/profile_round.sh.

  on the GPU box (called by profile_round.sh):   install_profiles.py <round> --traffic
      writes profiles/kernel_traffic.json from the PMC passes (keyed on the hash of the kernel sources they were
      measured on) so that the bench run that follows reports roofline.traffic, and a copy next to the other outputs;
  here, afterwards:                               install_profiles.py <round>
      copies gpurun_out/<round>_prof/* into profiles/<round>/ and profiles/kernel_traffic.json.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r5"
src = os.path.join(ROOT, "gpurun_out", rnd + "_prof") + "/"
dst = os.path.join(ROOT, "profiles", rnd) + "/"
KT = os.path.join(ROOT, "profiles", "kernel_traffic.json")

if "--traffic" in sys.argv:
    from unified_cvo_amd import build as hipbuild
    raw = json.load(open(src + "pmc_summary_raw.json"))

    def hbm(prefix):
        k = max((q for q in raw if q.startswith(prefix)), key=lambda q: raw[q]["FETCH_SIZE"]["launches"])  # (template arguments vary)
        return int((2 * raw[k]["FETCH_SIZE"]["avg_per_launch"] + raw[k]["WRITE_SIZE"]["avg_per_launch"]) * 1024)

    # VALU wave-instructions of the whole batch call the counters were collected on = one step of bench.py: every kernel,
    # every launch (early-exit launches included)
    valu_total = int(sum(v["SQ_INSTS_VALU"]["launches"] * v["SQ_INSTS_VALU"]["avg_per_launch"] for k, v in raw.items()
                         if "SQ_INSTS_VALU" in v and "k_hold" not in k))
    # FP64 VALU instructions (half rate: 8 cycles per wave instruction)
    f64_total = int(sum(v[c]["launches"] * v[c]["avg_per_launch"] for k, v in raw.items() if "k_hold" not in k
                        for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64") if c in v))
    # per-kernel averages of the rocprofv3 kernel trace of the bench command (production kernels of the timed steps)
    trace = {}
    if os.path.exists(src + "bench_kernel_stats.csv"):
        for row in csv.DictReader(open(src + "bench_kernel_stats.csv")):
            nm = row["Name"]
            for key, pat in (("k_assoc", "cvo_dev::k_assoc<"), ("k_coeff", "cvo_dev::k_coeff<false>"), ("k_scan", "cvo_dev::k_scan<"),
                             ("k_list", "cvo_dev::k_list<"), ("k_prep", "cvo_dev::k_prep")):
                if pat in nm and key not in trace and "true>" not in nm.split("(")[0][-8:]:
                    trace[key] = round(float(row["AverageNs"]) / 1e3, 3)
    kt = {"points": 10000, "pairs": 16, "valu_wave_insts_per_step": valu_total, "valu_f64_wave_insts_per_step": f64_total or None,
          "trace_avg_launch_us": trace,
          "valu_wave_insts_per_launch": {k.replace("cvo_dev::", "").split("<")[0]: int(v["SQ_INSTS_VALU"]["avg_per_launch"])
                                         for k, v in raw.items() if "SQ_INSTS_VALU" in v and "k_hold" not in k},
          "hbm_bytes_per_launch": {"k_coeff": hbm("cvo_dev::k_coeff"), "k_assoc": hbm("cvo_dev::k_assoc<"), "k_scan": hbm("cvo_dev::k_scan<")},
          "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE counts 64 B per "
                        "128-B request; WRITE_SIZE uncalibrated)",
          "source": f"profiles/{rnd}/pmc_summary.json (separate --pmc passes for FETCH_SIZE and WRITE_SIZE over one cvo_align_batch of the "
                    "headline workload; average per launch of one 16-pair sub-batch)",
          "source_sha": hipbuild.source_hash()}
    json.dump(kt, open(KT, "w"), indent=1)
    shutil.copy(KT, src + "kernel_traffic.json")
    print(json.dumps(kt["hbm_bytes_per_launch"]))
    sys.exit(0)

os.makedirs(dst, exist_ok=True)
for f in ("bench_kernel_stats.csv", "configs.json", "stress.txt", "cpp_host_bench.txt", "upload_probe.txt", "scale_probe.txt",
          "perf_probe.txt", "pmc_summary.txt", "scene_probe.txt", "scene_kernel_stats.csv", "queue_probe.txt", "queue_probe.json",
          "overlap_probe.txt", "rowmax_probe.txt", "rank_rehearsal.txt", "perf_probe.json", "scene_batch.txt", "soak.txt"):
    if os.path.exists(src + f):
        shutil.copy(src + f, dst + f)
for f in ("bench_n1.json", "bench_under_rocprof.json", "bench_torchrun_n1.json"):
    if not os.path.exists(src + f):
        continue
    line = open(src + f).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open(dst + f, "w"), indent=1)
raw = json.load(open(src + "pmc_summary_raw.json"))
raw.pop("cvo_dev::k_hold", None)
cmd = ("rocprofv3 --pmc <counters> --output-format csv -- python one_batch.py (scripts/profile_round.sh: one cvo_align_batch of the "
       "headline workload, 64 x 10k x 10k, 2000 iterations; one pass per counter group, no trace domains; every launch covers one "
       "sub-batch of 16 pairs; averages over all launches of the run, including the early-exit launches of k_prep / k_scan / k_list / "
       "k_assoc_dense in iterations that do not rebuild; summarised on the GPU box by scripts/summarize_pmc.py)")
json.dump({"command": cmd, "kernels": raw}, open(dst + "pmc_summary.json", "w"), indent=0)
shutil.copy(src + "kernel_traffic.json", KT)
d = json.load(open(dst + "bench_n1.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "k_coeff", d["roofline"]["avg_launch_ms"], "traffic", d["roofline"]["traffic"])
for row in list(csv.DictReader(open(dst + "bench_kernel_stats.csv")))[:5]:
    print(row["Name"][:50], row["Calls"], row["AverageNs"])
