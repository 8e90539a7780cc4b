#!/usr/bin/env python
"""Instruction-level profile of the shipped gfx950 kernels, CPU only (hipcc cross-compiles).

For every kernel named on the command line (default: the two per-iteration kernels of the headline workload and the
rebuild trio) the device assembly of the library's own build flags is cut into its loops (the compiler's
"Loop Header" comments and the backward branch that closes each loop) and every loop gets an instruction histogram:
VALU f32 / f64 / integer / moves, transcendental (quarter rate), SALU, vector memory, LDS, branches, waits.
The listing of a loop body can be printed with --dump.  VGPR / SGPR / occupancy / LDS / scratch per kernel come from
-Rpass-analysis=kernel-resource-usage (scripts/kernel_resources.py).

usage: isa_loops.py [--dump] [--out FILE] [--flags "..."] [kernel-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_cvo_amd import build as B  # noqa: E402

DEFAULT = ["k_assoc<unsigned short, 64, 0, false>", "k_coeff<false>", "k_list<unsigned short, 64>", "k_scan<2>",
           "k_prep", "k_assoc<unsigned short, 64, 2, false>", "k_assoc<unsigned short, 64, 3, false>"]


def classify(m):
    if m.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_trans"
    if m.startswith(("v_mov", "v_pk_mov", "v_accvgpr", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane",
                     "v_permlane", "v_swap")) or "_dpp" in m:
        return "valu_move"
    if m.startswith("v_") and "f64" in m:
        return "valu_f64"
    if m.startswith("v_") and ("f32" in m or "f16" in m):
        return "valu_f32"
    if m.startswith("v_"):
        return "valu_int"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if m.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("ds_"):
        return "lds"
    return "other"


ORDER = ["valu_f32", "valu_f64", "valu_int", "valu_move", "valu_trans", "salu", "smem", "vmem", "lds", "branch", "wait",
         "other"]


def device_asm(extra_flags):
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = os.path.join(tempfile.gettempdir(), "_cvo_isa.s")
    cmd = [B._hipcc()] + flags + extra_flags + ["-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                                                B.sources()[0]]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read().splitlines()


def kernels(lines):
    """name -> (first, last) line index of the kernel's code."""
    starts = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if m and i + 1 < len(lines):
            starts.append((i, m.group(1)))
    out = {}
    for (i, sym) in starts:
        end = i
        for j in range(i + 1, len(lines)):
            if lines[j].strip().startswith("s_endpgm"):
                end = j
            if lines[j].startswith(".Lfunc_end"):
                break
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name).split("(")[0].replace("cvo_dev::", "")
        out[name] = (i, end)
    return out


def instructions(lines, a, b):
    """(line index, mnemonic) of every instruction in lines[a:b]."""
    res = []
    for i in range(a, b + 1):
        s = lines[i].split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            continue
        res.append((i, s.split()[0]))
    return res


def loops(lines, a, b):
    """Loops of lines[a:b] from the compiler's basic-block annotations ("=>This [Inner] Loop Header: Depth=d",
    "in Loop: Header=BBx_y Depth=d", "Parent Loop BBx_y Depth=d"): header label -> (depth, parent header, line indices of
    its own blocks).  A block belongs to the innermost loop it is annotated with; nested loops are added to their
    parents by the caller."""
    info = {}     # header -> dict(depth, parent, lines)
    cur = None    # loop of the current block (None = straight-line code)
    pending_parent = None
    for i in range(a, b + 1):
        l = lines[i]
        is_block = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.\d+:", l)
        if is_block:
            lab = re.match(r"^\.(LBB\d+_\d+):", l)
            m_in = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            m_hd = re.search(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)", l)
            m_par = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", l)
            pending_parent = None
            if m_in:
                cur = m_in.group(1)
                info.setdefault(cur, dict(depth=int(m_in.group(2)), parent=None, lines=[]))
            elif m_hd and lab:
                cur = lab.group(1)[1:] if lab.group(1).startswith("L") else lab.group(1)
                info.setdefault(cur, dict(depth=int(m_hd.group(2)), parent=None, lines=[]))
            elif m_par and lab:  # header of a nested loop: "Parent Loop" lines first, the header line follows
                cur = lab.group(1)[1:]
                pending_parent = m_par.group(1)
                info.setdefault(cur, dict(depth=0, parent=pending_parent, lines=[]))
            else:
                cur = None
            continue
        if cur is not None and pending_parent is not None:
            m_hd = re.search(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)", l)
            m_par = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", l)
            if m_par:  # (deeper nests list every ancestor; the last one named is the direct parent)
                info[cur]["parent"] = m_par.group(1)
                continue
            if m_hd:
                info[cur]["depth"] = int(m_hd.group(2))
                pending_parent = None
                continue
        if cur is not None:
            info[cur]["lines"].append(i)
    # blocks of nested loops also run inside their parents
    for h, d in info.items():
        if d["parent"] is None and d["depth"] > 1:
            # "in Loop: Header=X Depth=2" blocks seen before X's header line: the parent is filled in when it appears
            pass
    res = []
    for h, d in info.items():
        own = list(d["lines"])
        kids = [k for k, kd in info.items() if kd["parent"] == h]
        allk = list(kids)
        while kids:
            k = kids.pop()
            more = [q for q, qd in info.items() if qd["parent"] == k]
            allk.extend(more)
            kids.extend(more)
        total = own + [i for k in allk for i in info[k]["lines"]]
        if total:
            res.append((h, sorted(total), d["depth"], len(allk)))
    res.sort(key=lambda r: r[1][0])
    return res


def histogram(ins):
    h = {}
    for _, m in ins:
        c = classify(m)
        h[c] = h.get(c, 0) + 1
    return h


def main():
    args = sys.argv[1:]
    dump = "--dump" in args
    out_path = None
    extra = []
    names = []
    it = iter(args)
    for a in it:
        if a == "--dump":
            continue
        if a == "--out":
            out_path = next(it)
        elif a == "--flags":
            extra = next(it).split()
        else:
            names.append(a)
    names = names or DEFAULT
    lines = device_asm(extra)
    ks = kernels(lines)
    res_tab = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_resources.py")], capture_output=True,
                             text=True, env=dict(os.environ, CVO_EXTRA_HIPCC_FLAGS=" ".join(extra))).stdout.splitlines()
    out = []
    out.append(f"# kernel-source hash {B.source_hash()}  flags: {' '.join(B.HIPCC_FLAGS + extra)}")
    for want in names:
        for name, (a, b) in ks.items():
            if want not in name:
                continue
            ins = instructions(lines, a, b)
            h = histogram(ins)
            out.append("")
            out.append(f"== {name}: {len(ins)} instructions  " + "  ".join(f"{c} {h[c]}" for c in ORDER if c in h))
            for r in res_tab:
                if r.split(" VGPRs")[0].strip() == name:
                    out.append("   " + " ".join(r.split()[len(name.split()):]))
            for (lab, idx, depth, nkids) in loops(lines, a, b):
                li = [x for i in idx for x in instructions(lines, i, i)]
                lh = histogram(li)
                valu = sum(lh.get(c, 0) for c in ORDER[:5])
                first, last = idx[0], idx[-1]
                out.append(f"   loop {lab:10s} depth {depth}{' (+%d nested)' % nkids if nkids else ''} lines {first - a:5d}-{last - a:5d}: "
                           f"{len(li):4d} instructions, {valu:4d} VALU  |  " + "  ".join(f"{c} {lh[c]}" for c in ORDER if c in lh))
                mn = {}
                for _, m in li:
                    mn[m] = mn.get(m, 0) + 1
                top = sorted(mn.items(), key=lambda kv: -kv[1])[:14]
                out.append("        " + ", ".join(f"{m} x{n}" for m, n in top))
                if dump:
                    out.extend("        | " + lines[i] for i in range(first, last + 1))
    text = "\n".join(out) + "\n"
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        open(out_path, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
