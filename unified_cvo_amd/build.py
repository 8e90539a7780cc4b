"""In-tree build of the gfx950 backend (libcvo_hip.so) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container;
the resulting shared object travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcvo_hip.so")
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",  # only the explicit fmaf() calls fuse (DESIGN.md "Numerics")
    # kernel arguments arrive in SGPRs with the wave instead of behind a first scalar-load round trip (the
    # per-iteration kernels are latency bound; the compiler keeps a fallback prologue for older firmware)
    "-mllvm", "-amdgpu-kernarg-preload-count=12",
    "-Wall",
    "-Wextra",
    "-Wno-unused-parameter",
    "-Wno-pass-failed",
]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the gfx950 backend cannot be built")


def sources():
    """The translation units hipcc is given: ONE (cvo_hip.hip includes its sections - cvo_ctx.hip, cvo_sched.hip ... - see
    the note at its top)."""
    return [os.path.join(CSRC, f) for f in ("cvo_hip.hip",)]


def headers():
    """Everything else a build depends on: the kernel headers, cvo_internal.h, the sections of cvo_hip.hip, the C-ABI."""
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                if f.endswith(".h") or (f.endswith(".hip") and f != "cvo_hip.hip"))
    hs.append(os.path.join(ROOT, "include", "cvo_hip.h"))
    hs.append(os.path.join(ROOT, "include", "cvo_hip_debug.h"))
    return hs


def source_hash():
    """sha256 (16 hex digits) over the kernel sources: keys measurements that belong to one version of the kernels
    (profiles/kernel_traffic.json carries it; bench.py reports PMC traffic only while it matches)."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(sources() + [q for q in headers() if q.startswith(CSRC)]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def _compile(lib, defines, verbose):
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("CVO_EXTRA_HIPCC_FLAGS", "").split()  # (experiments: -DCVO_ASSOC_WAVES=6 ...)
    cmd = [_hipcc()] + HIPCC_FLAGS + defines + extra + ["-I", os.path.join(ROOT, "include"), "-o", lib] + sources()
    if verbose:
        print("[unified_cvo_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into unified_cvo_amd/lib/libcvo_hip.so."""
    if not force and not needs_build():
        return LIB
    return _compile(LIB, [], verbose)


def build_variant(name, defines, verbose=True):
    """An experiment build of the same sources (lib/libcvo_hip_<name>.so) with extra -D switches: scripts/exp_time.py
    times it next to the product library and checks that the poses stay bit-identical."""
    return _compile(os.path.join(LIBDIR, f"libcvo_hip_{name}.so"), list(defines), verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
