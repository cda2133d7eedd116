"""Build recipe for libcoslam_hip.so (hand-written HIP for gfx950, no torch extension machinery).

The library is built IN-TREE (coslam_amd/lib/libcoslam_hip.so) so that it travels with the repo
snapshot to the GPU box.  hipcc cross-compiles gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libcoslam_hip.so")

HIP_SOURCES = [
    "klt_pyramid.hip",
    "klt_track.hip",
    "klt_track_rows.hip",
    "klt_detect.hip",
    "klt_seq.hip",
    "pose.hip",
    "handback.hip",
    "hostview.hip",
    "register.hip",
    "keyframe.hip",
    "poseupdate.hip",
    "ncc.hip",
    "newpts.hip",
    "posegraph.hip",
    "results.cpp",
    "ba.hip",
    "comm.hip",
]

COMPILE_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # the pyramid / detector kernels are bit-exact against the oracle: no FMA contraction
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-value",
    "-Wno-unused-result",
]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
HIPCC_FLAGS = COMPILE_FLAGS + ["-shared", "-ldl"]   # one-command form (build_variant)
OBJDIR = os.path.join(LIBDIR, "obj")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP path cannot be built (there is no CPU fallback)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_variant(name, defines, verbose=True):
    """an A/B build with extra -D flags into coslam_amd/lib/libcoslam_hip_<name>.so (select with COSLAM_HIP_LIB)"""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    out = os.path.join(LIBDIR, f"libcoslam_hip_{name}.so")
    cmd = [_hipcc()] + HIPCC_FLAGS + [f"-D{d}" for d in defines] + srcs + ["-o", out]
    if verbose:
        print("[coslam_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_hip(force=False, verbose=True):
    """every source to its own object (in parallel, only the stale ones: a header change rebuilds all), then one link"""
    from concurrent.futures import ThreadPoolExecutor

    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(ROOT, "..", "include", f) for f in os.listdir(os.path.join(ROOT, "..", "include"))
             if os.path.isfile(os.path.join(ROOT, "..", "include", f))]
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and not _stale(LIB, srcs + hdrs):
        return LIB
    hipcc = _hipcc()
    jobs = []
    for src in srcs:
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            # .cpp host sources go through hipcc as well (same flags, HIP headers available)
            jobs.append(([hipcc] + COMPILE_FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj], src))

    def run(job):
        cmd, src = job
        if verbose:
            print("[coslam_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJDIR, os.path.basename(src) + ".o") for src in srcs]
    cmd = [hipcc] + LINK_FLAGS + objs + ["-o", LIB]
    if verbose:
        print("[coslam_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    print(LIB)
