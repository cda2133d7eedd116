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
    "register.hip",
    "ncc.hip",
    "posegraph.hip",
    "results.cpp",
    "ba.hip",
    "comm.hip",
]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    # the pyramid / detector kernels are bit-exact against the oracle: no FMA contraction
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-value",
    "-Wno-unused-result",
    "-ldl",
]


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
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "..", "include", f) for f in os.listdir(os.path.join(ROOT, "..", "include"))]
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [_hipcc()] + HIPCC_FLAGS + srcs + ["-o", LIB]
    if verbose:
        print("[coslam_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    print(LIB)
