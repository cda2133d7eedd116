"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/coslam_hip.h declares, refuses to run without a device (no CPU fallback), and the C++ shim header
compiles against it."""
import ctypes as C
import os
import re
import subprocess

import pytest

import coslam_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "coslam_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = coslam_amd.lib()
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.cs_version() >= 100


def test_struct_layouts_match_the_reference_types():
    from coslam_amd.klt import KLT_SequenceTrackerConfig, KLT_TrackedFeature
    from coslam_amd.pose import IntraCamPoseOption

    assert KLT_TrackedFeature.itemsize == 20  # {int status; float pos[2]; float gain; int fed;}
    assert C.sizeof(KLT_SequenceTrackerConfig) == 44
    assert C.sizeof(IntraCamPoseOption) == 96
    cfg = KLT_SequenceTrackerConfig()  # defaults of v3d_gpuklt.h:181-191
    assert (cfg.nIterations, cfg.nLevels, cfg.levelSkip, cfg.windowWidth) == (12, 3, 2, 5)
    assert cfg.minDistance == 8 and cfg.trackWithGain == 0 and abs(cfg.SSD_Threshold - 5000.0) < 1e-6
    lib = coslam_amd.lib()
    d = KLT_SequenceTrackerConfig()
    d.nIterations = 0
    lib.cs_klt_config_default(C.byref(d))
    assert d.as_dict() == cfg.as_dict()


def test_no_device_means_loud_failure_not_cpu_fallback():
    lib = coslam_amd.lib()
    if lib.cs_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.KLT_SequenceTracker(coslam_amd.KLT_SequenceTrackerConfig(), device=0)
    import numpy as np

    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.intraCamEstimate(np.eye(3), np.eye(3), np.zeros(3), 4, None, np.ones((4, 3)), np.ones((4, 2)), 10.0)
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.bundleAdjustRobust(0, np.eye(3)[None], np.eye(3)[None].copy(), np.zeros((1, 3)), 0, np.ones((1, 3)),
                                      [[(0, 1.0, 1.0)]], 6.0, 1, 1)


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under coslam_amd/ may import, include, link or load it."""
    pkg = os.path.join(ROOT, "coslam_amd")
    banned = ("import oracle", "from oracle", "liboracle", "libintracam_ref", "oracle.h\"", "_oracle.c")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in banned:
                    assert pat not in src, (f, pat)


def test_cxx_shim_compiles_and_links():
    """include/v3d_gpuklt.h + SL_IntraCamPose.h + SL_BundleAdjust.h: the reference's C++ signatures over the C-ABI."""
    src = os.path.join(ROOT, "tests", "cxx", "shim_link_test.cpp")
    exe = os.path.join(ROOT, "tests", "cxx", "shim_link_test.bin")
    libdir = os.path.join(ROOT, "coslam_amd", "lib")
    cmd = ["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "shim"), src,
           "-L", libdir, "-lcoslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout


def test_cxx_dropin_bench_compiles():
    """tools/cxx/dropin_bench.cpp (latency of the reference-shaped synchronous calls) builds against the shims."""
    src = os.path.join(ROOT, "tools", "cxx", "dropin_bench.cpp")
    exe = os.path.join(ROOT, "tools", "cxx", "dropin_bench.bin")
    libdir = os.path.join(ROOT, "coslam_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "shim"),
           src, "-L", libdir, "-lcoslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-o", exe]
    subprocess.check_call(cmd)
    assert os.path.exists(exe)
