"""The C++ side of the drop-in boundary, on the MI355X.

* tests/cxx/shim_link_test.cpp         the three header-compatible shims (include/shim/) with stand-in caller types;
* oracle/_ref/ref_gpuklt_dropin_test   the REFERENCE'S OWN src/tracking/GPUKLT.cpp (+ SL_Track2D.cpp and the data model)
                                       compiled in place over the shims, driven as SingleSLAM drives it, and checked slot
                                       by slot against the on-device hand-back (cs_klt_handback_dev);
* oracle/_ref/ref_ba_dropin_test       the REFERENCE'S OWN src/app/SL_CoSLAMRobustBA.cpp (parseInputs / run),
                                       SL_InterCamPoseEstimator.cpp (addMapPoints / apply) and SL_SingleSLAM.cpp
                                       (chooseStaticFeatPts ...) compiled in place over the shims.
* oracle/_ref/ref_register_test        the REFERENCE'S OWN searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164) over
                                       FeaturePoints lists built with the reference's classes, against cs_register_search.
* oracle/_ref/ref_ncc_test             the REFERENCE'S OWN NCCBlock::computeScaled, matchNCCBlock (src/slam/SL_NCCBlock.cpp) and
                                       getEpiNccMat (src/slam/SL_FeatureMatching.cpp) against cs_ncc_match_between.
* oracle/_ref/ref_posegraph_test       the REFERENCE'S OWN GlobalPoseGraph::computeNewCameraRotations / computeNewCameraTranslations
                                       (src/slam/SL_GlobalPoseEstimation.cpp) against relaxPoseGraphs / cs_posegraph_* on graphs
                                       built with the reference's classes; ref_posegraph_methods_test: the two member functions
                                       taken from include/shim/slam/coslam_posegraph.h instead, checked against the golden file
                                       the first binary writes on the CPU.
* oracle/_ref/ref_export_test          the REFERENCE'S OWN CoSLAM::exportResultsVer1 (src/app/SL_CoSLAM.cpp) against
                                       cs_export_results_v1: six text files, byte for byte.
The oracle/_ref binaries are built by oracle/Makefile where the reference tree exists (__graft_entry__.build()) and
travel with the repo snapshot; the reference sources themselves are never copied."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, ok_text):
    assert os.path.exists(exe), f"{exe} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` where /root/reference exists"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ok_text in out.stdout, out.stdout + out.stderr
    return out.stdout


def test_reference_gpuklt_source_runs_over_the_shim_and_matches_the_device_handback(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_gpuklt_dropin_test"), "ref GPUKLT drop-in ok")
    print(out)


def test_reference_ba_callers_run_over_the_shim(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_ba_dropin_test"), "ref BA callers drop-in ok")
    assert "RobustBundleRTS drop-in ok" in out and "InterCamPoseEstimator drop-in ok" in out
    print(out)


def test_reference_search_function_agrees_with_the_registration_kernel(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_register_test"), "candidates agree with the reference's searchMahaNearestFeatPt")
    print(out)


def test_reference_ncc_code_agrees_with_the_ncc_kernels(hip):
    out = _run(os.path.join(ROOT, "oracle", "_ref", "ref_ncc_test"), "equal the reference's bit for bit")
    print(out)


def test_reference_posegraph_code_agrees_with_the_relaxation_kernel(hip, tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_posegraph_test")
    out = _run(exe, "vs the reference's own methods")
    print(out)
    gold = str(tmp_path / "pg.bin")
    subprocess.run([exe, "golden", gold], check=True, timeout=600)        # CPU: the reference's methods
    m = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_posegraph_methods_test"), gold], capture_output=True, text=True,
                       timeout=600)
    assert m.returncode == 0 and "ref_posegraph_methods_test: OK" in m.stdout, m.stdout + m.stderr
    print(m.stdout)


def test_reference_export_code_agrees_with_the_result_writer(hip, tmp_path):
    """the reference's own CoSLAM::exportResults (src/app/SL_CoSLAM.cpp compiled in place) against cs_export_results_v1; host
    code, also run by the CPU suite (tests/test_results_cpu.py)"""
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_export_test"), str(tmp_path)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and "ref_export_test: OK" in out.stdout, out.stdout + out.stderr


def test_cxx_shims_link_and_run(hip):
    """include/shim/: V3D_GPU::KLT_SequenceTracker, intraCamEstimate, bundleAdjustRobust with stand-in caller types."""
    src = os.path.join(ROOT, "tests", "cxx", "shim_link_test.cpp")
    exe = os.path.join(ROOT, "tests", "cxx", "shim_link_test.bin")
    libdir = os.path.join(ROOT, "coslam_amd", "lib")
    cmd = ["g++", "-std=c++11", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "shim"), src,
           "-L", libdir, "-lcoslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout
