"""cs_export_results_v1 (the result text files of a run; reference src/app/SL_CoSLAM.cpp:1914-2028) against the files the
reference's own CoSLAM::exportResults wrote (tests/golden/export_golden.npz, made by oracle/_ref/ref_export_test with
SL_CoSLAM.cpp compiled in place).  Host code: runs without a GPU -- the library only has to load."""
import os
import subprocess

import numpy as np
import pytest

import coslam_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FILES = ("input_videos.txt", "mappts.txt", "0_campose.txt", "1_campose.txt", "0_featpts.txt", "1_featpts.txt")


def _cams(g):
    cams = []
    for c in range(int(g["nCams"])):
        W, H, start = g[f"c{c}_whs"]
        cams.append(dict(videoFilePath=bytes(g[f"c{c}_path"]), K=g[f"c{c}_K"], kc=g[f"c{c}_kc"], W=W, H=H, startFrameInVideo=start,
                         poseFrame=g[f"c{c}_poseFrame"], poseR=g[f"c{c}_poseR"], poseT=g[f"c{c}_poseT"], featPtr=g[f"c{c}_featPtr"],
                         featPointId=g[f"c{c}_featId"], featXY=g[f"c{c}_featXY"]))
    return cams


def test_result_files_equal_the_references_byte_for_byte(tmp_path):
    g = np.load(os.path.join(GOLD, "export_golden.npz"))
    out = tmp_path / "run"
    coslam_amd.export_results_v1(out, _cams(g), int(g["curFrame"]), g["ptId"], g["ptM"], g["ptCov"], cov_as_reference=True)
    for n in FILES:
        assert (out / n).read_bytes() == bytes(g["file_" + n]), n
    # the cameras start at different frames, one frame has no static feature point at all, a path holds a blank
    assert b"cam 0.avi" in bytes(g["file_input_videos.txt"]) and len(g["c0_poseFrame"]) != len(g["c1_poseFrame"])
    assert (np.diff(g["c0_featPtr"]) == 0).any()


def test_result_files_covariance_line_and_errors(tmp_path):
    """the reference's mappts.txt repeats ONE line of 9 numbers for every point (its inner loop re-declares `i`,
    SL_CoSLAM.cpp:1965-1966: entry k of the k-th point); cov_as_reference=False writes each point's own covariance instead."""
    g = np.load(os.path.join(GOLD, "export_golden.npz"))
    lines = bytes(g["file_mappts.txt"]).decode().splitlines()
    n = int(lines[0])
    assert n == len(g["ptId"]) >= 9
    cov_lines = lines[3::3]
    assert len(cov_lines) == n and len(set(cov_lines)) == 1
    assert [float(v) for v in cov_lines[0].split()] == [float("%g" % g["ptCov"][k, k]) for k in range(9)]
    out = tmp_path / "own"
    coslam_amd.export_results_v1(out, _cams(g), int(g["curFrame"]), g["ptId"], g["ptM"], g["ptCov"], cov_as_reference=False)
    own = (out / "mappts.txt").read_text().splitlines()
    assert own[1::3] == lines[1::3] and own[2::3] == lines[2::3]
    for p in range(n):
        assert [float(v) for v in own[3 + 3 * p].split()] == [float("%g" % v) for v in g["ptCov"][p]]
    assert (out / "0_campose.txt").read_bytes() == bytes(g["file_0_campose.txt"])
    # fewer than 9 points: the reference would read past its list; the writer puts 0 there instead of failing
    coslam_amd.export_results_v1(tmp_path / "few", _cams(g), int(g["curFrame"]), g["ptId"][:4], g["ptM"][:4], g["ptCov"][:4])
    few = (tmp_path / "few" / "mappts.txt").read_text().splitlines()
    assert few[0] == "4" and few[3].split()[4:] == ["0"] * 5
    # a directory that cannot be created is an error with a message, not a silent no-op
    with pytest.raises(RuntimeError, match="cannot create"):
        coslam_amd.export_results_v1(tmp_path / "no" / "such" / "parent", _cams(g), int(g["curFrame"]), g["ptId"], g["ptM"], g["ptCov"])


def test_reference_export_code_agrees_with_the_writer(tmp_path):
    """oracle/_ref/ref_export_test: a CoSLAM object filled through the reference's own containers, exportResults run from
    SL_CoSLAM.cpp compiled in place, the six files compared with cs_export_results_v1's in the same process.  The binary is
    built where the reference tree exists and travels with the snapshot."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_export_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_export_test not built (needs /root/reference at build time)")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ref_export_test: OK" in out.stdout, out.stdout + out.stderr
