#!/usr/bin/env python3
"""profiles/r06_cfg5_klt_* from what tools/r06_cfg5.sh left under gpurun_out/r06/cfg5/ (kernel trace + the FETCH / WRITE / SQ counter passes of
the cfg5 KLT stage: 4 cameras 1920 x 1080 x 5000 slots as one camera group).  Run from the repository root after the gpurun call."""
import json
import re
import shutil

SRC = "gpurun_out/r06/cfg5"
K = "k_track_rows_fused<8, 7, false>"


def val(path, counter):
    for ln in open(path):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 5 and c[1].startswith(K) and c[2] == counter:
            return float(c[4]), int(c[3])
    raise KeyError((path, counter))


def main():
    for n in ("kernel_stats", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ"):
        shutil.copy(f"{SRC}/cfg5_klt_{n}.md", f"profiles/r06_cfg5_klt_{n}.md")
    f, n = val(f"{SRC}/cfg5_klt_pmc_FETCH_SIZE.md", "FETCH_SIZE")
    w, _ = val(f"{SRC}/cfg5_klt_pmc_WRITE_SIZE.md", "WRITE_SIZE")
    v, _ = val(f"{SRC}/cfg5_klt_pmc_SQ.md", "SQ_INSTS_VALU")
    avg = None
    for ln in open(f"{SRC}/cfg5_klt_kernel_stats.md"):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 5 and c[1].startswith(K):
            avg, calls = float(c[4]), int(c[2])
    cams_per_launch = 2
    alg = cams_per_launch * 5000 * 3096          # 4 levels x 768 B + 24 B of feature I/O per feature
    traffic = int(round(f * 1024 * 2 + w * 1024))   # gfx950 tallies 128-byte fetches at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    floor = v * 4 / (1024 * 2.4e9) * 1e6
    old = json.load(open("profiles/r05_cfg5_klt_pmc.json"))
    j = {"kernel": K,
         "workload": "BASELINE.json configs[4]'s KLT stage: 4 cameras x 5000 features (100 x 50), 1920x1080, 4 levels, 10 iterations, 7x7 window; 2 launches per "
                     "frame of 2 cameras each, a camera on four XCDs (round 6: the span is cut evenly; GROUP_CAM_CFG5=1 PMC_CAMS=4 tools/pmc_klt.py, tools/r06_cfg5.sh)",
         "dispatches": n, "avg_launch_us": avg, "launches_per_frame": 2, "cameras_per_launch": cams_per_launch,
         "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "fetch_correction": 2.0, "traffic_bytes_per_launch": traffic,
         "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg, "achieved_GBps": alg / (avg * 1e-6) / 1e9,
         "hbm_frac": alg / (avg * 1e-6) / 8e12, "valu_wave_insts_per_launch": v, "valu_issue_floor_us": floor, "valu_frac": floor / avg,
         "round_5": {k: old[k] for k in ("avg_launch_us", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic", "hbm_frac", "valu_frac")},
         "sources": [f"profiles/r06_cfg5_klt_{n_}.md" for n_ in ("kernel_stats", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ")]}
    json.dump(j, open("profiles/r06_cfg5_klt_pmc.json", "w"), indent=1)
    print(json.dumps({k: j[k] for k in ("avg_launch_us", "traffic_over_algorithmic", "hbm_frac", "valu_frac", "round_5")}))


if __name__ == "__main__":
    main()
