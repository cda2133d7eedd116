#!/usr/bin/env python
"""r05_drift_summary.py <dir with *.jsonl of tools/r05_drift.py> -> markdown table (profiles/r05_drift_summary.md)

One block per sampled frame (500 / 1000 / 1500), one row per variant: bench.py's raw pose error next to the errors after a
similarity / rigid alignment of the 8 camera centres, the gauge motion the alignment removed, the map error over the points in
use (raw / after their own alignment), the joint BA's converged cost per measurement and the bookkeeping that explains them."""
import glob
import json
import os
import sys


def load(f):
    out = []
    for line in open(f):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        if "frame" in r:
            out.append(r)
    return out


def main():
    d = sys.argv[1]
    frames = [int(x) for x in sys.argv[2:]] or [500, 1000, 1500]
    files = sorted(glob.glob(os.path.join(d, "*.jsonl")))
    runs = {os.path.basename(f)[:-6]: load(f) for f in files}
    print("| frame | variant | raw max \\|t - t_true\\| | centres raw | centres after Sim(3) | after rigid | gauge: scale / rot deg / trans m | "
          "map points in use: median err raw | after own Sim(3) (p90) | joint BA cost / meas | used points initial / new | false | LM steps / cam |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for fr in frames:
        for name, R in runs.items():
            r = next((x for x in R if x["frame"] == fr), None)
            if r is None:
                continue
            g = r["gauge_sim"]
            ba = r.get("joint_ba", {})
            mr, ms = r.get("map_err_raw", {}), r.get("map_err_self_aligned", {})
            lm = sum(x[1] for x in r["pose_rounds_lm"]) / max(len(r["pose_rounds_lm"]), 1)
            print(f"| {fr} | {name} | {r['t_err_max']:.3f} | {r['centre_err_raw_max']:.3f} | {r['centre_err_sim_max']:.4f} | {r['centre_err_rigid_max']:.4f} | "
                  f"{g['scale']:.3f} / {g['rot_deg']:.2f} / {g['trans']:.3f} | {mr.get('median', float('nan')):.3f} | "
                  f"{ms.get('median', float('nan')):.3f} ({ms.get('p90', float('nan')):.2f}) | {ba.get('cost_per_meas', float('nan')):.2f} | "
                  f"{r['used_points_initial']} / {r['used_points_new']} | {r['map_points_false']} | {lm:.0f} |")
    # the worst aligned error over each whole run
    print("\n| variant | max over the run of: centres after Sim(3) | after rigid | raw |\n|---|---|---|---|")
    for name, R in runs.items():
        if R:
            print(f"| {name} | {max(x['centre_err_sim_max'] for x in R):.4f} | {max(x['centre_err_rigid_max'] for x in R):.4f} | "
                  f"{max(x['centre_err_raw_max'] for x in R):.3f} |")


if __name__ == "__main__":
    main()
