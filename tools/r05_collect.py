#!/usr/bin/env python3
"""Copies what tools/r05_profiles.sh left under gpurun_out/r05/profiles/ into profiles/r05_* (the files' comment headers are kept, the bodies
replaced), refreshes profiles/r05_tracker_pmc.json (what bench.py's roofline.traffic reads) from the KLT stage's counter passes, and -- when
gpurun_out/r05/bench_default.json / bench_driver.json exist -- the bench lines.  Run from the repository root after a gpurun call."""
import json
import os
import re

SRC = "gpurun_out/r05/profiles"


def header_lines(path):
    out = []
    if not os.path.exists(path):
        path = path.replace("r05_", "r04_")   # (first collection of the round: the file's comment header comes from the previous round's)
    for ln in open(path):
        if ln.startswith("#") and not ln.startswith("# region") and not ln.startswith("##"):
            out.append(ln)
        else:
            break
    return out


def val(path, kernel, counter):
    for ln in open(path):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 5 and c[1].startswith(kernel) and c[2] == counter:
            return float(c[4]), int(c[3])
    raise KeyError((path, kernel, counter))


def main():
    line = open(f"{SRC}/headline_traced_bench_line.json").read().strip().splitlines()[-1]
    fps = json.loads(line)["value"]
    open("profiles/r05_headline_traced_bench_line.json", "w").write(line + "\n")
    h = [re.sub(r"\(\d+ frames/s under the", f"({fps:.0f} frames/s under the", ln) for ln in header_lines("profiles/r05_headline_bench_kernel_stats.md")]
    open("profiles/r05_headline_bench_kernel_stats.md", "w").write("".join(h) + "\n" + open(f"{SRC}/headline_bench_kernel_stats.md").read())
    for name in ("klt_pmc_FETCH_SIZE", "klt_pmc_WRITE_SIZE", "klt_pmc_SQ", "pose_stream_pmc_FETCH_SIZE", "pose_stream_pmc_WRITE_SIZE", "pose_stream_pmc_SQ"):
        h = header_lines(f"profiles/r05_{name}.md")
        open(f"profiles/r05_{name}.md", "w").write("".join(h) + "\n" + open(f"{SRC}/{name}.md").read())
    K = "k_track_rows_fused<8, 7, false>"
    f, n = val("profiles/r05_klt_pmc_FETCH_SIZE.md", K, "FETCH_SIZE")
    w, _ = val("profiles/r05_klt_pmc_WRITE_SIZE.md", K, "WRITE_SIZE")
    v, _ = val("profiles/r05_klt_pmc_SQ.md", K, "SQ_INSTS_VALU")
    j = json.load(open("profiles/r05_tracker_pmc.json" if os.path.exists("profiles/r05_tracker_pmc.json") else "profiles/r04_tracker_pmc.json"))
    j["sources"] = [x.replace("r04_", "r05_") if "klt_pmc" in x else x for x in j["sources"]]
    j["command"] = j["command"].replace("r04", "r05")
    j["note"] = j["note"].replace("unchanged since round 3 and so are its counters", "unchanged since round 3 and so are its counters (measured again in round 5)")
    j["FETCH_SIZE_KB_per_launch"], j["WRITE_SIZE_KB_per_launch"], j["dispatches"] = f, w, n
    j["traffic_bytes_per_launch"] = int(round(f * 1024 * 2 + w * 1024))
    j["valu_wave_insts_per_launch"] = v
    j["valu_note"] = re.sub(r"SQ_INSTS_VALU [\d.]+ M", f"SQ_INSTS_VALU {v / 1e6:.2f} M", j["valu_note"])
    j["valu_note"] = re.sub(r"a floor of [\d.]+ us", f"a floor of {v * 4 / (1024 * 2.4e9) * 1e6:.1f} us", j["valu_note"])
    json.dump(j, open("profiles/r05_tracker_pmc.json", "w"), indent=1)
    print(f"profiles/r05_*: traced {fps:.0f} frames/s; tracker FETCH {f:.1f} KB WRITE {w:.1f} KB VALU {v / 1e6:.2f} M per launch")
    if os.path.exists("gpurun_out/r05/bench_default.json") and os.path.exists("gpurun_out/r05/bench_driver.json"):
        load = lambda p: json.loads(open(p).read().strip().splitlines()[-1])   # noqa: E731
        d, dv = load("gpurun_out/r05/bench_default.json"), load("gpurun_out/r05/bench_driver.json")
        open("profiles/r05_bench_line.json", "w").write(json.dumps(d) + "\n")
        open("profiles/r05_bench_line_driver_cmd.json", "w").write(json.dumps(dv) + "\n")

        def row(cmd, jj):
            c, r = jj["config"], jj["roofline"]
            return (f"{cmd:<40s} {jj['value']:.1f} frames/s  {jj['ms_per_step']:.4f} ms/frame   C++ loop {(c.get('cxx_frame_loop') or {}).get('frames_per_s')}   "
                    f"with upload x{(c.get('with_upload') or {}).get('ratio_to_value'):.3f}   joint BA worker busy "
                    f"{c['key_frame_solves_duty']['joint_ba']['share_of_timed_region']:.2f}   roofline hbm {r['frac']:.4f} valu {r['valu']['frac']:.3f}")

        c, cb = d["config"], d["cpu_baseline"]
        suite = ""
        if os.path.exists("gpurun_out/r05/gpu_suite.log"):
            m = re.findall(r"\d+ passed[^\n]*", open("gpurun_out/r05/gpu_suite.log").read())
            suite = m[-1] if m else ""
        seq = {k: c["secondary_sequential_registration"][k] for k in ("frames_per_s", "ms_per_step", "steps", "ratio_to_value", "loops_whose_sweeps_did_not_settle")}
        txt = f"""# bench lines of round 4, one MI355X box (gpurun), the round's closing code (same call: GPU suite {suite}; smoke 6 legs; then:)
{row('python bench.py', d)}
{row('python bench.py --steps 20 --warmup 5', dv)}
cpu_baseline (oracle, kind port): {cb['value']:.2f} frames/s on {cb['cores']} threads, {cb['value_1_thread']:.2f} on 1 ({cb['sample']})
secondary_reference_ba_request_policy: {json.dumps(c['secondary_reference_ba_request_policy'])[:330]}
secondary_sequential_registration: {json.dumps(seq)}
secondary_reference_default_klt: {c['secondary_reference_default_klt']['frames_per_s']:.0f} frames/s (KLT stage, 6 levels / skip 2 / 12 iterations / 5x5)
register_decision: {json.dumps({k: v_ for k, v_ in c['register_decision'].items() if k != 'what' and k != 'merge'})}
register_decision.merge: {json.dumps({k: v_ for k, v_ in (c['register_decision'].get('merge') or {}).items() if k != 'what'})}
(earlier states of this round: 2153.2 / 2020.0 / C++ 2280.4 before the search was sized to fit beside the tracker; 2192.5 / 2036.4 / C++ 2345.6 before the
 bMerge frames joined the loop -- another box)
(box-to-box spread on the pool is ~8-9 %: the A/B tables in r05_ab_runs.txt are each from one box; the same commit measured 1978-2190 across boxes)
"""
        open("profiles/r05_bench_lines.txt", "w").write(txt)
        print(txt.splitlines()[1])
        print(txt.splitlines()[2])


if __name__ == "__main__":
    main()
