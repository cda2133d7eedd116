#!/usr/bin/env python3
"""Copies what tools/r06_profiles.sh left under gpurun_out/r06/ into profiles/r06_* (run from the repository root after the gpurun call):
the two traced loops' kernel tables and frame listings, the tracker's counter passes (+ profiles/r06_tracker_pmc.json, what bench.py's
roofline.traffic falls back to), the bench lines, the GPU suite's tail, the A/B records of the round."""
import json
import os
import re

S = "gpurun_out/r06"


def load(p):
    return json.loads(open(p).read().strip().splitlines()[-1])


def put(dst, head, body):
    open(dst, "w").write(head.rstrip("\n") + "\n\n" + body)
    print("wrote", dst)


def val(path, kernel, counter):
    for ln in open(path):
        c = [x.strip() for x in ln.split("|")]
        if len(c) > 5 and c[1].startswith(kernel) and c[2] == counter:
            return float(c[4]), int(c[3])
    raise KeyError((path, kernel, counter))


def main():
    t = load(f"{S}/trace/headline_traced_bench_line.json")
    put("profiles/r06_headline_bench_kernel_stats.md",
        "# rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-secondary --no-cxx-loop --no-upload-leg` (tools/r06_trace.sh): the\n"
        "# PYTHON frame loop's timed region (the launch-per-step registration sequence; %.0f frames/s under the profiler), summarised by\n"
        "# tools/rocpd_summary.py.  The bench's `value` is the C++ loop's: profiles/r06_cxx_frame_loop_kernel_stats.md." % t["value"],
        open(f"{S}/trace/headline_bench_kernel_stats.md").read())
    open("profiles/r06_headline_traced_bench_line.json", "w").write(json.dumps(t) + "\n")
    put("profiles/r06_headline_frames.txt", "# tools/r06_frames.py over the same trace: every kernel of six consecutive frames, per stream (s = stream id)",
        open(f"{S}/trace/frames.txt").read())
    c = load(f"{S}/cxx_trace/untraced_line.json")
    ct = load(f"{S}/cxx_trace/traced_line.json")
    put("profiles/r06_cxx_frame_loop_kernel_stats.md",
        "# rocprofv3 --kernel-trace of tools/cxx/frame_loop.bin <workload> 300 30 0 2 (tools/r06_cxx_trace.sh): the C++ frame loop, the fused\n"
        "# registration launches (cs_register_decide_kinds_rounds_dev, cs_feat_ref_advance_refine_dev); %.0f frames/s under the profiler, %.0f without\n"
        "# on the same box (this stand-alone run starts at frame 0 of the sequence: its first bMerge walks are the long ones; the bench runs the\n"
        "# same stretch as the Python loop's timed region)." % (ct["frames_per_s"], c["frames_per_s"]),
        open(f"{S}/cxx_trace/cxx_loop_kernel_stats.md").read())
    put("profiles/r06_cxx_frame_loop_frames.txt", "# tools/r06_frames.py over the same trace: every kernel of eight consecutive frames, per stream",
        open(f"{S}/cxx_trace/frames.txt").read())
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        put(f"profiles/r06_klt_pmc_{name}.md",
            f"# rocprofv3 --pmc {name if name != 'SQ' else 'SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU'} --kernel-trace -- "
            "python tools/pmc_klt.py (the camera group's KLT stage alone, its own pass: tools/r06_profiles.sh), per kernel by tools/rocpd_summary.py",
            open(f"{S}/pmc/klt_pmc_{name}.md").read())
    K = "k_track_rows_fused<8, 7, false>"
    f, n = val("profiles/r06_klt_pmc_FETCH_SIZE.md", K, "FETCH_SIZE")
    w, _ = val("profiles/r06_klt_pmc_WRITE_SIZE.md", K, "WRITE_SIZE")
    v, _ = val("profiles/r06_klt_pmc_SQ.md", K, "SQ_INSTS_VALU")
    j = json.load(open("profiles/r05_tracker_pmc.json"))
    j["sources"] = [x.replace("r05_", "r06_") if "klt_pmc" in x else x for x in j["sources"]]
    j["command"] = j["command"].replace("r05", "r06")
    j["note"] = re.sub(r"\(measured again in round 5\)", "(measured again in rounds 5 and 6)", j["note"])
    j["FETCH_SIZE_KB_per_launch"], j["WRITE_SIZE_KB_per_launch"], j["dispatches"] = f, w, n
    j["traffic_bytes_per_launch"] = int(round(f * 1024 * 2 + w * 1024))
    j["valu_wave_insts_per_launch"] = v
    j["valu_note"] = re.sub(r"SQ_INSTS_VALU [\d.]+ M", f"SQ_INSTS_VALU {v / 1e6:.2f} M", j["valu_note"])
    j["valu_note"] = re.sub(r"a floor of [\d.]+ us", f"a floor of {v * 4 / (1024 * 2.4e9) * 1e6:.1f} us", j["valu_note"])
    json.dump(j, open("profiles/r06_tracker_pmc.json", "w"), indent=1)
    print(f"tracker FETCH {f:.1f} KB WRITE {w:.1f} KB VALU {v / 1e6:.2f} M per launch")
    lines = []
    for n_, dst in (("bench_default", "r06_bench_line.json"), ("bench_driver", "r06_bench_line_driver_cmd.json"), ("bench_driver2", None)):
        d = load(f"{S}/{n_}.json")
        if dst:
            open("profiles/" + dst, "w").write(json.dumps(d) + "\n")
        cx = d["config"].get("cxx_frame_loop") or {}
        py = d["config"].get("python_frame_loop") or {}
        lines.append("%-14s steps %3d warmup %2d: value %.1f frames/s (%s; python loop %s), roofline frac %.3f (%s us per launch in the loop), cpu_baseline %.2f frames/s on %s core(s)" %
                     (n_, d["steps"], d["warmup"], d["value"], d["config"].get("value_source"), py.get("frames_per_s"), d["roofline"]["frac"],
                      d["roofline"].get("us_per_launch_in_loop", d["roofline"].get("us_per_launch")), (d.get("cpu_baseline") or {}).get("value", float("nan")),
                      (d.get("cpu_baseline") or {}).get("cores")))
        lines.append("               second visits: %s" % json.dumps((d["config"].get("register_decision") or {}).get("second_visits"))[:400])
        lines.append("               cxx loop: %s" % json.dumps({k: cx.get(k) for k in ("frames_per_s", "bmerge_frames", "second_visit_features_attached", "second_visit_conflicts_in_timed_region")}))
    put("profiles/r06_bench_lines.txt", "# python bench.py (default) and the driver's command (--steps 20 --warmup 5, twice), the closing run of round 6 (tools/r06_profiles.sh)", "\n".join(lines) + "\n")
    put("profiles/r06_gpu_suite.txt", "# python -m pytest tests -q -m gpu and __graft_entry__.smoke() on the GPU box, the closing run (tools/r06_profiles.sh)",
        "".join(x for x in open(f"{S}/gpu_suite.log") if " passed" in x or " failed" in x or " error" in x) +
        "".join(x for x in open(f"{S}/smoke.log") if x.startswith("[smoke]")))
    ab = []
    for title, path in (("registration's fused launches on (1) / off (0), same box (tools/r06_fused.sh; includes the scaled forward differences of k_intracam in both)", f"{S}/fused/ab.txt"),
                        ("classification over feature references on / off, same box (tools/r06_classify_refs.sh)", f"{S}/classify_refs/ab.txt"),
                        ("the closing build, two repeats (tools/r06_quick.sh)", f"{S}/quick/ab.txt")):
        if os.path.exists(path):
            ab.append("## " + title + "\n" + open(path).read())
    mp = f"{S}/merge_print/out.txt"
    if os.path.exists(mp):
        ls = [x[:330] for x in open(mp) if x.startswith("k_decide_merge")]
        rv = [x.strip() for x in open(mp) if x.startswith("k_revisit_decide")]
        ab.append("## k_decide_merge's own account, the seven bMerge walks of a 330-frame run from frame 0 (cs_debug_set(\"merge_print\", 1); tools/r06_merge_print.sh)\n" + "".join(ls))
        ab.append("## k_revisit_decide's phases, every 20th launch of the same run\n" + "\n".join(rv[::20]) + "\n")
    put("profiles/r06_ab_runs.txt", "# round 6's same-box A/B records (bench.py short legs: the C++ frame loop's `value`)", "\n".join(ab))


if __name__ == "__main__":
    main()
