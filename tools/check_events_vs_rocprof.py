"""Do the per-kernel figures of a bench.py line follow from a rocprofv3 kernel table?

    python tools/check_events_vs_rocprof.py LINE.json KERNEL_STATS.csv [--tol 0.03] [--steps-from KERNEL]

LINE.json: a line of `bench.py`: `roofline.kernels[family]` carries `ms_per_step` / `avg_launch_us` / `frac` (HIP-event time
minus the event pairs' cost measured in that process) and the same with `_raw_events`.  KERNEL_STATS.csv: `rocprofv3
--kernel-trace --stats --output-format csv` of `bench.py --streams 1 --graph 0` (one batch at a time: with batches in flight a
kernel's wall duration includes the time it shares the CUs).  Per family the script adds up the table's TotalDurationNs over
the family's kernels, divides by the number of steps the profiled process ran (= calls of a kernel every step holds once) and
recomputes the fraction of 8 TB/s from the line's algorithmic bytes.

THE check is the SAME-PROCESS one: the line printed by the process that ran under rocprofv3 (`*_under_rocprof.json`) against
that process' table - exit status 1 when the DOMINANT family (the one `roofline.frac` describes) differs by more than --tol.
(r4, three runs on two boxes: 0.996 / 0.997, 0.997 / 1.001, 1.017 / 1.017 for depthwise / pointwise; the raw event time of
the same lines: 1.11-1.15.)  A line from ANOTHER process of the same box can be given too, but then two processes are being
compared: on this pool they differ by up to 10 % whatever is measured (same command, same box, minutes apart: 109.8 k vs
115.8 k images/s), so that comparison is printed with its ratios and only gated when --cross-process-gate is given.
Round 6: the same-process ratios are 1.03-1.07 (rc 1) - under the profiler two consecutive dispatches are now 8.5 us apart
(6.1 us in round 5, 1.6 us un-profiled), and the calibration's back-to-back launches count that gap as kernel time, so the
event pairs' cost is under-estimated THERE; the un-profiled lines of the same box agree with the table to 0.1-2.4 %, and
tools/refresh_profiles.sh gates the un-profiled one-stream line too (--cross-process-gate).
"""
import csv
import json
import sys

FAMILIES = {
    "pwconv": ("pwconv_stream_kernel", "pwconv_sample_kernel", "pwconv_split_kernel", "quant_transpose_i8_kernel",
               "pwconv_i8_kernel", "qconv_pw", "pw_stat_kernel"),
    # (round 6: a recompute pair's fused launch stands for the depthwise layer, its statistic pass for the pointwise layer)
    "dwconv": ("dwconv3x3", "qconv_dw", "pwdw_kernel"),
    "stem": ("stem_mfma_kernel", "stem_conv3x3s2_kernel", "stem7_pool_kernel", "stem7_pool_lds_kernel", "stem3_rows_kernel"),
    "pool": ("gap_stat",),
    "dense": ("pwconv_rows_kernel",),
    "conv3x3": ("conv3x3_i8_kernel",),
    "stat": ("absmax_per_sample_kernel",),
    "apply_online": ("act_apply_kernel<true",),
    "bn_act": ("bn_act_stat_kernel", "bn_act_range_kernel"),
    "qrange": ("qconv_range",),
}
HBM_PEAK_GBS = 8000.0


def main(argv):
    tol = 0.03
    if "--tol" in argv:
        tol = float(argv[argv.index("--tol") + 1])
    line = json.load(open(argv[0]))
    rows = list(csv.DictReader(open(argv[1])))
    # a kernel every step holds exactly once: the pooling pass, or - since the last 1x1 stores the pooled means itself
    #  and that pass is gone from the default workload - the classifier's launch
    steps = 0
    for steps_kernel in ([argv[argv.index("--steps-from") + 1]] if "--steps-from" in argv else ["gap_stat", "pwconv_rows_kernel"]):
        steps = sum(int(r["Calls"]) for r in rows if steps_kernel in r["Name"])
        if steps:
            break
    if not steps:
        print("no '%s' kernel in the table: cannot tell how many steps the profiled process ran" % steps_kernel)
        return 2
    kernels = line["roofline"]["kernels"]
    step_us = sum(k["ms_per_step"] for k in kernels.values()) * 1e3
    print("steps in the profiled process: %d; families of the line: %s" % (steps, ", ".join(kernels)))
    same_process = "under_rocprof" in argv[0] or "--same-process" in argv
    gate = same_process or "--cross-process-gate" in argv
    print("%-10s %14s %14s %14s %8s %8s %10s %10s %10s" % ("family", "line us/step", "raw us/step", "rocprof us/step", "line/rp",
                                                          "raw/rp", "frac(line)", "frac(raw)", "frac(csv)"))
    bad = 0
    for fam, k in kernels.items():
        names = FAMILIES.get(fam)
        if not names:
            continue
        tot_ns = sum(float(r["TotalDurationNs"]) for r in rows
                     if any(n in r["Name"] for n in names) and not (fam == "pwconv" and "pwconv_rows_kernel" in r["Name"]))
        if tot_ns == 0:
            continue
        rp_us = tot_ns / steps / 1e3
        if "avg_launch_us_launch_overhead_removed" in k:      # a line of rounds 3 / early 4: the plain keys held the raw events
            k = dict(k, ms_per_step_raw_events=k["ms_per_step"], frac_raw_events=k["frac"],
                     ms_per_step=k["ms_per_step"] * k["avg_launch_us_launch_overhead_removed"] / k["avg_launch_us"],
                     avg_launch_us=k["avg_launch_us_launch_overhead_removed"], frac=k["frac_launch_overhead_removed"])
        ev_us = k["ms_per_step"] * 1e3
        raw_us = k.get("ms_per_step_raw_events", k["ms_per_step"]) * 1e3
        launches_per_step = ev_us / max(k["avg_launch_us"], 1e-9)
        frac_csv = k["algorithmic_bytes_per_launch"] * launches_per_step / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        ratio = ev_us / rp_us
        flag = ""
        is_dom = (line["roofline"].get("kernel") or "") == k.get("kernel")
        if abs(ratio - 1.0) > tol and ev_us >= 0.05 * step_us:
            bad += 1 if (is_dom and gate) else 0
            flag = "  <-- beyond %.0f %%%s" % (tol * 100, " (the dominant family)" if is_dom else "")
        elif is_dom:
            flag = "  (the dominant family)"
        print("%-10s %14.1f %14.1f %14.1f %8.3f %8.3f %10.4f %10.4f %10.4f%s"
              % (fam, ev_us, raw_us, rp_us, ratio, raw_us / rp_us, k["frac"], k.get("frac_raw_events", k["frac"]), frac_csv, flag))
    print("(%s)" % ("line and table from the SAME process: gated at %.0f %%" % (tol * 100) if same_process else
                    "line and table from two processes of one box: printed%s" % (", gated" if gate else ", not gated")))
    dom = line["roofline"].get("kernel")
    print("line: roofline.frac %.4f (%s)" % (line["roofline"]["frac"], (dom or "")[:60]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
