#!/usr/bin/env python3
"""Recompute every `roofline_all` row of a bench run from the rocprofv3 kernel statistics of THE SAME command.

    rocprofv3 --kernel-trace --stats -d DIR -- python3 bench.py --steps K --warmup W --no-fast --no-r03-leg \
        --no-precision-block --no-nxn-legs --no-cpu-baseline --no-train-leg          # -> DIR/.../*_kernel_stats.csv, bench_detail.json
    python tools/roofline_from_rocprof.py profiles/r06/rocprof_kernel_stats.csv profiles/r06/bench_detail_profiled.json

bench.py's rows divide ALGORITHMIC flops (2 * M * K * Cout of the convolution a launch computes; bytes for the HBM-bound rows) by
HIP-event launch times it samples inside the run.  This tool divides the same algorithmic work — `algorithmic_flops_per_step` per
device-kernel symbol from bench_detail.json, times the K + W steps the command ran — by the profiler's TotalDurationNs of the
symbol, so the committed line can be checked against the committed profile without a GPU (tests/test_host_logic.py does: 5 %).
The restricted command runs nothing but the headline leg, so every launch of an encoder symbol belongs to one of its K + W steps.
Prints one row per symbol and, with --json, the table as JSON."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def recompute(stats_csv, detail_json):
    import bench  # symbol_matches: bench.py's row name <-> the profiler's kernel name (no GPU needed to import it)

    detail = json.load(open(detail_json))
    line, head = detail["line"], detail["headline"]
    steps = int(line["steps"]) + int(line["warmup"])
    precision = line["dtype"]
    with open(stats_csv, newline="") as f:
        prof = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(f)]
    rows = []
    for k in head["roofline_all"]:
        if "algorithmic_flops_per_step" not in k or k["kernel"].startswith("encoder convolutions"):
            continue
        hit = [(n, c, t) for n, c, t in prof if bench.symbol_matches(k["kernel"], n, precision)]
        if not hit:
            rows.append({"kernel": k["kernel"], "bench_achieved": k["achieved"], "rocprof_achieved": None})
            continue
        total_s = sum(t for _, _, t in hit) * 1e-9
        calls = sum(c for _, c, _ in hit)
        ach = k["algorithmic_flops_per_step"] * steps / total_s / 1e12
        rows.append({"kernel": k["kernel"], "unit": "TFLOP/s", "peak": k["peak"], "bench_achieved": k["achieved"], "bench_frac": k["frac"],
                     "rocprof_achieved": ach, "rocprof_frac": ach / k["peak"], "rocprof_frac_of_dense_peak": ach / 2500.0,
                     "rocprof_calls": calls, "rocprof_calls_per_step": calls / steps, "bench_launches_per_step": k["launches_per_step"],
                     "rocprof_ms_per_step": total_s * 1e3 / steps, "bench_ms_per_step": k.get("ms_per_step_single_stream"),
                     "rel_diff": ach / k["achieved"] - 1.0})
    kernel_ms = sum(t for _, _, t in prof) * 1e-6 / steps
    fam = [n for n in ("pw_x3_kernel", "bneck_x3_kernel", "pw_chain_x3_kernel", "maxpool", "res2_x3_kernel")]
    streaming_ms = sum(t for n, _, t in prof if any(x in n for x in fam)) * 1e-6 / steps
    dominant = line["roofline"]["kernel"]
    dom = next((r for r in rows if r["kernel"] == dominant), None)
    return {"steps_profiled": steps, "precision": precision, "rows": rows, "dominant": dom, "line_frac": line["roofline"]["frac"],
            "kernel_ms_per_step": kernel_ms, "streaming_family_share": streaming_ms / kernel_ms if kernel_ms else None,
            "streaming_family": "pw_x3 + bneck_x3 + pw_chain_x3 + maxpool (+ res2_x3) device time over all kernel time"}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if len(args) != 2:
        sys.exit(__doc__)
    out = recompute(*args)
    if "--json" in sys.argv:
        print(json.dumps(out, indent=1))
        return
    print("%d steps profiled (%s); kernel time %.1f ms per step; streaming family %.1f %% of it" % (
        out["steps_profiled"], out["precision"], out["kernel_ms_per_step"], 100.0 * (out["streaming_family_share"] or 0.0)))
    print("%-42s %10s %10s %8s %8s %8s" % ("kernel", "bench TF/s", "rocprof", "frac", "of 2.5PF", "diff"))
    for r in out["rows"]:
        if r["rocprof_achieved"] is None:
            print("%-42s %10.1f %10s" % (r["kernel"], r["bench_achieved"], "-"))
            continue
        print("%-42s %10.1f %10.1f %8.3f %8.3f %+7.1f%%" % (r["kernel"], r["bench_achieved"], r["rocprof_achieved"], r["rocprof_frac"],
                                                           r["rocprof_frac_of_dense_peak"], 100.0 * r["rel_diff"]))
    d = out["dominant"]
    if d and d["rocprof_achieved"]:
        print("dominant %s: line frac %.3f, from the profile %.3f (%+.1f %%)" % (d["kernel"], out["line_frac"], d["rocprof_frac"],
                                                                               100.0 * (d["rocprof_frac"] / out["line_frac"] - 1.0)))


if __name__ == "__main__":
    main()
