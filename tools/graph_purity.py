"""Counts the kernels that are not ours between the sampler-step graph replays of a rocprofv3 kernel trace.

    python tools/graph_purity.py gpurun_out/r03_pipeline_stats/x_kernel_trace.csv [calls]

The timed run of tools/pipeline_bench.py is the LAST `calls` UNet calls (39 for 20 Heun steps) in the trace (the first run is the
eager warm-up that records the graphs); every call begins with `sampler_load_kernel`.  Between the first and the
last of those the trace must hold ddx kernels only: an ATen kernel there means a host-side tensor op crept back
into the step.
"""
import csv
import sys


def main() -> None:
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 39
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    loads = [i for i, n in enumerate(names) if "sampler_load" in n]
    if len(loads) < steps:
        raise SystemExit(f"only {len(loads)} sampler_load kernels in the trace")
    lo, hi = loads[-steps], loads[-1]
    foreign = [n for n in names[lo:hi] if "ddx" not in n and "sampler_load" not in n and "lincomb3_dev" not in n
               and "step_advance" not in n]
    span_us = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
    print(f"graph purity: {hi - lo} kernels over the last {steps} UNet calls of the sampler "
          f"({span_us / (steps - 1):.0f} us per call under the profiler), {len(foreign)} not ours"
          + (": " + ", ".join(sorted(set(f[:60] for f in foreign))) if foreign else ""))


if __name__ == "__main__":
    main()
