"""Per-micro-batch kernel-time breakdown from a rocprofv3 --kernel-trace CSV of bench.py (steady-state window: tools/prof_window.py).

    python tools/step_breakdown.py gpurun_out/prof/*/*_kernel_trace.csv [out.csv]
"""
import collections
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_window import load_window, short


def main():
    win, nmicro, wall = load_window(sys.argv[1])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in win:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    busy = sum(v[1] for v in agg.values())
    out = [("kernel", "calls_per_microbatch", "ms_per_microbatch", "avg_us", "pct_of_busy")]
    for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((k, round(c / nmicro, 2), round(ms / nmicro, 3), round(ms * 1e3 / c, 1), round(100 * ms / busy, 1)))
    out.append(("TOTAL busy", round(sum(v[0] for v in agg.values()) / nmicro, 1), round(busy / nmicro, 2), "", 100.0))
    out.append(("window: micro-batches / wall ms per micro-batch (GPU timeline, profiler overhead included)", nmicro, round(wall / nmicro, 2), "", ""))
    for o in out:
        print(",".join(str(x) for x in o))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            csv.writer(fh).writerows(out)


if __name__ == "__main__":
    try:
        main()
    except BrokenPipeError:
        pass
