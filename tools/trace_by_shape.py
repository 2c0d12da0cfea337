"""Kernel-trace breakdown keyed by (kernel, grid): separates the GEMM shapes that share one kernel instantiation.

    python tools/trace_by_shape.py gpurun_out/.../s_kernel_trace.csv [out.csv]
"""
import collections
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_window import load_window, short


def main():
    win, nmicro, wall = load_window(sys.argv[1])
    agg = collections.defaultdict(list)
    order = {}
    for i, r in enumerate(win):
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        k = (short(r["Kernel_Name"]), g)
        order.setdefault(k, i)
        agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = [("kernel", "grid", "calls_per_microbatch", "ms_per_microbatch", "median_us", "min_us", "max_us", "first_index")]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        out.append((k[0], "x".join(map(str, k[1])), round(len(v) / nmicro, 2), round(sum(v) / 1e3 / nmicro, 3),
                    round(v2[len(v2) // 2], 1), round(v2[0], 1), round(v2[-1], 1), order[k]))
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
