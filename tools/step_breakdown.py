"""Per-step kernel-time breakdown from a rocprofv3 --kernel-trace CSV of bench.py: the big adamw_kernel launch ends
every optimiser step, so the dispatches between two consecutive ones are exactly one steady-state step.

    python tools/step_breakdown.py gpurun_out/prof/*/*_kernel_trace.csv [out.csv]
"""
import collections
import csv
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("vq3gemm::", "").replace("void ", "")
    if "gemm_v2_kernel" in name or "gemm_nt_kernel" in name or "gemm_v3_kernel" in name or "gemm_v6_kernel" in name:
        return name.split("<")[0].split("::")[-1] + "<" + name.split("<")[1].split(">")[0].replace(" ", "") + ">"
    if name.startswith("at::native::"):
        return "torch:" + name.split("<")[0].split("::")[-1] + ":" + (name.split("at::native::")[2].split("<")[0] if name.count("at::native::") > 1 else "")
    return name.split("(")[0][:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]
             and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5_000_000]
    if len(marks) < 3:
        raise SystemExit("need at least 3 optimiser steps in the trace")
    lo, hi = marks[-2] + 1, marks[-1] + 1          # last full step
    step = rows[lo:hi]
    wall = (int(step[-1]["End_Timestamp"]) - int(rows[marks[-2]]["End_Timestamp"])) / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    busy = sum(v[1] for v in agg.values())
    out = [("kernel", "calls_per_step", "ms_per_step", "avg_us", "pct_of_busy")]
    for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((k, c, round(ms, 3), round(ms * 1e3 / c, 1), round(100 * ms / busy, 1)))
    out.append(("TOTAL busy", sum(v[0] for v in agg.values()), round(busy, 2), "", 100.0))
    out.append(("step wall (GPU timeline)", "", round(wall, 2), "", ""))
    for o in out:
        print(",".join(str(x) for x in o))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            csv.writer(fh).writerows(out)


if __name__ == "__main__":
    main()
