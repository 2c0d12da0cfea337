"""Steady-state window of a rocprofv3 CSV of bench.py: the rows between the last two optimiser steps (the big adamw_kernel
launch ends every accumulation cycle) and the number of micro-batches in it (one embed_splice_fwd_kernel per forward pass x micro-batches per merged pass).
Everything before it - warm-up, the GEMM autotuner's trial launches - is left out."""
import csv


def load_window(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: (int(r["Start_Timestamp"]), r.get("Counter_Name", "")))
    ncnt = max(1, len({r["Counter_Name"] for r in rows})) if rows and "Counter_Name" in rows[0] else 1   # rows per dispatch
    big = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]
           and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5_000_000]
    starts = sorted({rows[i]["Start_Timestamp"] for i in big}, key=int)       # one entry per optimiser step
    if len(starts) < 2:
        raise SystemExit("need at least 2 optimiser steps in the trace")
    # the accumulation window with the most micro-batches (bench.py ends with one-micro-batch instrumented windows; the timed
    # region's windows are the long ones); ties go to the latest
    ends = [max(int(rows[i]["End_Timestamp"]) for i in big if rows[i]["Start_Timestamp"] == st) for st in starts]
    splice = sorted(int(r["Start_Timestamp"]) for r in rows if "embed_splice_fwd_kernel" in r["Kernel_Name"])
    best, t_lo, t_hi = -1, ends[-2], ends[-1]
    for a, b in zip(ends[:-1], ends[1:]):
        n = sum(1 for t in splice if a <= t <= b)
        if n >= best:
            best, t_lo, t_hi = n, a, b
    win = [r for r in rows if t_lo <= int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t_hi]
    # one embed_splice_fwd_kernel per forward pass; a pass is VQ3_PROF_MERGE micro-batches when the trainer merges them (default 8)
    import os
    nmicro = max(1, sum(1 for r in win if "embed_splice_fwd_kernel" in r["Kernel_Name"]) // ncnt) * int(os.environ.get("VQ3_PROF_MERGE", "8"))
    wall_ms = (t_hi - t_lo) / 1e6
    return win, nmicro, wall_ms


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("vq3gemm::", "").replace("void ", "")
    for fam in ("gemm_v2_kernel", "gemm_v3_kernel", "gemm_v6_kernel", "gemm_nt_kernel", "gemm_fp8_kernel", "flash_attn_hd64_kernel"):
        if fam in name:
            return fam + "<" + name.split("<")[1].split(">")[0].replace(" ", "") + ">"
    if name.startswith("at::native::"):
        return "torch:" + name.split("<")[0].split("::")[-1]
    return name.split("(")[0][:56]
