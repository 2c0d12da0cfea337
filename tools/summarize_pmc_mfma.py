"""MFMA-pipe utilisation and LDS bank conflicts per kernel from two rocprofv3 --pmc passes of bench.py:

    pass 1: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      pass 2: --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
    python tools/summarize_pmc_mfma.py <pass1>_counter_collection.csv <pass2>_counter_collection.csv profiles/r2_pmc_mfma_lds.csv

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the fraction of SIMD-cycles the matrix pipe was busy
while the kernel ran (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs). Steady-state window only (tools/prof_window.py)."""
import collections
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_window import load_window, short


def load(path):
    win, _, _ = load_window(path)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    t = collections.defaultdict(float)
    for r in win:
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = r.get("Dispatch_Id") or r["Start_Timestamp"]
        if key not in n[k]:
            n[k].add(key)
            t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, {k: len(v) for k, v in n.items()}, t


def main():
    a, na, ta = load(sys.argv[1])
    b, nb, _ = load(sys.argv[2])
    rows = [("kernel", "launches_in_window", "avg_us", "mfma_util", "lds_bank_conflict_over_lds_active")]
    for k in sorted(a, key=lambda k: -ta[k]):
        busy, gui = a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a[k].get("GRBM_GUI_ACTIVE", 0.0)
        conf, act = b.get(k, {}).get("SQ_LDS_BANK_CONFLICT", 0.0), b.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0)
        rows.append((k, na[k], round(ta[k] / na[k], 1), round(busy / (gui / 8 * 1024), 4) if gui else "",
                     round(conf / act, 4) if act else ""))
    with open(sys.argv[3], "w") as fh:
        csv.writer(fh).writerows(rows)
    for r in rows[:30]:
        print(",".join(str(x) for x in r))


if __name__ == "__main__":
    main()
