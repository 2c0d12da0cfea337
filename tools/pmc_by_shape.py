"""Per-shape FETCH / WRITE table from rocprofv3 --pmc passes of a GEMM micro-benchmark: dispatches of one kernel instantiation are
separated by (grid, duration cluster), since several shapes share one instantiation (and the persistent kernels one grid).

    python tools/pmc_by_shape.py <fetch_counter_collection.csv> <write_counter_collection.csv> [hit_counter_collection.csv]
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read: MI355X_MICROARCH.md, HBM section); both in MB per launch."""
import collections
import csv
import re
import sys


def load(path, cname):
    rows = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if r["Counter_Name"] != cname:
                continue
            m = re.search(r"((gemm_v\d|flash_attn_hd64|qwen_flash_\w+)_kernel(<[^>]*>)?)", k)
            if not m:
                continue
            key = (m.group(1), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            rows[key].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, float(r["Counter_Value"])))
    return rows


def clusters(v, gap=1.12):
    v = sorted(v)
    out, cur = [], [v[0]]
    for x in v[1:]:
        if x[0] > cur[-1][0] * gap:
            out.append(cur)
            cur = [x]
        else:
            cur.append(x)
    out.append(cur)
    return out


def main():
    tables = [("FETCH_x2_MB", load(sys.argv[1], "FETCH_SIZE"), 2 * 1024 / 1e6), ("WRITE_MB", load(sys.argv[2], "WRITE_SIZE"), 1024 / 1e6)]
    if len(sys.argv) > 3:
        tables.append(("TCC_HIT_M", load(sys.argv[3], "TCC_HIT_sum"), 1e-6))
        tables.append(("TCC_MISS_M", load(sys.argv[3], "TCC_MISS_sum"), 1e-6))
    for name, d, scale in tables:
        print(name)
        for k in sorted(d):
            for c in clusters(d[k]):
                if len(c) < 20:
                    continue
                us = sorted(x[0] for x in c)[len(c) // 2]
                val = sorted(x[1] for x in c)[len(c) // 2] * scale
                print(f"  {k[0]:48s} nwg={k[1]:6d} n={len(c):5d} {us:8.1f} us {val:9.1f}")


if __name__ == "__main__":
    main()
