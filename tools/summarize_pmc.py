"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE do not fit one pass) into a per-kernel summary + the JSON bench.py reports as `roofline.traffic`.

    python tools/summarize_pmc.py <fetch>_counter_collection.csv <write>_counter_collection.csv profiles/r2_pmc "<commit>" "<command>"

Only the steady-state window (tools/prof_window.py) is counted. gfx950 correction applied: FETCH_SIZE tallies 128-B requests at
64 B, i.e. reads exactly half of a wide coalesced stream, so fetched bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE / WRITE_SIZE are in
KiB); WRITE_SIZE is exact. Infinity-Cache hits are included in both (they are memory-side requests of the L2), so this is
L2<->fabric traffic, an upper bound on HBM traffic."""
import collections
import csv
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_window import load_window, short


def load(path):
    win, nmicro, _ = load_window(path)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in win:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])   # one counter per pass -> one row per dispatch
        agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, nmicro


def main():
    fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    command = sys.argv[5] if len(sys.argv) > 5 else None
    (f, nm), (w, _) = load(fetch), load(write)
    rows = []
    for k in sorted(f, key=lambda k: -f[k][2]):
        calls = f[k][0]
        fb = 2.0 * f[k][1] * 1024
        wb = w.get(k, [0, 0.0, 0.0])[1] * 1024
        rows.append(dict(kernel=k, calls_per_microbatch=round(calls / nm, 2), avg_us=f[k][2] / calls, fetch_MB_per_launch=fb / calls / 1e6,
                         write_MB_per_launch=wb / max(1, w.get(k, [1])[0]) / 1e6))
    with open(out + "_summary.csv", "w") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
    g = [r for r in rows if r["kernel"].startswith("gemm_")]
    calls = sum(f[r["kernel"]][0] for r in g)
    tot = sum((r["fetch_MB_per_launch"] + r["write_MB_per_launch"]) * f[r["kernel"]][0] for r in g)
    json.dump({"kernel": "gemm (all instantiations behind vq3_gemm_bf16_nt)", "launches": calls, "traffic_bytes_per_launch": tot / calls * 1e6,
               "commit": commit, "command": command,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, steady-state window only; fetched = 2*FETCH_SIZE KiB "
                       "(gfx950), Infinity-Cache hits included (L2<->fabric bytes)"}, open(out + "_traffic.json", "w"), indent=1)
    print(open(out + "_summary.csv").read()[:3000])


if __name__ == "__main__":
    main()
