"""Per-shape L2<->fabric traffic of the GEMM launches of one steady-state accumulation window of bench.py, against the algorithmic bytes:

    python tools/pmc_step_by_shape.py <fetch>_counter_collection.csv <write>_counter_collection.csv [out.txt]

Launches are grouped by (kernel instantiation, workgroups); the group is matched to its (M, N, K) through the tile counts of the step's
GEMM shapes (config C2, 8 or 10 micro-batches per pass; launches of one grid that differ in K only are priced as an equal mix). FETCH_SIZE doubled (gfx950 counts a
128-byte request as 64 B); algorithmic bytes = 2 (M K + N K) + 2 M N (bf16 C) per launch - residual / accumulate operands add 2 M N each and
are not included, so a ratio slightly above 1 on those launches is expected."""
import collections
import csv
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from prof_window import load_window

def shapes_for(P):
    """GEMM shapes of a forward/backward pass over P micro-batches of 6 samples (config C2: 1029 tower tokens, 200 text tokens, 128 latents)."""
    t, x, q = P * 6 * 1029, P * 1200, P * 6 * 128
    return [(t, 4096, 1024), (t, 3072, 1024), (t, 1024, 4096), (t, 1024, 1024), (x, 19456, 2560), (x, 2560, 9728),
            (x, 2560, 19456), (x, 9728, 2560), (x, 6144, 2560), (x, 2560, 6144), (x, 4096, 2560), (x, 2560, 4096),
            (19456, 2560, x), (2560, 9728, x), (6144, 2560, x), (2560, 4096, x), (q, 16384, 4096), (q, 4096, 16384),
            (q, 4096, 4096), (q, 8192, 4096), (q, 4096, 2048), (q, 2560, 4096)]


SHAPES = sorted(set(shapes_for(8) + shapes_for(10)))


def split_grid(M, N, K, ncu=256):
    """workgroups of the 256 x 256 kernel with its last round split along K (gemm6.hip: sk_plan), or None."""
    tiles = -(-M // 256) * -(-N // 256)
    r = tiles % ncu
    if r == 0 or r > ncu // 2 or r > 128:
        return None
    sl = min(ncu // r, 4, (K // 64) // 8)
    while sl >= 2 and r * (sl - 1) > 192:
        sl -= 1
    if sl < 2:
        return None
    return tiles - r + ((r + 7) // 8 * 8) * sl


def tile_of(name):
    m = re.search(r"gemm_v6_kernel<(\d), ?(\d)", name)
    if m:
        return 128 * int(m.group(1)), 128 * int(m.group(2))
    m = re.search(r"gemm_v2_kernel<(\d+), ?(\d+)", name)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.search(r"gemm_v3_kernel<(\d+)", name)
    if m:
        return int(m.group(1)), 128
    return None


def main():
    fw, nmicro, _ = load_window(sys.argv[1])
    ww, _, _ = load_window(sys.argv[2])

    def agg(win):
        d = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in win:
            if "gemm_v" not in r["Kernel_Name"]:
                continue
            k = re.search(r"(gemm_v\d_kernel<[^>]*>)", r["Kernel_Name"]).group(1)
            key = (k, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            d[key][0] += 1
            d[key][1] += float(r["Counter_Value"])
            d[key][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        return d
    f, w = agg(fw), agg(ww)
    lines = ["# kernel, workgroups, matched (M,N,K), launches/micro-batch, avg us, fetch MB (x2), write MB, algorithmic MB, (fetch+write)/algorithmic"]
    tot_t, tot_a = 0.0, 0.0
    for key in sorted(f, key=lambda k: -f[k][2]):
        n, fv, us = f[key]
        wv = w.get(key, [0, 0.0, 0.0])[1]
        fmb, wmb = 2 * fv * 1024 / n / 1e6, (wv * 1024 / max(1, w.get(key, [1])[0])) / 1e6
        t = tile_of(key[0])
        match = []
        is_split = re.search(r"gemm_v6_kernel<2, ?2, ?false, ?0, ?true", key[0]) is not None
        if t:
            for (M, N, K) in SHAPES:
                tiles = -(-M // t[0]) * -(-N // t[1])
                if is_split:
                    if split_grid(M, N, K) == key[1]:
                        match.append((M, N, K))
                elif tiles == key[1] or (key[1] in (256, 248) and tiles > key[1]):      # persistent launches: one workgroup per CU
                    match.append((M, N, K))
        alg = None
        if match:
            # several shapes with one grid (the [rows, 2560] outputs of a layer differ in K only): each occurs once per layer and pass, so
            # the group's launches are an equal mix of them
            alg = sum((2.0 * (M * K + N * K) + 2.0 * M * N) / 1e6 for (M, N, K) in match) / len(match)
        ratio = (fmb + wmb) / alg if alg else None
        if alg:
            tot_t += (fmb + wmb) * n; tot_a += alg * n
        lines.append(f"{key[0]:44s} {key[1]:6d} {str(match[0]) if len(match) == 1 else ('mix of %d shapes' % len(match) if match else '-'):24s} "
                     f"{n / nmicro:6.2f} {us / n:8.1f} {fmb:9.1f} {wmb:8.1f} {alg if alg else float('nan'):9.1f} {ratio if ratio else float('nan'):6.2f}")
    if tot_a:
        lines.append(f"# matched launches together: traffic / algorithmic = {tot_t / tot_a:.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 3:
        Path(sys.argv[3]).write_text(out + "\n")


if __name__ == "__main__":
    main()
