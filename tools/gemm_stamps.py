"""Where a GEMM tile spends its time (diagnostic build path of gemm6.hip: p.stamps): thread 0 of every workgroup stamps the 100 MHz
wall clock at tile start (0), first operands landed (1), main loop done (2), C image staged in LDS (3), row stores issued (4), its own
stores acknowledged (5). Prints the median / p90 of every segment over the tiles of one launch, per configuration.

    python tools/gemm_stamps.py M N K [cfg ...] [--epi=plain|gelu|res]
Shares only, never an absolute (the stamps serialise what the real kernel overlaps): cdna guide section 7, In-kernel stamps."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    M, N, K = (int(x) for x in args[:3])
    cfgs = [int(a) for a in args[3:]] or [20, 21, 22]
    epi = ([a[6:] for a in sys.argv if a.startswith("--epi=")] or ["plain"])[0]
    torch.manual_seed(0)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.03 for _ in range(8)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    kw = {}
    if epi == "gelu":
        kw = dict(bias=bias, act=ops.ACT_GELU)
    elif epi == "res":
        kw = dict(bias=bias, colscale=bias, residual=res)
    for cfg in cfgs:
        bm, bn = {20: (256, 256), 21: (256, 128), 22: (128, 256)}[cfg]
        ntile = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        st = torch.zeros(ntile * 8, device="cuda", dtype=torch.int64)
        ops.gemm_force_config(cfg)
        os.environ.pop("VQ3_GEMM_STAMP_PTR", None)
        for i in range(3):
            ops.linear(a, ws[i], out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.linear(a, ws[3], out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        t_plain = e0.elapsed_time(e1) * 1e3
        os.environ["VQ3_GEMM_STAMP_PTR"] = hex(st.data_ptr())
        ops.linear(a, ws[4], out=out, **kw)
        torch.cuda.synchronize()
        e0.record()
        ops.linear(a, ws[5], out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        t_st = e0.elapsed_time(e1) * 1e3
        os.environ.pop("VQ3_GEMM_STAMP_PTR", None)
        s = st.view(ntile, 8).cpu().numpy().astype("int64")
        t0 = s[:, 0].min()
        seg = {"prologue (0->1)": s[:, 1] - s[:, 0], "main loop (1->2)": s[:, 2] - s[:, 1], "stage C (2->3)": s[:, 3] - s[:, 2], "  of which own quads (2->6)": s[:, 6] - s[:, 2],
               "issue stores (3->4)": s[:, 4] - s[:, 3], "ack own stores (4->5)": s[:, 5] - s[:, 4], "whole tile (0->5)": s[:, 5] - s[:, 0]}
        print(f"cfg {cfg} M={M} N={N} K={K} epi={epi}: {ntile} tiles, launch {t_plain:.1f} us plain / {t_st:.1f} us stamped; "
              f"kernel span by stamps {(s[:, 5].max() - t0) / 100:.1f} us")
        import numpy as np
        for k, v in seg.items():
            v = v / 100.0
            print(f"    {k:24s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}   max {v.max():7.2f}")
        # per workgroup: idle gaps between consecutive tiles (persistent kernels) and the end-of-launch tail
        wg = (s[:, 7] >> 32)
        xcc = (s[:, 7] & 0xf)
        order = np.lexsort((s[:, 0], wg))
        gaps = []
        for i, j in zip(order[:-1], order[1:]):
            if wg[i] == wg[j]:
                gaps.append((s[j, 0] - s[i, 5]) / 100.0)
        if gaps:
            print(f"    gap between a workgroup's tiles: median {np.median(gaps):.2f} us, max {max(gaps):.2f}")
        # per CU (XCC id, CU / SH / SE id of HW_REG_HW_ID): what lies between one workgroup's last store being issued and the next workgroup's
        # first instruction on the same CU (dispatch + wave launch; the one-tile-per-workgroup kernels only)
        cu = (s[:, 7] & 0xf) | (((s[:, 7] >> 16) & 0xff) << 4)
        order = np.lexsort((s[:, 0], cu))
        g04, g05 = [], []
        for i, j in zip(order[:-1], order[1:]):
            if cu[i] == cu[j] and wg[i] != wg[j]:
                g04.append((s[j, 0] - s[i, 4]) / 100.0)
                g05.append((s[j, 0] - s[i, 5]) / 100.0)
        if g04:
            print(f"    same CU, consecutive workgroups ({len(np.unique(cu))} CUs seen): next start - stores issued: median {np.median(g04):.2f} us p90 {np.percentile(g04, 90):.2f}; "
                  f"next start - stores acknowledged: median {np.median(g05):.2f} us p90 {np.percentile(g05, 90):.2f}")
        ends = np.array([s[wg == w, 5].max() for w in np.unique(wg)])
        print(f"    workgroups finish between {(ends.min() - t0) / 100:.1f} and {(ends.max() - t0) / 100:.1f} us; starts spread {(s[:, 0].min() - t0) / 100:.1f}..{np.percentile(s[:, 0] - t0, 5) / 100:.1f} us (p5)")
        print(f"    XCC ids seen: {sorted(set(xcc.tolist()))}; tiles per XCC: {[int((xcc == x).sum()) for x in sorted(set(xcc.tolist()))]}")
    ops.gemm_force_config(-3)


if __name__ == "__main__":
    main()
