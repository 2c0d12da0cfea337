"""Epilogue cost on the VGGT block shapes (M = 6174): plain vs bias / GELU / LayerScale+residual / fused q|k|v epilogue, cold
weights, HIP-event medians. Usage: python tools/bench_epilogue.py [cfg]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops


def timeit(fn, n=20, rounds=5):
    ts = []
    for _ in range(rounds):
        fn(0); fn(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -3
    Mrows = int(sys.argv[2]) if len(sys.argv) > 2 else 6174
    ops.gemm_force_config(cfg)
    torch.manual_seed(0)
    M, C = Mrows, 1024
    nw = 40 if M < 10000 else 12
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    h = torch.randn(M, 4 * C, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    w1 = [torch.randn(4 * C, C, device="cuda").to(torch.bfloat16) * 0.03 for _ in range(nw)]
    w2 = [torch.randn(C, 4 * C, device="cuda").to(torch.bfloat16) * 0.03 for _ in range(nw)]
    wq = [torch.randn(3 * C, C, device="cuda").to(torch.bfloat16) * 0.03 for _ in range(nw)]
    wp = [torch.randn(C, C, device="cuda").to(torch.bfloat16) * 0.03 for _ in range(nw)]
    b4 = torch.randn(4 * C, device="cuda"); b3 = torch.randn(3 * C, device="cuda"); b1 = torch.randn(C, device="cuda")
    ls = torch.randn(C, device="cuda")
    o4 = torch.empty(M, 4 * C, device="cuda", dtype=torch.bfloat16)
    o3 = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
    o1 = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    qn = (torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"))
    N, NH = (1029 if M % 1029 == 0 else M), 16
    cos = torch.randn(33, 32, device="cuda").to(torch.bfloat16); sin = torch.randn(33, 32, device="cuda").to(torch.bfloat16)
    rows = [
        ("fc1 plain", 2.0 * M * 4 * C * C, lambda i: ops.linear(x, w1[i % nw], out=o4)),
        ("fc1 +bias", 2.0 * M * 4 * C * C, lambda i: ops.linear(x, w1[i % nw], bias=b4, out=o4)),
        ("fc1 +bias+gelu", 2.0 * M * 4 * C * C, lambda i: ops.linear(x, w1[i % nw], bias=b4, act=1, out=o4)),
        ("fc2 plain", 2.0 * M * 4 * C * C, lambda i: ops.linear(h, w2[i % nw], out=o1)),
        ("fc2 +bias+ls+res", 2.0 * M * 4 * C * C, lambda i: ops.linear(h, w2[i % nw], bias=b1, colscale=ls, residual=res, out=o1)),
        ("proj plain", 2.0 * M * C * C, lambda i: ops.linear(x, wp[i % nw], out=o1)),
        ("proj +bias+ls+res", 2.0 * M * C * C, lambda i: ops.linear(x, wp[i % nw], bias=b1, colscale=ls, residual=res, out=o1)),
        ("qkv plain", 2.0 * M * 3 * C * C, lambda i: ops.linear(x, wq[i % nw], out=o3)),
        ("qkv +bias", 2.0 * M * 3 * C * C, lambda i: ops.linear(x, wq[i % nw], bias=b3, out=o3)),
        ("qkv fused (bias only)", 2.0 * M * 3 * C * C, lambda i: ops.linear_vit_qkv(x, wq[i % nw], b3, N, NH)),
        ("qkv fused norm+rope", 2.0 * M * 3 * C * C, lambda i: ops.linear_vit_qkv(x, wq[i % nw], b3, N, NH, qn=qn, kn=qn, cos=cos, sin=sin,
                                                                                   tokens_per_frame=N, patch_start=5, Wp=32)),
    ]
    st = ops.rowstats128(x)
    st_o = torch.empty((M, C // 128, 2), device="cuda", dtype=torch.float32)
    c4 = torch.randn(4 * C, device="cuda"); c3 = torch.randn(3 * C, device="cuda")
    lnf = ops.layernorm_fwd
    g1, be1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rows += [
        ("layernorm alone", 0.0, lambda i: lnf(x, g1, be1, 1e-5)),
        ("fc1 +bias+gelu +ln_in", 2.0 * M * 4 * C * C, lambda i: ops.linear(x, w1[i % nw], bias=b4, act=1, out=o4,
                                                                          ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=c4))),
        ("fc2 +bias+ls+res +st_out", 2.0 * M * 4 * C * C, lambda i: ops.linear(h, w2[i % nw], bias=b1, colscale=ls, residual=res, out=o1,
                                                                             ln_fold=ops.ln_fold(stats_out=st_o))),
        ("proj +bias+ls+res +st_out", 2.0 * M * C * C, lambda i: ops.linear(x, wp[i % nw], bias=b1, colscale=ls, residual=res, out=o1,
                                                                          ln_fold=ops.ln_fold(stats_out=st_o))),
        ("qkv fused norm+rope +ln_in", 2.0 * M * 3 * C * C, lambda i: ops.linear_vit_qkv(x, wq[i % nw], b3, N, NH, qn=qn, kn=qn, cos=cos, sin=sin,
                                                                                          tokens_per_frame=N, patch_start=5, Wp=32,
                                                                                          ln_fold=ops.ln_fold(stats_in=st, eps=1e-5, colsum=c3))),
    ]
    for name, fl, fn in rows:
        t = timeit(fn)
        print(f"{name:28s} {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)
    ops.gemm_force_config(-3)


if __name__ == "__main__":
    main()
