"""Correctness + race screen + timing of the opt-in stream-K GEMM (run with VQ3_GEMM_STREAMK=1). For every shape: the
stream-K result against an fp32 reference and against the per-tile kernels, then N repetitions that must be bit-identical
(each tile's summation order is fixed, so any difference is a synchronisation bug), then cold-weight timing of both."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vggt_qwen3_amd import ops  # noqa: E402

SHAPES = [(1200, 2560, 4096), (1200, 2560, 9728), (1200, 2560, 19456), (1200, 2560, 6144), (1200, 4096, 2560), (768, 4096, 4096),
          (200, 2560, 640), (1200, 2560, 64), (130, 8200, 1024)]


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    assert os.environ.get("VQ3_GEMM_STREAMK") == "1", "run with VQ3_GEMM_STREAMK=1"
    ok = True
    for M, N, K in SHAPES:
        torch.manual_seed(M + N + K)
        nb = 4
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        ws = [(torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16) for _ in range(nb)]
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        ref = x.float() @ ws[0].float().t() + res.float()
        ops.disable_streamk()
        base = ops.linear(x, ws[0], residual=res)
        t_base = timeit(lambda i: ops.linear(x, ws[i % nb], residual=res))
        ops.enable_streamk()
        first = ops.linear(x, ws[0], residual=res)
        e_ref = ((first.float() - ref).norm() / ref.norm()).item()
        e_base = ((first.float() - base.float()).norm() / base.float().norm()).item()
        same = True
        for _ in range(reps):
            same &= torch.equal(ops.linear(x, ws[0], residual=res), first)
        t_sk = timeit(lambda i: ops.linear(x, ws[i % nb], residual=res))
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        good = e_ref < 6e-3 and e_base < 6e-3 and same
        ok &= good
        print(f"M={M:5d} N={N:5d} K={K:6d} tiles={tiles:4d}: vs fp32 {e_ref:.2e}  vs per-tile {e_base:.2e}  {reps} reps identical: {same}  "
              f"per-tile {t_base:7.1f} us  stream-K {t_sk:7.1f} us  ({t_base / t_sk:.2f}x){'' if good else '   <-- FAIL'}", flush=True)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
