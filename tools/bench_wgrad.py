"""Weight-gradient products dW[N_out, K_in] += dY^T . X (both operands token-major = k-major) of a pass of P micro-batches: the
any-layout kernels of gemm3.hip (cfg 105: 256 x 128 loader ring) against the k-major form of the 256 x 256 8-phase kernel (cfg 106),
cold operands, HIP-event medians. Usage: python tools/bench_wgrad.py [P]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    T = P * 1200
    torch.manual_seed(0)
    for name, n_out, k_in in (("gate|up", 19456, 2560), ("down", 2560, 9728), ("q|k|v", 6144, 2560), ("o", 2560, 4096)):
        nset = 3
        dys = [torch.randn(T, n_out, device="cuda").to(torch.bfloat16) for _ in range(nset)]
        xs = [torch.randn(T, k_in, device="cuda").to(torch.bfloat16) for _ in range(nset)]
        g = torch.zeros(n_out, k_in, device="cuda", dtype=torch.bfloat16)
        ref = None
        line = f"{name:8s} [{n_out:5d}, {k_in:5d}] x {T} tokens "
        res = {}
        for cfg in (105, 106):
            ops.gemm_force_config(cfg)
            g.zero_()
            ops.gemm_raw(dys[0], xs[0], g, n_out, k_in, T, n_out, k_in, k_in, transA=True, transB=True, accumulate=True)
            if ref is None:
                ref = g.float().clone()
            err = ((g.float() - ref).norm() / ref.norm()).item()
            ts = []
            for _ in range(5):
                for i in range(2):
                    ops.gemm_raw(dys[i % nset], xs[i % nset], g, n_out, k_in, T, n_out, k_in, k_in, transA=True, transB=True, accumulate=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(6):
                    ops.gemm_raw(dys[i % nset], xs[i % nset], g, n_out, k_in, T, n_out, k_in, k_in, transA=True, transB=True, accumulate=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 6)
            t = sorted(ts)[len(ts) // 2]
            res[cfg] = t
            line += f"| cfg {cfg}: {t * 1e3:7.1f} us {2.0 * n_out * k_in * T / t / 1e9:7.1f} TF/s (rel diff {err:.1e}) "
        print(line, flush=True)
    ops.gemm_force_config(-3)


if __name__ == "__main__":
    main()
