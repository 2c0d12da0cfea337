"""A/B of vq3_gemm_bf16_nt tile configurations in ONE process (interleaved rounds, random data, cold weights): for every
shape of the Stage-1 step each candidate config is checked against an fp32 torch matmul and timed with HIP events;
prints median TF/s per config and the automatic choice. Usage: python tools/bench_gemm_cfg.py [cfg ...] [--warm]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from vggt_qwen3_amd import ops

SHAPES = [
    ("vggt fc1", 6174, 4096, 1024), ("vggt qkv", 6174, 3072, 1024), ("vggt fc2", 6174, 1024, 4096),
    ("vggt proj", 6174, 1024, 1024), ("qwen gate_up", 1200, 19456, 2560), ("qwen dgrad gu", 1200, 2560, 19456),
    ("qwen down", 1200, 2560, 9728), ("qwen qkv", 1200, 6144, 2560), ("qwen dgrad qkv", 1200, 2560, 6144),
    ("qwen o", 1200, 2560, 4096), ("qwen dgrad o", 1200, 4096, 2560), ("perc ffn1", 768, 16384, 4096),
    ("perc ffn2", 768, 4096, 16384), ("perc kv", 768, 8192, 4096), ("square 4096", 4096, 4096, 4096),
    ("square 8192", 8192, 8192, 8192),
]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cfgs = [int(a) for a in args] or [-3, 20]
    warm = "--warm" in sys.argv
    only = [a[7:] for a in sys.argv if a.startswith("--only=")]
    extra = [tuple(int(x) for x in a[8:].split(",")) for a in sys.argv if a.startswith("--shape=")]
    shapes = [("custom", *e) for e in extra] if extra else SHAPES
    torch.manual_seed(0)
    for name, M, N, K in shapes:
        if only and not any(o in name for o in only):
            continue
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        ncopy = 1 if warm else max(1, int(6e8 // (N * K * 2)) + 1)
        ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(ncopy)]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = (a[:512].float() @ ws[0].float().t())
        res = {}
        for cfg in cfgs:
            ops.gemm_force_config(cfg)
            out.zero_()
            ops.linear(a, ws[0], out=out)
            err = ((out[:512].float() - ref).abs().max() / ref.abs().max()).item()
            # the last rows too (M edge)
            ref2 = a[-64:].float() @ ws[0].float().t()
            err2 = ((out[-64:].float() - ref2).abs().max() / ref2.abs().max()).item()
            res[cfg] = [max(err, err2), []]
        rounds = 5
        it = max(10, ncopy)
        for _ in range(rounds):
            for cfg in cfgs:
                ops.gemm_force_config(cfg)
                for i in range(2):
                    ops.linear(a, ws[i % ncopy], out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(it):
                    ops.linear(a, ws[i % ncopy], out=out)
                e1.record()
                torch.cuda.synchronize()
                res[cfg][1].append(e0.elapsed_time(e1) / it)
        line = f"{name:16s} M={M:6d} N={N:6d} K={K:6d} "
        for cfg in cfgs:
            t = sorted(res[cfg][1])
            med = t[len(t) // 2]
            line += f"| cfg {cfg:3d}: {med * 1e3:7.1f} us {2.0 * M * N * K / med / 1e9:7.1f} TF/s err {res[cfg][0]:.1e} "
        print(line, flush=True)
    ops.gemm_force_config(-3)


if __name__ == "__main__":
    main()
