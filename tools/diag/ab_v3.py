import os, sys, torch
sys.path.insert(0, "/root/repo")
from vggt_qwen3_amd import ops
M, N, K = 1200, 9728, 2560
nc = 6
As = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(nc)]
Bs = [torch.randn(K, N, device="cuda").to(torch.bfloat16) for _ in range(nc)]
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
def run(i): ops.gemm_raw(As[i % nc], Bs[i % nc], C, M, N, K, K, N, N, transB=True)
for _ in range(3): run(0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(12): run(i)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("VQ3_HIP_LIB", "new"), os.environ.get("VQ3_GEMM_V3_STAGES"), e0.elapsed_time(e1) / 12 * 1e3, "us")
