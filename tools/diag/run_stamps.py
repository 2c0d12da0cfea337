"""Diagnostic build (never part of the product library): s_memtime stamps around the wait / DMA-issue / compute
phases of the v2 GEMM main loop. Prints mean shader cycles per K step per wave for each phase."""
import ctypes as C, subprocess, sys
from pathlib import Path
import torch
here = Path(__file__).resolve().parent
so = here / "gemm_stamp.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950",
                "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-I", str(here.parents[1] / "include"),
                str(here / "gemm_stamp.hip"), "-o", str(so)], check=True)
lib = C.CDLL(str(so))
lib.diag_gemm.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
NWS = {0: 8, 1: 8, 2: 4, 3: 8, 6: 4, 7: 8}
TILE = {0: (256, 128), 1: (128, 256), 2: (128, 128), 3: (128, 128), 6: (128, 128), 7: (128, 128)}
for (M, N, K) in [(1200, 6144, 2560), (1200, 19456, 2560), (6174, 4096, 1024), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for cfg in (0, 3, 7):
        bm, bn = TILE[cfg]
        nblk = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        st = torch.zeros(nblk * NWS[cfg] * 4, device="cuda", dtype=torch.int64)
        for _ in range(3):
            lib.diag_gemm(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), M, N, K, cfg, st.data_ptr())
        torch.cuda.synchronize()
        s = st.view(-1, 4).double()
        nt = K // 64
        w, i, c, tot = (s[:, j].mean().item() / nt for j in range(4))
        print(f"M={M} N={N} K={K} cfg{cfg}: per K-step cycles  wait+barrier {w:7.0f}  issue {i:6.0f}  compute {c:7.0f}  total/nt {tot:7.0f}  (blocks {nblk})")
