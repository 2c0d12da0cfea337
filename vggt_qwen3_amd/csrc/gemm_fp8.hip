// FP8 (OCP e4m3) GEMM for gfx950 on the block-scaled MFMA: C[M,N] = (sx[m] * sw[n]) * sum_k Xq[m,k] Wq[n,k]  (+ residual).
//
// BASELINE config C5: Qwen3 linear weights in e4m3 with one fp32 scale per output channel, activations quantised per
// token on the fly, fp32 accumulation. v_mfma_scale_f32_16x16x128_f8f6f4 with unit (E8M0 = 127) block scales runs at
// twice the bf16 MFMA rate; the real scales are per row / per column and are applied once, in fp32, in the epilogue.
//
// Same skeleton as the bf16 LDS-DMA kernel (gemm2.hip): a tile row is 128 BYTES in LDS either way, so one K step is now
// 128 deep instead of 64 - half the staging bytes and half the LDS fragment traffic per FLOP. 128x128 tile, 8 compute
// waves (32x64 each) + 2 DMA-loader waves, 4-stage ring, one s_barrier per K step, XOR swizzle applied to the DMA source
// address and again on the fragment reads. A lane's MFMA operand is 32 consecutive k of one row = two 16-byte chunks.
#include "gemm_common.h"
#include "vq3_hip.h"

namespace vq3gemm {
namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int BKB = 128;   // bytes (= fp8 elements) of K per stage row

template <int BM, int BN, int WM, int WN, int NSTAGE, int NLOAD>
__global__ __launch_bounds__(64 * (WM * WN + NLOAD), (WM * WN + NLOAD + 3) / 4) void gemm_fp8_kernel(
    GemmParams p, const float* __restrict__ rowscale, const float* __restrict__ colscale) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NPIECE = (BM + BN) / 8;
  constexpr int PPW = NPIECE / NLOAD;
  static_assert(NPIECE % NLOAD == 0 && NSTAGE >= 3 && NLOAD > 0, "loader-ring configuration");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= NW;
  const int iw = loader ? wid - NW : 0;
  const int wm = (wid % NW) / WN, wn = wid % WN;
  int m0, n0;
  tile_coords(p, BM, BN, m0, n0);
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* B = reinterpret_cast<const char*>(p.B);
  const int nt = p.K / BKB;
  const int last = nt - 1;

  if (loader) {
    const int prow = lane >> 3;
    const int kch = (lane & 7) ^ prow;
    const char* gsrc[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int pi = iw + NLOAD * j;
      const int trow = pi * 8 + prow;
      if (trow < BM) {
        int r = m0 + trow; r = r < p.M ? r : p.M - 1;
        gsrc[j] = A + (long)r * p.lda + kch * 16;
      } else {
        int r = n0 + (trow - BM); r = r < p.N ? r : p.N - 1;
        gsrc[j] = B + (long)r * p.ldb + kch * 16;
      }
    }
    auto issue = [&](int tile, int stage) {
      char* sb = smem + stage * STAGE;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int pi = iw + NLOAD * j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (long)tile * BKB),
                                         (__attribute__((address_space(3))) void*)(sb + pi * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i)
      if (i < nt) issue(i, i);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
      const int newer = (last - t) < (NSTAGE - 2) ? (last - t) : (NSTAGE - 2);
      if (newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
      else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + NSTAGE - 1 <= last) {
        int s2 = stage + NSTAGE - 1; s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
        issue(t + NSTAGE - 1, s2);
      }
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    return;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand of lane (fr, fq): row fr of the 16-row slab, k = 32 fq .. 32 fq + 31 -> 16-byte chunks 2 fq and 2 fq + 1
  const int fr = lane & 15, fq = lane >> 4;
  int a_lo[TM], a_hi[TM], b_lo[TN], b_hi[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * (BM / WM) + i * 16 + fr;
    a_lo[i] = row * 128 + (((2 * fq) ^ (row & 7)) << 4);
    a_hi[i] = row * 128 + (((2 * fq + 1) ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * (BN / WN) + j * 16 + fr;
    b_lo[j] = BM * 128 + row * 128 + (((2 * fq) ^ (row & 7)) << 4);
    b_hi[j] = BM * 128 + row * 128 + (((2 * fq + 1) ^ (row & 7)) << 4);
  }
  constexpr int UNIT = 0x7f7f7f7f;   // E8M0 127 = 2^0 in every byte: the block scales are not used
  int stage = 0;
  for (int t = 0; t < nt; ++t) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const char* sb = smem + stage * STAGE;
    i32x8 xa[TM], wb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + a_lo[i]);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(sb + a_hi[i]);
      xa[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + b_lo[j]);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(sb + b_hi[j]);
      wb[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wb[j], xa[i], acc[i][j], 0, 0, 0, UNIT, 0, UNIT);
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WM) + i * 16 + fr;
    if (m >= p.M) continue;
    const float sx = rowscale[m];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= sx * colscale[n + r < p.N ? n + r : p.N - 1];
      store_quad<false>(p, 0, 0, m, n, v);
    }
  }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int NLOAD>
int launch_fp8(GemmParams& p, const float* rs, const float* cs, hipStream_t stream) {
  constexpr int SMEM = NSTAGE * (BM + BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_fp8_kernel<BM, BN, WM, WN, NSTAGE, NLOAD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      vq3_set_error("gemm fp8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  choose_tile_order(p, BM, BN, 1);
  hipLaunchKernelGGL((gemm_fp8_kernel<BM, BN, WM, WN, NSTAGE, NLOAD>), dim3(p.mtiles * p.ntiles), dim3(64 * (WM * WN + NLOAD)),
                     SMEM, stream, p, rs, cs);
  return 0;
}

// ---------------------------------------------------------------------------------------------------- quantisation
// per-row e4m3 quantisation of a bf16 matrix: scale[r] = amax_r / 448 (1 for an all-zero row), q = rne_e4m3(x * (448/amax))
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ x, long ldx, int K, uint8_t* __restrict__ q,
                                                        long ldq, float* __restrict__ scale) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const bf16_t* xr = x + r * ldx;
  float amax = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      amax = fmaxf(amax, fabsf(__builtin_bit_cast(float, v[j] << 16)));
      amax = fmaxf(amax, fabsf(__builtin_bit_cast(float, v[j] & 0xffff0000u)));
    }
  }
  amax = block_max<4>(amax, red);
  const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
  if (threadIdx.x == 0) scale[r] = amax > 0.f ? amax / 448.0f : 1.0f;
  uint8_t* qr = q + r * ldq;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + k);
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = __builtin_bit_cast(float, v[j] << 16) * inv;
      f[2 * j + 1] = __builtin_bit_cast(float, v[j] & 0xffff0000u) * inv;
    }
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    *reinterpret_cast<u32x2*>(qr + k) = u32x2{(unsigned)w0, (unsigned)w1};
  }
}

}  // namespace
}  // namespace vq3gemm

extern "C" int vq3_gemm_fp8_nt(const void* Xq, const float* x_scale, const void* Wq, const float* w_scale, void* C,
                               const void* residual, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldw, int64_t ldc,
                               int64_t ldr, void* stream) {
  using namespace vq3gemm;
  VQ3_CHECK_ARG(Xq && x_scale && Wq && w_scale && C, "gemm_fp8: null pointer");
  VQ3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 128 == 0, "gemm_fp8: need K %% 128 == 0 (M=%d N=%d K=%d)", M, N, K);
  VQ3_CHECK_ARG(ldx >= K && ldw >= K && ldx % 16 == 0 && ldw % 16 == 0 && ldc >= N, "gemm_fp8: bad leading dimensions");
  VQ3_CHECK_ARG((uintptr_t)Xq % 16 == 0 && (uintptr_t)Wq % 16 == 0, "gemm_fp8: operands must be 16-byte aligned");
  GemmParams p{};
  p.A = (const bf16_t*)Xq; p.B = (const bf16_t*)Wq; p.C = C; p.R = residual;
  p.M = M; p.N = N; p.K = K; p.lda = (int)ldx; p.ldb = (int)ldw; p.ldc = (int)ldc; p.ldr = (int)ldr;
  p.nb2 = 1; p.b2divB = 1; p.kper = K; p.nsplit = 1; p.alpha = 1.0f;
  p.vec_ok = (ldc % 4 == 0 && (uintptr_t)C % 8 == 0 && (!residual || (ldr % 4 == 0 && (uintptr_t)residual % 8 == 0))) ? 1 : 0;
  // 256x128 tiles halve the L2 -> LDS bytes per FLOP (the fp8 MFMA rate makes the 128x128 tile L2-bound); they pay off
  // once the grid still fills the chip
  static int force = -1;
  if (force < 0) {
    const char* e = getenv("VQ3_FP8_TILE");
    force = e ? atoi(e) : 0;
  }
  const long tiles256 = (long)((M + 255) / 256) * ((N + 127) / 128);
  const bool big = force ? force == 256 : tiles256 >= 200;
  const int rc = big ? launch_fp8<256, 128, 4, 2, 3, 2>(p, x_scale, w_scale, (hipStream_t)stream)
                     : launch_fp8<128, 128, 4, 2, 4, 2>(p, x_scale, w_scale, (hipStream_t)stream);
  if (rc) return rc;
  VQ3_CHECK_LAUNCH("gemm_fp8");
  return 0;
}

extern "C" int vq3_quant_fp8_rows(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, void* q, int64_t ldq, float* scale,
                                  void* stream) {
  VQ3_CHECK_ARG(x_bf16 && q && scale, "quant_fp8_rows: null pointer");
  VQ3_CHECK_ARG(rows > 0 && K > 0 && K % 8 == 0 && ldx >= K && ldq >= K && ldx % 8 == 0 && ldq % 8 == 0,
                "quant_fp8_rows: need K %% 8 == 0 and 8-aligned leading dimensions");
  VQ3_CHECK_ARG(rows < (1l << 31), "quant_fp8_rows: too many rows");
  hipLaunchKernelGGL(vq3gemm::quant_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x_bf16, (long)ldx, K, (uint8_t*)q, (long)ldq, scale);
  VQ3_CHECK_LAUNCH("quant_fp8_rows");
  return 0;
}
