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
// colmul (optional, [K] f32): x[r, k] * colmul[k] is what gets quantised - the dgrad GEMMs fold the weight's per-output-channel scale
// (which runs along THEIR contraction) into dY this way: dX = t_m * sum_n q(dY[m,n] s_n)[m,n] Wq[n,k]
// One workgroup per row; NCH chunks of 8 elements per thread stay in REGISTERS between the amax pass and the conversion (K <= NCH * 2048:
// the row is read once - d(gate|up) rows are 19 456 wide, 373 MB per pass of 8 micro-batches: 346 us with two reads); NCH = 0: any K, two reads.
template <int NCH>
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ x, long ldx, int K, uint8_t* __restrict__ q,
                                                        long ldq, float* __restrict__ scale, const float* __restrict__ colmul) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const bf16_t* xr = x + r * ldx;
  uint8_t* qr = q + r * ldq;
  auto load8 = [&](int k, float (&f)[8]) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = __builtin_bit_cast(float, v[j] << 16);
      f[2 * j + 1] = __builtin_bit_cast(float, v[j] & 0xffff0000u);
    }
    if (colmul) {
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(colmul + k), c1 = *reinterpret_cast<const f32x4*>(colmul + k + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[j] *= c0[j]; f[4 + j] *= c1[j]; }
    }
  };
  auto store8 = [&](int k, const float (&f)[8], float inv) {
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, w1, true);
    *reinterpret_cast<u32x2*>(qr + k) = u32x2{(unsigned)w0, (unsigned)w1};
  };
  float amax = 0.f;
  if constexpr (NCH > 0) {
    float f[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = threadIdx.x * 8 + c * 2048;
      if (k < K) {
        load8(k, f[c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[c][j]));
      }
    }
    amax = block_max<4>(amax, red);
    const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
    if (threadIdx.x == 0) scale[r] = amax > 0.f ? amax / 448.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = threadIdx.x * 8 + c * 2048;
      if (k < K) store8(k, f[c], inv);
    }
  } else {
    for (int k = threadIdx.x * 8; k < K; k += 2048) {
      float f[8];
      load8(k, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
    amax = block_max<4>(amax, red);
    const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
    if (threadIdx.x == 0) scale[r] = amax > 0.f ? amax / 448.0f : 1.0f;
    for (int k = threadIdx.x * 8; k < K; k += 2048) {
      float f[8];
      load8(k, f);
      store8(k, f, inv);
    }
  }
}

// dst[c, r] = src[r, c] for a matrix of BYTES (the e4m3 weights' W^T copies: the NT operand of the dgrad GEMMs): 64 x 64 tiles through
// LDS, 16-byte global accesses on both sides (R, C, lds, ldd multiples of 16)
__global__ __launch_bounds__(256) void transpose_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int R, int C,
                                                           long lds_, long ldd) {
  __shared__ uint8_t tile[64][68];                 // rows of 17 dwords: the column reads below are conflict-free
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int rr = threadIdx.x >> 2, ch = threadIdx.x & 3;            // 64 rows x 4 chunks of 16 bytes
  if (r0 + rr < R && c0 + ch * 16 < C) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)(r0 + rr) * lds_ + c0 + ch * 16);
    unsigned* tp = reinterpret_cast<unsigned*>(&tile[rr][ch * 16]);
    tp[0] = v[0]; tp[1] = v[1]; tp[2] = v[2]; tp[3] = v[3];
  }
  __syncthreads();
  const int cc = threadIdx.x >> 2;                                  // dst row = source column c0 + cc, 16 consecutive source rows
  if (c0 + cc < C && r0 + ch * 16 < R) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned x = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) x |= (unsigned)tile[ch * 16 + k * 4 + b][cc] << (8 * b);
      w[k] = x;
    }
    *reinterpret_cast<u32x4*>(dst + (long)(c0 + cc) * ldd + r0 + ch * 16) = u32x4{w[0], w[1], w[2], w[3]};
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
  // the 256 x 256 8-phase kernel once its tile grid fills half the chip (VQ3_FP8_V6=0: the loader-ring kernels below, for A/B runs)
  static int v6 = -1;
  if (v6 < 0) { const char* e = getenv("VQ3_FP8_V6"); v6 = e ? atoi(e) : 1; }
  if (v6 && !force && (long)((M + 255) / 256) * ((N + 255) / 256) >= 128) {
    GemmParams q = p;
    q.f8_rs = x_scale; q.f8_cs = w_scale;
    q.sC1 = q.sC2 = q.sR1 = q.sR2 = 0;
    const int rc6 = launch_gemm_v6_f8(q, (hipStream_t)stream);
    if (rc6 > 0) return rc6;
    if (rc6 == 0) {
      VQ3_CHECK_LAUNCH("gemm_fp8(v6)");
      return 0;
    }
  }
  const long tiles256 = (long)((M + 255) / 256) * ((N + 127) / 128);
  const bool big = force ? force == 256 : tiles256 >= 200;
  const int rc = big ? launch_fp8<256, 128, 4, 2, 3, 2>(p, x_scale, w_scale, (hipStream_t)stream)
                     : launch_fp8<128, 128, 4, 2, 4, 2>(p, x_scale, w_scale, (hipStream_t)stream);
  if (rc) return rc;
  VQ3_CHECK_LAUNCH("gemm_fp8");
  return 0;
}

static int quant_rows_impl(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, const float* colmul, void* q, int64_t ldq, float* scale,
                           void* stream) {
  VQ3_CHECK_ARG(x_bf16 && q && scale, "quant_fp8_rows: null pointer");
  VQ3_CHECK_ARG(rows > 0 && K > 0 && K % 8 == 0 && ldx >= K && ldq >= K && ldx % 8 == 0 && ldq % 8 == 0,
                "quant_fp8_rows: need K %% 8 == 0 and 8-aligned leading dimensions");
  VQ3_CHECK_ARG(rows < (1l << 31), "quant_fp8_rows: too many rows");
  VQ3_CHECK_ARG(!colmul || (uintptr_t)colmul % 16 == 0, "quant_fp8_rows: colmul must be 16-byte aligned");
#define VQ3_QUANT_LAUNCH(NCH)                                                                                          \
  hipLaunchKernelGGL(vq3gemm::quant_rows_kernel<NCH>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,       \
                     (const bf16_t*)x_bf16, (long)ldx, K, (uint8_t*)q, (long)ldq, scale, colmul)
  if (K <= 2 * 2048) VQ3_QUANT_LAUNCH(2);
  else if (K <= 5 * 2048) VQ3_QUANT_LAUNCH(5);
  else if (K <= 10 * 2048) VQ3_QUANT_LAUNCH(10);
  else VQ3_QUANT_LAUNCH(0);
#undef VQ3_QUANT_LAUNCH
  VQ3_CHECK_LAUNCH("quant_fp8_rows");
  return 0;
}

extern "C" int vq3_quant_fp8_rows(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, void* q, int64_t ldq, float* scale,
                                  void* stream) {
  return quant_rows_impl(x_bf16, ldx, rows, K, nullptr, q, ldq, scale, stream);
}

extern "C" int vq3_quant_fp8_rows_scaled(const void* x_bf16, int64_t ldx, int64_t rows, int32_t K, const float* colmul, void* q, int64_t ldq,
                                         float* scale, void* stream) {
  VQ3_CHECK_ARG(colmul != nullptr, "quant_fp8_rows_scaled: null column multipliers");
  return quant_rows_impl(x_bf16, ldx, rows, K, colmul, q, ldq, scale, stream);
}

extern "C" int vq3_transpose_u8(const void* src, void* dst, int32_t R, int32_t C, int64_t lds, int64_t ldd, void* stream) {
  VQ3_CHECK_ARG(src && dst && R > 0 && C > 0, "transpose_u8: bad arguments");
  VQ3_CHECK_ARG(R % 16 == 0 && C % 16 == 0 && lds >= C && ldd >= R && lds % 16 == 0 && ldd % 16 == 0 && (uintptr_t)src % 16 == 0 &&
                    (uintptr_t)dst % 16 == 0, "transpose_u8: extents, leading dimensions and pointers must be multiples of 16");
  hipLaunchKernelGGL(vq3gemm::transpose_u8_kernel, dim3((R + 63) / 64, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)src, (uint8_t*)dst, R, C, (long)lds, (long)ldd);
  VQ3_CHECK_LAUNCH("transpose_u8");
  return 0;
}

// e4m3 GEMM with the fused epilogues of the bf16 path (config C5: forward AND dgrad projections of the text model):
//   mode 0: C = (xs ws) Xq Wq^T (+ residual)                      mode 1: SwiGLU forward  - Wq = gate|up weight [2 I, K], C = act [M, I], gu out
//   mode 2: SwiGLU backward - the product is d(act) [M, N], never stored: dgu [M, 2 N] from the saved gate|up
// w_scale may be NULL (= 1): the dgrad GEMMs fold the weight's per-output-channel scales into dY before quantising it.
extern "C" int vq3_gemm_fp8_ex(const vq3_gemm_fp8_desc* d, void* stream) {
  using namespace vq3gemm;
  VQ3_CHECK_ARG(d && d->Xq && d->x_scale && d->Wq, "gemm_fp8_ex: null pointer");
  VQ3_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->K % 128 == 0, "gemm_fp8_ex: need K %% 128 == 0");
  VQ3_CHECK_ARG(d->ldx >= d->K && d->ldw >= d->K && d->ldx % 16 == 0 && d->ldw % 16 == 0, "gemm_fp8_ex: bad leading dimensions");
  VQ3_CHECK_ARG((uintptr_t)d->Xq % 16 == 0 && (uintptr_t)d->Wq % 16 == 0, "gemm_fp8_ex: operands must be 16-byte aligned");
  VQ3_CHECK_ARG(d->mode >= 0 && d->mode <= 2, "gemm_fp8_ex: mode 0..2");
  if (d->mode == 0) {
    VQ3_CHECK_ARG(d->C != nullptr, "gemm_fp8_ex: null output");
    if (d->w_scale) return vq3_gemm_fp8_nt(d->Xq, d->x_scale, d->Wq, d->w_scale, d->C, d->residual, d->M, d->N, d->K, d->ldx, d->ldw, d->ldc, d->ldr, stream);
  }
  GemmParams p{};
  p.A = (const bf16_t*)d->Xq; p.B = (const bf16_t*)d->Wq; p.C = d->C; p.R = d->residual;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = (int)d->ldx; p.ldb = (int)d->ldw; p.ldc = (int)d->ldc; p.ldr = (int)d->ldr;
  p.nb2 = 1; p.b2divB = 1; p.kper = d->K; p.nsplit = 1; p.alpha = 1.0f;
  p.f8_rs = d->x_scale; p.f8_cs = d->w_scale;
  if (d->mode == 1) {
    VQ3_CHECK_ARG(d->C && d->N % 256 == 0 && d->ldc >= d->N / 2 && d->ldc % 8 == 0 && !d->residual, "gemm_fp8_ex: SwiGLU forward needs N = 2 I, I %% 128 == 0");
    VQ3_CHECK_ARG(((uintptr_t)d->gu | (uintptr_t)d->C) % 16 == 0, "gemm_fp8_ex: gate|up / act must be 16-byte aligned");
    p.epi = 3; p.sw_dgu = (bf16_t*)d->gu;
  } else if (d->mode == 2) {
    VQ3_CHECK_ARG(d->gu && d->dgu && d->N % 8 == 0 && !d->residual, "gemm_fp8_ex: SwiGLU backward needs the saved gate|up and an output");
    VQ3_CHECK_ARG(((uintptr_t)d->gu | (uintptr_t)d->dgu) % 16 == 0, "gemm_fp8_ex: gu / dgu must be 16-byte aligned");
    p.epi = 2; p.sw_gu = (const bf16_t*)d->gu; p.sw_dgu = (bf16_t*)d->dgu;
    p.C = d->dgu; p.ldc = d->N;            // placeholders (alignment checks only)
  }
  p.vec_ok = (p.ldc % 4 == 0 && (uintptr_t)p.C % 8 == 0 && (!p.R || (p.ldr % 4 == 0 && (uintptr_t)p.R % 8 == 0))) ? 1 : 0;
  const int rc = launch_gemm_v6_f8(p, (hipStream_t)stream);
  if (rc > 0) return rc;
  if (rc == 0) {
    VQ3_CHECK_LAUNCH("gemm_fp8_ex(v6)");
    return 0;
  }
  VQ3_CHECK_ARG(d->mode == 0, "gemm_fp8_ex: the fused SwiGLU epilogues need the 256 x 256 kernel's contract (16-byte aligned rows)");
  // plain product without column scales on the loader-ring kernel: a vector of ones stands in
  vq3_set_error("gemm_fp8_ex: mode 0 without w_scale needs 16-byte aligned bf16 rows");
  return 1;
}
