// GEMM v7 for gfx950: 256x128x64 tile, FOUR waves (2 x 2, wave tile 128 x 64 = the 8-phase kernel's), 80 KiB of LDS and at most 256
// registers per lane, so that TWO workgroups share a CU - one wave of each on every SIMD.
//
// Why (round 4; tools/gemm_stamps.py on the 256 x 256 kernel at K = 1024, the VGGT blocks): a tile is 24-27 us of main loop wrapped in
// 2 us of dispatch gap, 2.4 us of prologue and 8-16 us of epilogue (C image, GELU / LayerNorm + RoPE, row stores) during which the
// CU's MFMA pipes idle - MFMA-busy 0.34-0.35 - and a one-workgroup-per-CU kernel has nobody to fill them: its eight waves reach the
// epilogue together. Here the co-resident workgroup (its own barriers, its own tile, naturally out of phase) multiplies while this one
// runs its epilogue, is being dispatched or waits for its first operands. Same contract as the other NT kernels
// (C = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, f32 accumulate, K % 64 == 0, LDS-staged bf16 epilogue only).
//
// Main loop: LDS = a ring of FIVE 16 KiB half-tile slots (128 rows x 64 k, the DMA image of gemm6.hip). The half-tiles of the K tiles
// are requested in the order they are read - s = 3 t + {0: A rows 0-127, 1: B, 2: A rows 128-255} into slot s % 5 - and a K tile is
// two phases of 32 MFMAs per wave with ONE barrier each:
//     phase 2t:    vmcnt(8) ; s_barrier ; request B(t+1)            ; read A0(t), B(t) ; 32 MFMA -> rows   0-127
//     phase 2t+1:  vmcnt(8) ; s_barrier ; request A1(t+1), A0(t+2)  ; read A1(t)       ; 32 MFMA -> rows 128-255 (B fragments kept)
//   RAW: a wave's own four 1-KiB pieces of a half-tile are retired by its counted vmcnt (the two youngest half-tiles stay in flight),
//        every other wave's by the barrier behind it.
//   WAR: slot s % 5 is requested again (s + 5) one phase after its last read, behind that phase's barrier - which a wave reaches only
//        after the MFMAs that consumed its fragments were issued, i.e. after its reads returned.
#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int BK7 = 64;
constexpr int HALF7 = 128 * 128;         // bytes per half-tile (128 rows x 64 k)
constexpr int NSLOT7 = 5;

#define V7_BARRIER()                    \
  do {                                  \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

// EK as in gemm6.hip: 0 plain C epilogue, 3 the same behind a folded LayerNorm, 1 fused q|k|v (LayerNorm fold optional), 2 SwiGLU backward,
// 4 SwiGLU forward (a tile multiplies 64 gate rows and the SAME 64 up rows of the fused weight)
template <int EK>
__global__ __launch_bounds__(256, 2) void gemm_v7_kernel(GemmParams p) {
  constexpr bool HAS_LN = (EK == 1 || EK == 3);
  constexpr int BM = 256, BN = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  int m0, n0;
  tile_coords(p, BM, BN, m0, n0);
  // folded LayerNorm: thread r fetches row r's statistics now (two registers through the main loop; the round trip runs under the prologue)
  float ln_mu = 0.f, ln_rs = 0.f;
  if (HAS_LN && p.ln_in) ln_row(p, m0 + tid, ln_mu, ln_rs);
  const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z % p.nb2;
  const bf16_t* A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* B = p.B + b1 * p.sB1 + (long)(b2 / p.b2divB) * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;
  const long roff = b1 * p.sR1 + b2 * p.sR2;

  // ---- DMA sources. A half-tile is 16 pieces of 8 rows x 128 B; wave w issues pieces w, w + 4, w + 8, w + 12. Lane l of a piece reads
  // k-chunk (l & 7) ^ (l >> 3) of row l >> 3 (the read-side swizzle chunk ^ (row & 7), applied to the source).
  const int prow = lane >> 3;
  const int kch = (lane & 7) ^ prow;
  unsigned offA[2][4], offB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wid + 4 * j) * 8 + prow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ra = m0 + h * 128 + r; ra = ra < p.M ? ra : p.M - 1;
      offA[h][j] = (unsigned)(((long)ra * p.lda + kch * 8) * 2);
    }
    int rb = n0 + r;
    if constexpr (EK == 4) rb = r < BN / 2 ? (n0 >> 1) + r : (p.N >> 1) + (n0 >> 1) + r - BN / 2;
    rb = rb < p.N ? rb : p.N - 1;
    offB[j] = (unsigned)(((long)rb * p.ldb + kch * 8) * 2);
  }
  int ws = 0;                             // ring slot of the next request (wave-uniform)
  auto request = [&](const bf16_t* base, const unsigned (&off)[4], int t) {
    const char* g = reinterpret_cast<const char*>(base) + (long)t * (BK7 * 2);
    char* slot = smem + ws * HALF7;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off[j]),
                                       (__attribute__((address_space(3))) void*)(slot + (wid + 4 * j) * 1024), 16, 0, 0);
    ws = ws == NSLOT7 - 1 ? 0 : ws + 1;
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int a_base = (wr * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
  const int b_base = (wc * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
  bf16x8 xa[4][2], wb[4][2];
  int rsl = 0;                            // ring slot of the next read
  auto read_a = [&]() {
    const char* slot = smem + rsl * HALF7;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xa[mt][kh] = *reinterpret_cast<const bf16x8*>(slot + ((a_base ^ (kh << 6)) + mt * 2048));
    rsl = rsl == NSLOT7 - 1 ? 0 : rsl + 1;
  };
  auto read_b = [&]() {
    const char* slot = smem + rsl * HALF7;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int nt2 = 0; nt2 < 4; ++nt2) wb[nt2][kh] = *reinterpret_cast<const bf16x8*>(slot + ((b_base ^ (kh << 6)) + nt2 * 2048));
    rsl = rsl == NSLOT7 - 1 ? 0 : rsl + 1;
  };
#define V7_MMA(I)                                                                                                \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                             \
      _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
        _Pragma("unroll") for (int nt2 = 0; nt2 < 4; ++nt2)                                                      \
          acc[(I) * 4 + mt][nt2] =                                                                               \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nt2][kh], xa[mt][kh], acc[(I) * 4 + mt][nt2], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)

  const int nt = p.K / BK7;
  // ---- prologue: the first four half-tiles
  request(A, offA[0], 0);
  request(B, offB, 0);
  request(A, offA[1], 0);
  if (nt > 1) request(A, offA[0], 1);
  for (int t = 0; t < nt; ++t) {
    const bool last = t + 1 >= nt;
    // phase 2t: A rows 0-127 x B
    if (last) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    V7_BARRIER();
    if (!last) request(B, offB, t + 1);
    read_a();
    read_b();
    V7_MMA(0);
    // phase 2t+1: A rows 128-255 x the same B fragments
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    V7_BARRIER();
    if (!last) request(A, offA[1], t + 1);
    if (t + 2 < nt) request(A, offA[0], t + 2);
    read_a();
    V7_MMA(1);
  }

  // ---- epilogue: through LDS (the ring is dead once every wave passed the barrier below), out as whole rows
  const bool ln_on = HAS_LN && p.ln_in;
  float2* lnp = reinterpret_cast<float2*>(smem + BM * BN * 2);      // (mu, rstd) pairs above the 64 KiB C image
  f32x4 cc[4];
  f32x4 bias_r[4], cs_r[4];
#pragma unroll
  for (int nt2 = 0; nt2 < 4; ++nt2) {
    int n = n0 + wc * 64 + nt2 * 16 + 4 * fq;
    n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);                 // columns past N are never stored
    if (p.bias) bias_r[nt2] = *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.colscale) cs_r[nt2] = *reinterpret_cast<const f32x4*>(p.colscale + n);
    if (ln_on) cc[nt2] = ln_colsum(p, n);
  }
  EpiPre pre;
  pre.on = false;
  if constexpr (EK == 0) staged_prefetch<BM, BN>(p, coff, roff, m0, n0, tid, 256, pre);
  V7_BARRIER();
  if (ln_on) {
    lnp[tid] = float2{ln_mu, ln_rs};
    // The raw s_barrier does NOT wait for this wave's outstanding LDS write (V7_BARRIER carries no s_waitcnt: in the main loop the LDS is
    // written by DMA only, retired by the counted vmcnt): without the wait below a wave could pass the barrier with its pair still in
    // flight and another wave's first read of lnp[] - row group (i = 0, mt = 0) - returned what the ring's slot 4 held before: ONE
    // wave's 16 rows x 64 columns of one tile per launch or so came out normalised with a stale (mu, rstd) (round 5: found through the
    // full-depth parity test, tools/diag/g7_pattern.py; tests/test_kernels_gpu.py::test_gemm_v7_ln_fold_is_repeatable).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    V7_BARRIER();
  }
  auto stage_all = [&](auto act_tag, auto mode_tag, auto ln_tag) {
    constexpr int ACT = decltype(act_tag)::value, MODE = decltype(mode_tag)::value;
    constexpr bool LN = HAS_LN && decltype(ln_tag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = i * 128 + wr * 64 + mt * 16 + fr;
        float2 lv = float2{0.f, 1.f};
        if (LN) lv = lnp[row];
#pragma unroll
        for (int nt2 = 0; nt2 < 4; ++nt2) {
          f32x4 a = acc[i * 4 + mt][nt2];
          if (LN) a = ln_apply(a, lv.x, lv.y, cc[nt2]);
          stage_quad<BN, ACT, MODE>(p, smem, row, wc * 64 + nt2 * 16 + 4 * fq, a, bias_r[nt2], cs_r[nt2]);
        }
      }
  };
#define V7_STAGE(ACT_, MODE_)                                                                                                     \
  do {                                                                                                                            \
    if constexpr (EK == 3) stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::true_type{}); \
    else if constexpr (EK == 1) {                                                                                                 \
      if (ln_on) stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::true_type{});          \
      else stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::false_type{});               \
    } else stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::false_type{});               \
  } while (0)
  if constexpr (EK == 2 || EK == 4) V7_STAGE(0, 0);              // (host: no bias, no LayerScale, alpha == 1)
  else if constexpr (EK == 1) {
    if (p.bias && p.alpha == 1.f) V7_STAGE(0, 1);
    else V7_STAGE(0, -1);
  } else VQ3_STAGE_DISPATCH(p, V7_STAGE);
#undef V7_STAGE
  __syncthreads();
  staged_store<BM, BN, (EK == 3 ? 0 : EK)>(p, smem, coff, roff, m0, n0, tid, 256, &pre);
}

constexpr int SMEM7 = NSLOT7 * HALF7;      // 80 KiB: two workgroups per CU

template <int EK>
int launch_ek(const GemmParams& p, dim3 grid, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_v7_kernel<EK>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM7);
    if (e != hipSuccess) {
      vq3_set_error("gemm v7: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_v7_kernel<EK>), grid, dim3(256), SMEM7, stream, p);
  return 0;
}

}  // namespace

// Returns -1 (nothing launched) when the problem is outside the kernel's contract (f32 output, rows that cannot take the staged epilogue):
// the caller picks another kernel.
int launch_gemm_v7(GemmParams& p, int nbatch, hipStream_t stream) {
  static_assert(256 * 128 * 2 + 256 * 8 <= SMEM7, "C image + LayerNorm pairs fit the ring");
  if (p.out_f32 || !host_staged_ok(p) || p.K % BK7 != 0) return -1;
  if (p.epi == 3 && (p.N >> 1) % 64 != 0) return -1;
  if ((p.epi == 2 || p.epi == 3) && p.ln_in) return -1;
  p.mtiles = (p.M + 255) / 256;
  p.ntiles = (p.N + 127) / 128;
  choose_tile_order(p, 256, 128, 2);
  const dim3 grid(p.mtiles * p.ntiles, 1, nbatch);
  if (p.epi == 1) return launch_ek<1>(p, grid, stream);
  if (p.epi == 2) return launch_ek<2>(p, grid, stream);
  if (p.epi == 3) return launch_ek<4>(p, grid, stream);
  if (p.ln_in) return launch_ek<3>(p, grid, stream);
  return launch_ek<0>(p, grid, stream);
}

}  // namespace vq3gemm
