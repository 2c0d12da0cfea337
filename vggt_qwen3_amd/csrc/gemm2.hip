// GEMM v2 for gfx950: LDS-DMA (global_load_lds_dwordx4) staged, 3-stage ring, counted vmcnt + raw s_barrier.
//
// Same contract as gemm.hip's kernel (C = epilogue(alpha * A[M,K] . B[N,K]^T), NT, bf16 in / f32 MFMA accumulate), but
// the operands never pass through VGPRs: every wave issues 1-KiB DMA pieces (8 tile rows x 128 B) straight into the
// XOR-swizzled LDS image. The DMA destination is lane-linear (wave-uniform base + lane*16), so the swizzle
// chunk' = chunk ^ (row & 7) is applied to each lane's SOURCE address (lane l reads k-chunk (l&7)^(l>>3) of row l>>3)
// and again on the fragment reads - the same involution on both sides.
// Pipeline, one barrier per 64-deep K step, tile t+1 always in flight across the barrier:
//     s_waitcnt vmcnt(P)   (all but this wave's newest P pieces landed -> its share of tile t is in LDS)
//     s_barrier            (everyone's share of tile t landed; everyone finished reading the stage tile t+2 reuses)
//     issue tile t+2 -> stage (t+2)%3
//     ds_read fragments of stage t%3, 16x16x32 bf16 MFMAs
// One workgroup per CU (up to 144 KiB LDS), 8 waves for the 256x128 / 128x256 tiles, 4 for 128x128.
#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN, int NSTAGE, int NLOAD, bool OUT_F32>
__global__ __launch_bounds__(64 * (WM * WN + NLOAD), (NSTAGE == 2 ? 2 : 1) * ((WM * WN + NLOAD + 3) / 4)) void gemm_v2_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;   // 16x16 MFMA tiles per wave
  constexpr int STAGE = (BM + BN) * 128;                 // bytes per stage
  constexpr int NPIECE = (BM + BN) / 8;                  // 1-KiB pieces per stage
  // NLOAD > 0: dedicated DMA waves (wid >= NW) issue every piece; the compute waves never leave the MFMA/ds_read stream
  constexpr int NISS = NLOAD > 0 ? NLOAD : NW;            // waves that issue DMA
  constexpr int PPW = NPIECE / NISS;                     // pieces per issuing wave per stage
  static_assert(NPIECE % NISS == 0, "pieces must divide evenly over the issuing waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = NLOAD > 0 && wid >= NW;
  const int iw = NLOAD > 0 ? (wid >= NW ? wid - NW : 0) : wid;   // index among the issuing waves
  const int wm = (wid % NW) / WN, wn = wid % WN;
  int m0, n0;
  tile_coords(p, BM, BN, m0, n0);
  // folded LayerNorm: thread r < BM fetches row r's statistics now (two registers through the main loop) - at the epilogue the
  // round trip to L2 / HBM would be exposed on every tile
  static_assert(BM <= 64 * WM * WN, "one statistics thread per tile row");
  float ln_mu = 0.f, ln_rs = 0.f;
  if (p.ln_in && tid < BM) ln_row(p, m0 + tid, ln_mu, ln_rs);
  const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z % p.nb2;
  const bf16_t* A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* B = p.B + b1 * p.sB1 + (long)(b2 / p.b2divB) * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;
  const long roff = b1 * p.sR1 + b2 * p.sR2;

  // ---- DMA source pointers: piece pi = wid + NW*j covers tile rows 8*pi .. 8*pi+7 of [A rows | B rows]
  const int prow = lane >> 3;                       // row inside the piece (== row & 7 of the tile row)
  const int kch = (lane & 7) ^ prow;                // pre-swizzled k-chunk
  const bf16_t* gsrc[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int pi = iw + NISS * j;
    const int trow = pi * 8 + prow;                 // row in the stacked [A|B] tile
    if (trow < BM) {
      int r = m0 + trow; r = r < p.M ? r : p.M - 1;
      gsrc[j] = A + (long)r * p.lda + kch * 8;
    } else {
      int r = n0 + (trow - BM); r = r < p.N ? r : p.N - 1;
      gsrc[j] = B + (long)r * p.ldb + kch * 8;
    }
  }
  auto issue = [&](int tile, int stage) {
    char* sb = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int pi = iw + NISS * j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (long)tile * BK),
                                       (__attribute__((address_space(3))) void*)(sb + pi * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * (BM / WM) + i * 16 + fr;
    a_off[i] = row * 128 + ((fq ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * (BN / WN) + j * 16 + fr;
    b_off[j] = BM * 128 + row * 128 + ((fq ^ (row & 7)) << 4);
  }

  const int nt = p.K / BK;
  const int last = nt - 1;
  auto compute = [&](int stage) {
    const char* sb = smem + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xa[TM], wb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(sb + (a_off[i] ^ (ks << 6)));
#pragma unroll
      for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(sb + (b_off[j] ^ (ks << 6)));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
  };
  // Epilogue operands (bias / LayerScale column vectors, residual quads) are fetched before the LAST K step's MFMAs so
  // their L2 round trip is hidden; only for the common case (bf16 output, 4-wide aligned rows).
  constexpr bool PRE_RES = TM * TN <= 8;      // the 256-row tiles have no registers to spare for the residual quads
  f32x4 bias_r[TN], cs_r[TN];
  u32x2 res_r[PRE_RES ? TM : 1][TN];
  const bool staged = !OUT_F32 && staged_ok(p, coff, roff);     // LDS-staged epilogue (whole-row stores): gemm_common.h
  const bool pre_ok = !staged && !OUT_F32 && p.vec_ok && (p.bias || p.colscale || p.R);
  auto prefetch_epilogue = [&]() {
    if (!pre_ok) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
      n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);     // clamped lanes fall back to the generic path at store time
      if (p.bias) bias_r[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.colscale) cs_r[j] = *reinterpret_cast<const f32x4*>(p.colscale + n);
      if (PRE_RES && p.R) {
#pragma unroll
        for (int i = 0; i < (PRE_RES ? TM : 1); ++i) {
          int m = m0 + wm * (BM / WM) + i * 16 + fr;
          m = m < p.M ? m : p.M - 1;
          res_r[i][j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.R) + roff + (long)m * p.ldr + n);
        }
      }
    }
  };
  if (NSTAGE >= 3 && NLOAD > 0) {
    // producer / consumer split inside the workgroup (one s_barrier per K step for everybody):
    //   loader waves:  wait until their pieces of tile t landed -> barrier -> issue tile t+2
    //   compute waves: barrier -> fragments of tile t -> MFMAs
    if (loader) {
      // the ring runs NSTAGE-1 tiles ahead: cold (HBM-resident) weights need more than two tiles of latency cover
#pragma unroll
      for (int i = 0; i < NSTAGE - 1; ++i)
        if (i < nt) issue(i, i);
      int stage = 0;
      for (int t = 0; t < nt; ++t) {
        const int newer = (last - t) < (NSTAGE - 2) ? (last - t) : (NSTAGE - 2);   // tiles allowed to stay in flight
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
    int stage = 0;
    for (int t = 0; t < last; ++t) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      compute(stage);
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    prefetch_epilogue();
    compute(stage);
  } else if (NSTAGE == 3) {
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    int stage = 0;
    for (int t = 0; t < last; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 <= last) {
        int s2 = stage + 2; s2 = s2 >= 3 ? s2 - 3 : s2;
        issue(t + 2, s2);
      }
      compute(stage);
      stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    prefetch_epilogue();
    compute(stage);
  } else {
    // two stages, two workgroups per CU: tile t+1 streams in while tile t is multiplied; the co-resident workgroup
    // (its own barrier, naturally out of phase) fills the MFMA pipe while this one waits.
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < last; ++t) {
      issue(t + 1, (t + 1) & 1);
      compute(t & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    prefetch_epilogue();
    compute(last & 1);
  }

  if (!OUT_F32 && staged) {
    // loader waves have returned; the compute waves meet once (all fragment reads of the last stage are done), lay the
    // finished tile out in LDS and store whole rows
    f32x4 bias_v[TN], cs_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
      n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);
      if (p.bias) bias_v[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.colscale) cs_v[j] = *reinterpret_cast<const f32x4*>(p.colscale + n);
    }
    __syncthreads();
    if (p.ln_in) {      // folded LayerNorm: rstd (acc - mu c) per lane-owned row, before the usual epilogue
      f32x4 cc[TN];
      float mu[TM], rs[TM];
#pragma unroll
      for (int j = 0; j < TN; ++j) cc[j] = ln_colsum(p, n0 + wn * (BN / WN) + j * 16 + 4 * fq);
      if (tid < BM) reinterpret_cast<float2*>(smem)[tid] = float2{ln_mu, ln_rs};   // (barrier above: the last stage's reads are done)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float2 v = reinterpret_cast<const float2*>(smem)[wm * (BM / WM) + i * 16 + fr];
        mu[i] = v.x; rs[i] = v.y;
      }
      __syncthreads();                                       // before the C image overwrites the pairs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = ln_apply(acc[i][j], mu[i], rs[i], cc[j]);
    }
#define V2_STAGE(ACT_, MODE_)                                                                                                       \
  do {                                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                                  \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                                \
        stage_quad<BN, ACT_, MODE_>(p, smem, wm * (BM / WM) + i * 16 + fr, wn * (BN / WN) + j * 16 + 4 * fq, acc[i][j], bias_v[j], cs_v[j]); \
  } while (0)
    VQ3_STAGE_DISPATCH(p, V2_STAGE);
#undef V2_STAGE
    __syncthreads();
    staged_store<BM, BN>(p, smem, coff, roff, m0, n0, tid, 64 * NW);
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WM) + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
      if (n >= p.N) continue;
      if (pre_ok && n + 3 < p.N) {
        u32x2 rv = res_r[PRE_RES ? i : 0][j];
        if (!PRE_RES && p.R)
          rv = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.R) + roff + (long)m * p.ldr + n);
        store_quad_pre(p, coff, m, n, acc[i][j], bias_r[j], cs_r[j], rv);
      }
      else store_quad<OUT_F32>(p, coff, roff, m, n, acc[i][j]);
    }
  }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int NLOAD = 0>
int launch_cfg(GemmParams& p, int nbatch, hipStream_t stream) {
  constexpr int SMEM = NSTAGE * (BM + BN) * 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_v2_kernel<BM, BN, WM, WN, NSTAGE, NLOAD, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_v2_kernel<BM, BN, WM, WN, NSTAGE, NLOAD, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vq3_set_error("gemm v2: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 2;
    }
    attr_done = true;
  }
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  choose_tile_order(p, BM, BN, NSTAGE == 2 ? 2 : 1);
  dim3 grid(p.mtiles * p.ntiles, 1, nbatch);
  if (p.out_f32)
    hipLaunchKernelGGL((gemm_v2_kernel<BM, BN, WM, WN, NSTAGE, NLOAD, true>), grid, dim3(64 * (WM * WN + NLOAD)), SMEM, stream, p);
  else
    hipLaunchKernelGGL((gemm_v2_kernel<BM, BN, WM, WN, NSTAGE, NLOAD, false>), grid, dim3(64 * (WM * WN + NLOAD)), SMEM, stream, p);
  return 0;
}

}  // namespace

int launch_gemm_v2(GemmParams& p, int cfg, int nbatch, hipStream_t stream) {
  switch (cfg) {
    case 0: return launch_cfg<256, 128, 4, 2, 3>(p, nbatch, stream);
    case 1: return launch_cfg<128, 256, 2, 4, 3>(p, nbatch, stream);
    case 2: return launch_cfg<128, 128, 2, 2, 3>(p, nbatch, stream);
    case 3: return launch_cfg<128, 128, 4, 2, 3>(p, nbatch, stream);
    case 4: return launch_cfg<128, 64, 4, 2, 3>(p, nbatch, stream);
    case 5: return launch_cfg<64, 128, 2, 4, 3>(p, nbatch, stream);
    case 6: return launch_cfg<128, 128, 2, 2, 2>(p, nbatch, stream);
    case 7: return launch_cfg<128, 128, 4, 2, 2>(p, nbatch, stream);
    case 8: return launch_cfg<256, 128, 4, 2, 3, 4>(p, nbatch, stream);
    case 9: return launch_cfg<128, 128, 4, 2, 3, 4>(p, nbatch, stream);
    case 10: return launch_cfg<128, 128, 2, 2, 3, 4>(p, nbatch, stream);
    case 11: return launch_cfg<256, 128, 4, 2, 3, 2>(p, nbatch, stream);
    case 12: return launch_cfg<128, 128, 4, 2, 3, 2>(p, nbatch, stream);
    case 13: return launch_cfg<128, 128, 4, 2, 4, 2>(p, nbatch, stream);
    // 224 x 128 / 192 x 128 (7 x 2 / 6 x 2 MFMA tiles per wave): M = 6174 = 27.6 x 224 gives 224 tiles for N = 1024 (0.875 fill in
    // one round instead of 0.78) - measured: ties 256 x 128 (the 7 x 2 wave tile moves 0.64 KB of LDS per MFMA, and a single
    // round lasts as long as its slowest tile either way), so the tuner does not list them; kept for vq3_gemm_force_config
    case 15: return launch_cfg<224, 128, 2, 4, 3, 2>(p, nbatch, stream);
    case 16: return launch_cfg<192, 128, 2, 4, 3, 2>(p, nbatch, stream);
    // 192 x 128, FOUR waves (96 x 64 each: 0.42 KB of LDS per MFMA), 2 stages = 80 KiB: TWO workgroups per CU, each one wave per SIMD.
    // At K = 1024 a tile's epilogue (staging C through LDS + the row stores: 8-9 us of a 24 us tile on the 8-wave kernels, during which
    // the MFMA pipe idles - tools/gemm_stamps.py) runs under the co-resident workgroup's main loop; M = 49 392 = 257.25 x 192.
    case 17: return launch_cfg<192, 128, 2, 2, 2>(p, nbatch, stream);
    default: return launch_cfg<128, 128, 4, 2, 5, 2>(p, nbatch, stream);
  }
}

}  // namespace vq3gemm
