// GEMM v3 for gfx950: the v2 LDS-DMA pipeline (gemm2.hip) generalised to every operand layout, so that no dgrad,
// wgrad or attention product needs a transposed copy in HBM:
//     C[M,N] = epilogue(alpha * sum_k opA(A)[m,k] * opB(B)[n,k])
//     transA = 0: A[m*lda + k]  (contraction contiguous)      transA = 1: A[k*lda + m]  ("k-major")
//     transB = 0: B[n*ldb + k]                                  transB = 1: B[k*ldb + n]
// and K only has to be a multiple of 8 (the tail of the last 64-deep tile is zero-filled in LDS on the A side and
// read from clamped, finite addresses on the B side).
//
// A k-major 64(k) x 128 tile is DMA'd as 16 pieces of 4 k-rows x 256 B. MFMA fragments need 8 consecutive k for one
// m (a COLUMN of that image): they come from ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a
// 4-row x 16-column block - two of them per fragment, no transposed copy anywhere. Bank conflicts: a 32-lane half
// reads 2 groups x 4 rows at one column offset; the 32-byte segment index is XORed with (k&3) | ((k>>3)&1)<<2 (again
// on the DMA SOURCE address, the LDS image being lane-linear), which spreads those 8 (row, group) pairs over all
// eight 32-byte segments of the 256-byte bank row: conflict-free.
// Tile 128x128x64, 8 compute waves (4x2, 32x64 per wave); either 2 stages x 2 workgroups/CU, or a 4-stage ring fed
// by 2 dedicated DMA-loader waves x 1 workgroup/CU.
#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int BK = 64, BN = 128;
constexpr int OPB = 128 * BK * 2;  // bytes per 128-row (or 128-column) operand sub-tile (16 KiB)
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ds_read_b64_tr_b16 through inline asm, NOT through __builtin_amdgcn_ds_read_tr16_b64: with the builtin hipcc
// (ROCm 7.2) orders the read behind every in-flight LDS-DMA and emits s_waitcnt vmcnt(0) in front of it in each K step
// (it does not do that for plain ds_read_b128), which drains the prefetch pipeline (measured: -35 % on the dgrad
// shapes). The asm read is invisible to the waitcnt pass, so its completion is waited for by hand: one
// s_waitcnt lgkmcnt(0) + sched_barrier(0) in front of the MFMAs that consume the fragments (LDS returns in order, so
// mixing with compiler-counted ds_read_b128 is only ever conservative).
template <int OFF>
__device__ __forceinline__ u32x2 ds_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ void tr_frag(unsigned addr, u32x2& lo, u32x2& hi) {
  lo = ds_tr<OFF>(addr);
  hi = ds_tr<OFF + 4 * 256>(addr);
}
__device__ __forceinline__ bf16x8 join(const u32x2& lo, const u32x2& hi) {
  const u32x4 t = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, t);
}

// BM = 128 or 256: the A tile is BM / 128 sub-tiles of 128 rows, each with the 64 x 128 image described above (a 256-row tile
// at 64 x 64 per wave moves 0.5 KB of LDS per MFMA instead of 1.0: the k-major counterpart of gemm2's 256 x 128 loader kernel).
template <int BM, int WM, int WN, int NSTAGE, int NLOAD, bool AKM, bool BKM, bool OUT_F32>
__global__ __launch_bounds__(64 * (WM * WN + NLOAD), (NSTAGE == 2 ? 2 : 1) * ((WM * WN + NLOAD + 3) / 4)) void gemm_v3_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int SA = BM / 128;                        // A sub-tiles
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int STAGE = (SA + 1) * OPB;
  // NLOAD > 0: dedicated DMA waves (wid >= NW) issue every piece (see gemm2.hip)
  constexpr int NISS = NLOAD > 0 ? NLOAD : NW;
  constexpr int NPIECE = 16 * (SA + 1);               // 16 pieces per sub-tile: A sub-tiles first, then B
  constexpr int PPW = NPIECE / NISS;
  static_assert(16 % NISS == 0, "the 16 pieces of a sub-tile must divide evenly over the issuing waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = NLOAD > 0 && wid >= NW;
  const int iw = NLOAD > 0 ? (wid >= NW ? wid - NW : 0) : wid;
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
  // split-K: this workgroup contracts k in [kbeg, kbeg + K)
  const int kbeg = blockIdx.y * p.kper;
  const int K = (p.K - kbeg) < p.kper ? (p.K - kbeg) : p.kper;
  A += AKM ? (long)kbeg * p.lda : (long)kbeg;
  B += BKM ? (long)kbeg * p.ldb : (long)kbeg;
  const int nt = (K + BK - 1) / BK, last = nt - 1;
  const int krem = K & (BK - 1);  // 0 or a multiple of 8: valid depth of the last tile

  // ---- DMA sources. piece index within an operand: po = (wid + NW*j) & 15; j < PPW/2 -> A, else B.
  // tile_ptr(op, piece, tile): per-lane source address of that piece.
  auto src_ptr = [&](int sub, int po, int tile) -> const bf16_t* {
    const bool isA = sub < SA;
    const bf16_t* base = isA ? A : B;
    const long ld = isA ? p.lda : p.ldb;
    const int ext = isA ? p.M : p.N, x0 = isA ? m0 + 128 * sub : n0;
    const bool km = isA ? AKM : BKM;
    if (!km) {
      const int prow = lane >> 3;
      int r = x0 + po * 8 + prow; r = r < ext ? r : ext - 1;
      int k = tile * BK + (((lane & 7) ^ prow) << 3);
      k = k <= K - 8 ? k : K - 8;                       // tail: clamp to the last valid 16-byte chunk (finite data)
      return base + (long)r * ld + k;
    } else {
      const int krow = po * 4 + (lane >> 4);            // 0..63
      const int pc = lane & 15;
      const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int lchunk = (((pc >> 1) ^ f) << 1) | (pc & 1);
      int col = x0 + lchunk * 8; col = col <= ext - 8 ? col : ext - 8;
      int k = tile * BK + krow; k = k < K ? k : K - 1;  // tail rows: re-read the last valid row
      return base + (long)k * ld + col;
    }
  };
  auto issue = [&](int tile, int stage) {
    char* sb = smem + stage * STAGE;
#pragma unroll
    for (int sub = 0; sub <= SA; ++sub)               // compile-time sub-tile: operand, layout and leading dimension fold
#pragma unroll
      for (int jj = 0; jj < 16 / NISS; ++jj) {
        const int po = iw + NISS * jj;                // this wave's pieces of the sub-tile
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_ptr(sub, po, tile),
                                         (__attribute__((address_space(3))) void*)(sb + sub * OPB + po * 1024), 16, 0, 0);
      }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TM], b_off[TN];
  {
    // k-major: addr = (32ks + 4r + 8q + (li>>2)) * 256 + ((seg ^ fl) << 5) + lane_part
    const int li = fr;
    const int fl = (li >> 2) | ((fq & 1) << 2);
    const int lane_part = (((li >> 1) & 1) << 4) | ((li & 1) << 3);
    const int krow_base = 8 * fq + (li >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int trow = wm * (BM / WM) + i * 16;
      const int sub = trow >> 7, row = trow & 127;     // 128-row sub-tile and the row inside it
      if (AKM) a_off[i] = sub * OPB + krow_base * 256 + ((((row >> 4)) ^ fl) << 5) + lane_part;
      else { const int rr = row + fr; a_off[i] = sub * OPB + rr * 128 + ((fq ^ (rr & 7)) << 4); }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * (BN / WN) + j * 16;
      if (BKM) b_off[j] = SA * OPB + krow_base * 256 + ((((row >> 4)) ^ fl) << 5) + lane_part;
      else { const int rr = row + fr; b_off[j] = SA * OPB + rr * 128 + ((fq ^ (rr & 7)) << 4); }
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto compute = [&](int stage) {
    const char* sb = smem + stage * STAGE;
    const unsigned sa = lds0 + stage * STAGE;
    // all fragment reads of both k-steps first (k-major operands by asm, the others by the compiler), one wait, MFMAs
    u32x2 alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
    bf16x8 xa[2][TM], wb[2][TN];
    if (AKM) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        tr_frag<0>(sa + a_off[i], alo[0][i], ahi[0][i]);
        tr_frag<32 * 256>(sa + a_off[i], alo[1][i], ahi[1][i]);
      }
    }
    if (BKM) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        tr_frag<0>(sa + b_off[j], blo[0][j], bhi[0][j]);
        tr_frag<32 * 256>(sa + b_off[j], blo[1][j], bhi[1][j]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (!AKM) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[ks][i] = *reinterpret_cast<const bf16x8*>(sb + (a_off[i] ^ (ks << 6)));
      }
      if (!BKM) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[ks][j] = *reinterpret_cast<const bf16x8*>(sb + (b_off[j] ^ (ks << 6)));
      }
    }
    if (AKM || BKM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // The transposed reads' destinations carry no data dependence on the wait ("memory" orders memory operations only, and the
      // MFMAs below are register-only): every destination passes through an empty volatile statement BEHIND the wait (volatile
      // statements keep their order), so no use of it can be scheduled above the wait (cdna guide 5.7 item 1, form ii / item 3).
      if (AKM) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(alo[ks][i]), "+v"(ahi[ks][i]));
      }
      if (BKM) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(blo[ks][j]), "+v"(bhi[ks][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (AKM) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[ks][i] = join(alo[ks][i], ahi[ks][i]);
      }
      if (BKM) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[ks][j] = join(blo[ks][j], bhi[ks][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks][j], xa[ks][i], acc[i][j], 0, 0, 0);
    }
  };
  // zero the k >= krem part of the A tile of the LAST stage (B's tail holds finite duplicates: 0 * finite = 0)
  auto zero_tail = [&](int stage) {
    char* sa = smem + stage * STAGE;
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (AKM) {
      const int n16 = (BK - krem) * 16;
      for (int sub = 0; sub < SA; ++sub)
        for (int idx = tid; idx < n16; idx += 64 * NW) *reinterpret_cast<u32x4*>(sa + sub * OPB + krem * 256 + idx * 16) = z;
    } else {
      const int c0 = krem >> 3, nch = 8 - c0;
      for (int idx = tid; idx < BM * nch; idx += 64 * NW) {
        const int trow = idx & (BM - 1), kc = c0 + idx / BM;
        const int sub = trow >> 7, row = trow & 127;
        *reinterpret_cast<u32x4*>(sa + sub * OPB + row * 128 + ((kc ^ (row & 7)) << 4)) = z;
      }
    }
    __syncthreads();
  };

  // Read-modify-write operands of the epilogue (the gradient being accumulated into, a residual) are fetched before the
  // last K tile's MFMAs, so their L2 round trip is not paid between the last MFMA and the first store.
  u32x2 pre_r[TM][TN];     // one operand only (registers): the accumulation target, else the residual
  const bool staged = !OUT_F32 && staged_ok(p, coff, roff);     // LDS-staged epilogue (whole-row stores): gemm_common.h
  const bool pre_ok = !staged && !OUT_F32 && p.vec_ok && p.nsplit == 1 && !p.bias && !p.colscale && (p.accumulate != 0) != (p.R != nullptr);
  auto prefetch_epilogue = [&]() {
    if (!pre_ok) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int m = m0 + wm * (BM / WM) + i * 16 + fr;
      m = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
        n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);
        if (p.accumulate) pre_r[i][j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.C) + coff + (long)m * p.ldc + n);
        else pre_r[i][j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(p.R) + roff + (long)m * p.ldr + n);
      }
    }
  };

  // The last K tile is peeled out of the loop: its zero-fill does LDS stores, and with LDS stores inside the loop
  // hipcc orders every fragment read behind the in-flight LDS-DMA (s_waitcnt vmcnt(0) per K step), which drains the
  // prefetch pipeline. The steady-state loop touches LDS by DMA and ds_read only.
  if (NSTAGE >= 3 && NLOAD > 0) {
    // producer / consumer split (gemm2.hip): loader waves own the DMA, compute waves only ds_read + MFMA.
    if (loader) {
      // the ring runs NSTAGE-1 tiles ahead (cold, HBM-resident weights need more than two tiles of latency cover)
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
      if (krem) {            // take part in the zero-fill barrier of the last tile
        __syncthreads();
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
    if (krem) zero_tail(stage);
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
    if (krem) zero_tail(stage);
    prefetch_epilogue();
    compute(stage);
  } else {
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
    if (krem) zero_tail(last & 1);
    prefetch_epilogue();
    compute(last & 1);
  }

  if (!OUT_F32 && staged) {
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
#define V3_STAGE(ACT_, MODE_)                                                                                                       \
  do {                                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                                  \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                                \
        stage_quad<BN, ACT_, MODE_>(p, smem, wm * (BM / WM) + i * 16 + fr, wn * (BN / WN) + j * 16 + 4 * fq, acc[i][j], bias_v[j], cs_v[j]); \
  } while (0)
    VQ3_STAGE_DISPATCH(p, V3_STAGE);
#undef V3_STAGE
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
      if (pre_ok && n + 3 < p.N) store_quad_pre(p, coff, m, n, acc[i][j], f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, pre_r[i][j], &pre_r[i][j]);
      else store_quad<OUT_F32>(p, coff, roff, m, n, acc[i][j]);
    }
  }
}

template <int BM, int NSTAGE, int NLOAD, bool AKM, bool BKM>
int launch_v3(GemmParams& p, int nbatch, hipStream_t stream) {
  constexpr int SMEM = NSTAGE * (BM / 128 + 1) * OPB;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_v3_kernel<BM, 4, 2, NSTAGE, NLOAD, AKM, BKM, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_v3_kernel<BM, 4, 2, NSTAGE, NLOAD, AKM, BKM, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vq3_set_error("gemm v3: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 2;
    }
    attr_done = true;
  }
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  choose_tile_order(p, BM, BN, NSTAGE == 2 ? 2 : 1);
  dim3 grid(p.mtiles * p.ntiles, p.nsplit, nbatch);
  if (p.out_f32)
    hipLaunchKernelGGL((gemm_v3_kernel<BM, 4, 2, NSTAGE, NLOAD, AKM, BKM, true>), grid, dim3(64 * (8 + NLOAD)), SMEM, stream, p);
  else
    hipLaunchKernelGGL((gemm_v3_kernel<BM, 4, 2, NSTAGE, NLOAD, AKM, BKM, false>), grid, dim3(64 * (8 + NLOAD)), SMEM, stream, p);
  return 0;
}

}  // namespace

// nstage: 2 = 128x128 tile, 2 stages x 2 workgroups per CU; 3 = 128x128, 4-stage ring + 2 loader waves;
//         5 = 256x128 tile, 3-stage ring + 2 loader waves (one workgroup per CU)
int launch_gemm_v3(GemmParams& p, int transA, int transB, int nstage, int nbatch, hipStream_t stream) {
  const int lay = (transA ? 2 : 0) | (transB ? 1 : 0);
  if (nstage == 5) {
    switch (lay) {
      case 0: return launch_v3<256, 3, 2, false, false>(p, nbatch, stream);
      case 1: return launch_v3<256, 3, 2, false, true>(p, nbatch, stream);
      case 2: return launch_v3<256, 3, 2, true, false>(p, nbatch, stream);
      default: return launch_v3<256, 3, 2, true, true>(p, nbatch, stream);
    }
  }
  if (nstage == 3) {
    switch (lay) {
      case 0: return launch_v3<128, 4, 2, false, false>(p, nbatch, stream);
      case 1: return launch_v3<128, 4, 2, false, true>(p, nbatch, stream);
      case 2: return launch_v3<128, 4, 2, true, false>(p, nbatch, stream);
      default: return launch_v3<128, 4, 2, true, true>(p, nbatch, stream);
    }
  }
  switch (lay) {
    case 0: return launch_v3<128, 2, 0, false, false>(p, nbatch, stream);
    case 1: return launch_v3<128, 2, 0, false, true>(p, nbatch, stream);
    case 2: return launch_v3<128, 2, 0, true, false>(p, nbatch, stream);
    default: return launch_v3<128, 2, 0, true, true>(p, nbatch, stream);
  }
}

}  // namespace vq3gemm
