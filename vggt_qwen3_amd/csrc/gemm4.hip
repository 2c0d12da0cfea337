// GEMM v4 for gfx950: PERSISTENT producer/consumer kernel. Same contract and operand layouts as v3 (gemm3.hip:
// C = epilogue(alpha * sum_k opA(A)[m,k] opB(B)[n,k]), k-major operands via ds_read_b64_tr_b16, K % 8 == 0), same
// 128x128x64 tile, 8 compute waves (4x2, 32x64 each) + 2 DMA-loader waves and a 4-stage LDS ring - but one workgroup
// per CU walks MANY output tiles (tile ids b, b + gridDim.x, ...) and the ring never drains between them:
//   * the loader waves run up to 3 K-steps ahead ACROSS tile boundaries, so the next tile's first operands stream in
//     while the compute waves are still in the current tile's epilogue - no pipeline-fill bubble per tile (it was
//     ~20 % of a K = 1024 or K = 1200 tile: the VGGT and wgrad shapes);
//   * no second "round" of workgroups: the tail of the launch is one partly filled pass instead of a fresh launch wave.
// A "unit" is one (tile, K-step); both roles walk the same unit sequence and meet at exactly one s_barrier per unit
// (+ one more at a tile's last unit when K has a tail that is zero-filled in LDS), so the barrier counts match by
// construction and every loop is bounded.
#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int BK = 64, BM = 128, BN = 128;
constexpr int OPB = BM * BK * 2;  // bytes per operand tile (16 KiB)
constexpr int NSTAGE = 4, NW = 8, NLOAD = 2, WN = 2;
constexpr int STAGE = 2 * OPB;
constexpr int PPW = 32 / NLOAD;   // DMA pieces per loader wave per unit
constexpr int TM = 2, TN = 4;

template <int OFF>
__device__ __forceinline__ u32x2 ds_tr4(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ bf16x8 join4(const u32x2& lo, const u32x2& hi) {
  const u32x4 t = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, t);
}

// tile id -> (batch index, m0, n0): XCD-aware order inside a batch (gemm_common.h: tile_coords)
__device__ __forceinline__ void tile_of(const GemmParams& p, int T, int& bz, int& m0, int& n0) {
  const int per = p.mtiles * p.ntiles;
  bz = T / per;
  tile_coords_id(p, T - bz * per, BM, BN, m0, n0);
}

template <bool AKM, bool BKM, bool OUT_F32>
__global__ __launch_bounds__(64 * (NW + NLOAD), 3) void gemm_v4_kernel(GemmParams p, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= NW;
  const int K = p.K;
  const int nt = (K + BK - 1) / BK;
  const int krem = K & (BK - 1);
  const int first = blockIdx.x, stride = gridDim.x;
  const int ntile_my = (total_tiles - first + stride - 1) / stride;
  const int units = ntile_my * nt;

  if (loader) {
    const int iw = wid - NW;
    // issue-side tile state
    int ti = 0, ki = 0;             // tile ordinal of this workgroup, k-step inside it
    int bz, m0, n0;
    tile_of(p, first, bz, m0, n0);
    const bf16_t* A = p.A + (bz / p.nb2) * p.sA1 + (bz % p.nb2) * p.sA2;
    const bf16_t* B = p.B + (bz / p.nb2) * p.sB1 + (long)((bz % p.nb2) / p.b2divB) * p.sB2;
    auto src_ptr = [&](bool isA, int po, int tile) -> const bf16_t* {
      const bf16_t* base = isA ? A : B;
      const long ld = isA ? p.lda : p.ldb;
      const int ext = isA ? p.M : p.N, x0 = isA ? m0 : n0;
      const bool km = isA ? AKM : BKM;
      if (!km) {
        const int prow = lane >> 3;
        int r = x0 + po * 8 + prow; r = r < ext ? r : ext - 1;
        int k = tile * BK + (((lane & 7) ^ prow) << 3);
        k = k <= K - 8 ? k : K - 8;
        return base + (long)r * ld + k;
      } else {
        const int krow = po * 4 + (lane >> 4);
        const int pc = lane & 15;
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        const int lchunk = (((pc >> 1) ^ f) << 1) | (pc & 1);
        int col = x0 + lchunk * 8; col = col <= ext - 8 ? col : ext - 8;
        int k = tile * BK + krow; k = k < K ? k : K - 1;
        return base + (long)k * ld + col;
      }
    };
    // per-tile piece bases (k = 0 position of each of this wave's 16 pieces); a unit then costs one 64-bit add per
    // piece. Only the K-tail unit takes the clamped slow path.
    const bf16_t* gbase[PPW];
    const long stepA = AKM ? (long)BK * p.lda : (long)BK, stepB = BKM ? (long)BK * p.ldb : (long)BK;
    auto set_bases = [&]() {
#pragma unroll
      for (int j = 0; j < PPW; ++j) gbase[j] = src_ptr(j < PPW / 2, (iw + NLOAD * j) & 15, 0);
    };
    set_bases();
    auto issue_next = [&](int stage) {   // issues unit (ti, ki) and advances the issue-side state
      char* sb = smem + stage * STAGE;
      const bool tail = krem && (ki == nt - 1);
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const bool isA = j < PPW / 2;
        const int po = (iw + NLOAD * j) & 15;
        const bf16_t* src = tail ? src_ptr(isA, po, ki) : gbase[j] + (long)ki * (isA ? stepA : stepB);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sb + (isA ? 0 : OPB) + po * 1024),
                                         16, 0, 0);
      }
      if (++ki == nt) {
        ki = 0;
        if (++ti < ntile_my) {
          tile_of(p, first + ti * stride, bz, m0, n0);
          A = p.A + (bz / p.nb2) * p.sA1 + (bz % p.nb2) * p.sA2;
          B = p.B + (bz / p.nb2) * p.sB1 + (long)((bz % p.nb2) / p.b2divB) * p.sB2;
          set_bases();
        }
      }
    };
    int issued = 0;
    for (; issued < NSTAGE - 1 && issued < units; ++issued) issue_next(issued);
    int stage = 0, kc = 0;
    for (int u = 0; u < units; ++u) {
      const int rem = units - 1 - u;
      const int newer = rem < (NSTAGE - 2) ? rem : (NSTAGE - 2);    // units allowed to stay in flight behind unit u
      if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (issued < units) {
        int s2 = stage + NSTAGE - 1; s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
        issue_next(s2);
        ++issued;
      }
      if (++kc == nt) {
        kc = 0;
        if (krem) __builtin_amdgcn_s_barrier();   // the compute waves' zero-fill barrier of this tile's last unit (raw:
                                                  // __syncthreads() would drain the DMA ring with a vmcnt(0))
      }
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int wm = wid / WN, wn = wid % WN;
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TM], b_off[TN];
  {
    const int li = fr;
    const int fl = (li >> 2) | ((fq & 1) << 2);
    const int lane_part = (((li >> 1) & 1) << 4) | ((li & 1) << 3);
    const int krow_base = 8 * fq + (li >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * 32 + i * 16;
      if (AKM) a_off[i] = krow_base * 256 + (((row >> 4) ^ fl) << 5) + lane_part;
      else { const int rr = row + fr; a_off[i] = rr * 128 + ((fq ^ (rr & 7)) << 4); }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * 64 + j * 16;
      if (BKM) b_off[j] = OPB + krow_base * 256 + (((row >> 4) ^ fl) << 5) + lane_part;
      else { const int rr = row + fr; b_off[j] = OPB + rr * 128 + ((fq ^ (rr & 7)) << 4); }
    }
  }
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  auto compute = [&](int stage) {
    const char* sb = smem + stage * STAGE;
    const unsigned sa = lds0 + stage * STAGE;
    u32x2 alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
    bf16x8 xa[2][TM], wb[2][TN];
    if (AKM) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        alo[0][i] = ds_tr4<0>(sa + a_off[i]); ahi[0][i] = ds_tr4<4 * 256>(sa + a_off[i]);
        alo[1][i] = ds_tr4<32 * 256>(sa + a_off[i]); ahi[1][i] = ds_tr4<36 * 256>(sa + a_off[i]);
      }
    }
    if (BKM) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        blo[0][j] = ds_tr4<0>(sa + b_off[j]); bhi[0][j] = ds_tr4<4 * 256>(sa + b_off[j]);
        blo[1][j] = ds_tr4<32 * 256>(sa + b_off[j]); bhi[1][j] = ds_tr4<36 * 256>(sa + b_off[j]);
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
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (AKM) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[ks][i] = join4(alo[ks][i], ahi[ks][i]);
      }
      if (BKM) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[ks][j] = join4(blo[ks][j], bhi[ks][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks][j], xa[ks][i], acc[i][j], 0, 0, 0);
    }
  };
  auto zero_tail = [&](int stage) {   // compute waves only; the loaders join the barrier
    char* sa = smem + stage * STAGE;
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (AKM) {
      const int n16 = (BK - krem) * 16;
      for (int idx = tid; idx < n16; idx += 64 * NW) *reinterpret_cast<u32x4*>(sa + krem * 256 + idx * 16) = z;
    } else {
      const int c0 = krem >> 3, nch = 8 - c0;
      for (int idx = tid; idx < BM * nch; idx += 64 * NW) {
        const int row = idx & (BM - 1), kc = c0 + idx / BM;
        *reinterpret_cast<u32x4*>(sa + row * 128 + ((kc ^ (row & 7)) << 4)) = z;
      }
    }
    __syncthreads();
  };

  int tc = 0, kc = 0, stage = 0;
  int bz, m0, n0;
  tile_of(p, first, bz, m0, n0);
  for (int u = 0; u < units; ++u) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const bool last_k = (kc == nt - 1);
    if (last_k && krem) zero_tail(stage);
    compute(stage);
    if (last_k) {
      const long coff = (bz / p.nb2) * p.sC1 + (bz % p.nb2) * p.sC2;
      const long roff = (bz / p.nb2) * p.sR1 + (bz % p.nb2) * p.sR2;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 32 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * 64 + j * 16 + 4 * fq;
          if (m < p.M && n < p.N) store_quad<OUT_F32>(p, coff, roff, m, n, acc[i][j]);
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      kc = 0;
      if (++tc < ntile_my) tile_of(p, first + tc * stride, bz, m0, n0);
    } else {
      ++kc;
    }
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
  }
}

template <bool AKM, bool BKM>
int launch_v4(GemmParams& p, int nbatch, int ncu, hipStream_t stream) {
  constexpr int SMEM = NSTAGE * STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_v4_kernel<AKM, BKM, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_v4_kernel<AKM, BKM, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vq3_set_error("gemm v4: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 2;
    }
    attr_done = true;
  }
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  p.xm = choose_xm(p.mtiles, p.ntiles);
  const long total = (long)p.mtiles * p.ntiles * nbatch;
  if (total >= (1L << 31)) { vq3_set_error("gemm v4: too many tiles"); return 1; }
  const int grid = total < ncu ? (int)total : ncu;
  if (p.out_f32)
    hipLaunchKernelGGL((gemm_v4_kernel<AKM, BKM, true>), dim3(grid), dim3(64 * (NW + NLOAD)), SMEM, stream, p, (int)total);
  else
    hipLaunchKernelGGL((gemm_v4_kernel<AKM, BKM, false>), dim3(grid), dim3(64 * (NW + NLOAD)), SMEM, stream, p, (int)total);
  return 0;
}

}  // namespace

int launch_gemm_v4(GemmParams& p, int transA, int transB, int nbatch, int ncu, hipStream_t stream) {
  switch ((transA ? 2 : 0) | (transB ? 1 : 0)) {
    case 0: return launch_v4<false, false>(p, nbatch, ncu, stream);
    case 1: return launch_v4<false, true>(p, nbatch, ncu, stream);
    case 2: return launch_v4<true, false>(p, nbatch, ncu, stream);
    default: return launch_v4<true, true>(p, nbatch, ncu, stream);
  }
}

}  // namespace vq3gemm
