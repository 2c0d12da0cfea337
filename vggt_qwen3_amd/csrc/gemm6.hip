// GEMM v6 for gfx950: 256x256x64 tile, 8 waves (2 x 4), the 8-phase schedule of the CDNA4 guide (two K tiles per
// iteration, four phases per K tile), LDS-DMA staging with a counted vmcnt, two wave groups staggered by one barrier.
//
// Same contract as the other NT kernels (C = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, f32 MFMA accumulate,
// K % 64 == 0). What is different is the schedule of the main loop:
//
//   * LDS = 2 buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 64 k (16 KiB each, 128 KiB in all). Half-tile Xh
//     holds tile rows h*128 .. h*128+127; inside it wave row wr owns rows wr*64..+63 (A) and wave column wc owns rows
//     wc*32..+31 (B): a wave's 128 x 64 output is the four quadrants (i, j) = (A half, B half) of 64 x 32, and every
//     half-tile is read in exactly one phase of the K tile:
//         phase 1: read A0, B0   -> MFMA quadrant (0,0)        stage A1 of tile t+1
//         phase 2: read B1       -> MFMA quadrant (0,1)        stage A0 of tile t+2   (A0 was last read in phase 1)
//         phase 3: read A1       -> MFMA quadrant (1,1)        stage B0 of tile t+2
//         phase 4: (B0 kept)     -> MFMA quadrant (1,0)        stage B1 of tile t+2, s_waitcnt vmcnt(6)
//     so a slot is restaged one phase after its last read and three half-tiles stay in flight across every barrier.
//   * A phase is  { ds_reads ; 2 x global_load_lds ; lgkmcnt(0) ; s_barrier ; 16 MFMA ; s_barrier }. Waves 4-7 run one
//     barrier behind waves 0-3, so in every barrier interval one wave of each SIMD is in its MFMA segment and the
//     other in its load segment. The reads are retired BEFORE the phase's first barrier: a slot restaged in phase p+1
//     by either group is then free for both (the lagging group's reads of phase p finished before the barrier the
//     leading group passes on its way into phase p+1).
//   * RAW: the vmcnt(6) of phase 4 sits before that phase's first barrier and retires every half-tile of tile t+1; tile
//     t+1 is first read in the next phase, i.e. after a barrier every issuing wave reached after its wait.
#include <map>
#include <mutex>

#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int BK6 = 64;
constexpr int HALF = 128 * 128;          // bytes per half-tile (128 rows x 64 k)

#define V6_FENCE() asm volatile("" ::: "memory")
// diagnostic stamps (p.stamps != nullptr only under tools/gemm_stamps.py): wall clock (100 MHz) of thread 0 at the marks of one tile
#define V6_STAMP(IDX)                                                                           \
  do {                                                                                          \
    if (p.stamps && tid == 0) {                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                        \
      p.stamps[(long)tile * 8 + (IDX)] = __builtin_amdgcn_s_memrealtime();                      \
      __builtin_amdgcn_sched_barrier(0);                                                        \
    }                                                                                           \
  } while (0)

// AH / BH = number of 128-row half-tiles of A / B per K tile: (2,2) = 256x256 (4 phases per K tile), (2,1) = 256x128 and
// (1,2) = 128x256 (2 phases per K tile, same 16 MFMAs per phase and wave).
// EK = epilogue kind, one instantiation each: 0 = plain C epilogue (alpha / bias / activation / LayerScale / residual / accumulate / output
// statistics), 3 = the same behind a folded LayerNorm (p.ln_in), 1 = fused q|k|v epilogue (LayerNorm fold optional), 2 = SwiGLU backward,
// 4 = SwiGLU forward (B = gate|up weight [2 I, K]: a tile multiplies BN/2 gate rows and the SAME BN/2 up rows, see set_offsets).
// Compiled into ONE kernel they cost the 256 x 256 instantiation (256 VGPRs) 80 spilled registers in every launch's epilogue: the plain
// fc1 launch (49 392 x 4096 x 1024) took 524 us with them and 443 us without (tools/gemm_stamps.py: "stage C" 13.1 -> 6.6 us per tile).
// SPLIT (256 x 256; plain, LayerNorm-fold and fused q|k|v epilogues - the reducer runs them on the summed tile; on the tower's own shapes,
// 61740 x 4096 / 3072 x 1024 = 15 rounds + 32 tiles / 11 rounds + 88, it measures 598 against 585 us and 465 against 439 - tools/
// bench_tower_split.py - so the shipped table keeps cfg 20 there): the launch's last, partly filled round of tiles is cut along K over the CUs it would leave idle
// (GemmParams: sk_*). 9600 x 2560 outputs - every o / down projection and two of the four dgrads of a pass of 8 micro-batches - are 380
// tiles = 1.48 rounds of 256 CUs: 124 tiles x 2 K halves run as ONE half-length round instead of a full one.
// F8 (256 x 256, EK 0 / 2 / 4; config C5): A and B are OCP e4m3 bytes, a K tile is 128 elements - the same 128-byte rows, the same DMA
// image, the same number of LDS reads, and ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales, 32 cycles) where the bf16 kernel
// issues two 16x16x32 MFMAs (16 cycles each) per K tile and 16 x 16 output: twice the FLOPs per K tile at the same cadence. The real
// scales - one per A row (token), one per B row (output channel) - multiply the accumulators in the epilogue (p.f8_rs / p.f8_cs).
typedef int i32x8_t __attribute__((ext_vector_type(8)));
// KM (256 x 256, plain epilogue; round 4): BOTH operands k-major - A[k * lda + m], B[k * ldb + n], the weight-gradient products
// dW = dY^T . X with dY and X as the backward stores them (token-major). A half-tile is then a 64(k) x 128 image of 256-byte k rows
// (gemm3.hip's sub-tile: 16 DMA pieces of 4 k rows, 32-byte segments XORed with (k & 3) | ((k >> 3) & 1) << 2 on the source side) and a
// fragment is two ds_read_b64_tr_b16 instead of one ds_read_b128; buffers, phases, waits and barriers are the NT kernel's. K % 8 == 0:
// the k rows past K of the last tile are read from a row of zeros (both operands), so nothing is zero-filled in LDS.
__device__ __attribute__((aligned(256))) bf16_t g_v6_zero_row[256];
template <int OFF>
__device__ __forceinline__ u32x2 v6_ds_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int AH, int BH, bool OUT_F32, int EK = 0, bool SPLIT = false, bool F8 = false, bool KM = false>
__global__ __launch_bounds__(512, 2) void gemm_v6_kernel(GemmParams p) {
  constexpr bool HAS_LN = (EK == 1 || EK == 3);
  constexpr int BM = 128 * AH, BN = 128 * BH;
  constexpr int NSLOT = AH + BH;
  constexpr int BUF = NSLOT * HALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  // The two-phase variants are PERSISTENT: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (gridDim.x = #CUs, a multiple of 8, so
  // it stays on its XCD's slice of the tile order) and requests the next tile's first two K tiles BEFORE it runs the current
  // tile's epilogue: the C image lives above the two buffers they land in (160 KiB of LDS), so at K = 1024 - 16 K steps per tile,
  // the VGGT blocks - a tile no longer starts with an empty pipeline (the round trip was 2-3 us of a 21-26 us tile).
  constexpr bool PERSIST = (AH + BH) < 4;
  char* const smem_c = PERSIST ? smem + 2 * BUF : smem;       // C image / epilogue scratch
  const int ntile = p.mtiles * p.ntiles;
  int m0 = 0, n0 = 0;
  // folded LayerNorm: the two-phase variants (199 VGPRs) fetch row r's statistics at the tile's start, thread r < BM - two registers
  // through the main loop; the 256 x 256 kernel has none to spare (spills, 3 % on every launch) and fetches them inside the epilogue
  constexpr bool LN_EARLY = (AH + BH) < 4;
  float ln_mu = 0.f, ln_rs = 0.f;
  const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z % p.nb2;
  const bf16_t* A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* B = p.B + b1 * p.sB1 + (long)(b2 / p.b2divB) * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;
  const long roff = b1 * p.sR1 + b2 * p.sR2;

  // ---- DMA sources. A half-tile is 16 pieces of 8 rows x 128 B; wave w issues pieces w and w + 8. Lane l of a piece
  // reads k-chunk (l & 7) ^ (l >> 3) of row l >> 3 (the read-side swizzle chunk ^ (row & 7), applied to the source).
  const int prow = lane >> 3;
  const int kch = (lane & 7) ^ prow;
  unsigned offA[AH][2], offB[BH][2];     // byte offsets from A / B, [half][piece]
  auto set_offsets = [&](int tm0, int tn0) {
    if constexpr (KM) {
      // piece po = wid + 8 j holds k rows 4 po .. 4 po + 3; lane: k row 4 po + (lane >> 4), 16-byte chunk (lane & 15) of its 256 bytes,
      // source chunk = the segment-swizzled one (gemm3.hip: src_ptr)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int krow = (wid + 8 * j) * 4 + (lane >> 4);
        const int pc = lane & 15;
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        const int lchunk = (((pc >> 1) ^ f) << 1) | (pc & 1);
#pragma unroll
        for (int h = 0; h < AH; ++h) {
          int col = tm0 + h * 128 + lchunk * 8; col = col <= p.M - 8 ? col : p.M - 8;
          offA[h][j] = (unsigned)(((long)krow * p.lda + col) * 2);
        }
#pragma unroll
        for (int h = 0; h < BH; ++h) {
          int col = tn0 + h * 128 + lchunk * 8; col = col <= p.N - 8 ? col : p.N - 8;
          offB[h][j] = (unsigned)(((long)krow * p.ldb + col) * 2);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wid + 8 * j) * 8 + prow;
#pragma unroll
      for (int h = 0; h < AH; ++h) {
        int ra = tm0 + h * 128 + r; ra = ra < p.M ? ra : p.M - 1;
        offA[h][j] = F8 ? (unsigned)((long)ra * p.lda + kch * 16) : (unsigned)(((long)ra * p.lda + kch * 8) * 2);
      }
#pragma unroll
      for (int h = 0; h < BH; ++h) {
        int rb = tn0 + h * 128 + r;
        if constexpr (EK == 4) {       // tile columns [0, BN/2) = gate features tn0/2 .., [BN/2, BN) = the same up features (weight rows I + ..)
          const int li = h * 128 + r;
          rb = li < BN / 2 ? (tn0 >> 1) + li : (p.N >> 1) + (tn0 >> 1) + li - BN / 2;
        }
        rb = rb < p.N ? rb : p.N - 1;
        offB[h][j] = F8 ? (unsigned)((long)rb * p.ldb + kch * 16) : (unsigned)(((long)rb * p.ldb + kch * 8) * 2);
      }
    }
  };
  int tail_tile = -1, krem = 0;          // (KM) K tile whose k rows >= krem lie past K: read from the row of zeros
  auto stage_op = [&](const bf16_t* base, long ld, const unsigned (&off)[2], int tile, char* slot) {
    const char* g = reinterpret_cast<const char*>(base) + (KM ? (long)tile * (BK6 * 2) * ld : (long)tile * (BK6 * 2));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* src = g + off[j];
      if constexpr (KM) {
        if (tile == tail_tile && (wid + 8 * j) * 4 + (lane >> 4) >= krem) src = reinterpret_cast<const char*>(g_v6_zero_row) + (lane & 15) * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + (wid + 8 * j) * 1024), 16, 0, 0);
    }
  };
  auto stageA = [&](const unsigned (&off)[2], int tile, char* slot) { stage_op(A, p.lda, off, tile, slot); };
  auto stageB = [&](const unsigned (&off)[2], int tile, char* slot) { stage_op(B, p.ldb, off, tile, slot); };

  f32x4 acc[AH * 4][BH * 2];

  const int fr = lane & 15, fq = lane >> 4;
  // bf16: lane (fr, fq) holds k = 8 fq .. + 7 of half kh (16-byte chunk fq + 4 kh); e4m3: k = 32 fq .. + 31 (chunks 2 fq and 2 fq + 1,
  // kept as the two halves [.][0], [.][1] of the operand)
  const int a_base = (wr * 64 + fr) * 128 + (((F8 ? 2 * fq : fq) ^ (fr & 7)) << 4);
  const int b_base = (wc * 32 + fr) * 128 + (((F8 ? 2 * fq : fq) ^ (fr & 7)) << 4);
  constexpr int KHX = F8 ? 4 : 6;        // address bit that separates the two 16-byte pieces of a fragment

  // k-major fragments (gemm3.hip): lane (li = fr, fq) of k step ks takes columns li of two 4-row x 16-column blocks - k rows
  // 32 ks + 8 fq + (li >> 2) and + 4 - of the 16-column group g: byte = k row * 256 + ((g ^ fl) << 5) + lane part
  const int km_fl = (fr >> 2) | ((fq & 1) << 2);
  const int km_base = (8 * fq + (fr >> 2)) * 256 + ((((fr >> 1) & 1) << 4) | ((fr & 1) << 3));
  unsigned km_a[4], km_b[2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) km_a[mt] = (unsigned)(km_base + ((((wr * 4 + mt)) ^ km_fl) << 5));
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) km_b[n2] = (unsigned)(km_base + ((((wc * 2 + n2)) ^ km_fl) << 5));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto tr_join = [](const u32x2& lo, const u32x2& hi) -> bf16x8 {
    const u32x4 t = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, t);
  };

  bf16x8 xa[4][2], wb0[2][2], wb1[2][2];
  // (KM) the transposed reads go through asm, invisible to hipcc's waitcnt pass: their destinations stay RAW (lo / hi halves) until the
  // phase's own s_waitcnt lgkmcnt(0) has passed - km_fix_* pins them behind it (an empty volatile statement per register pair: volatile
  // statements keep their order) and only then forms the MFMA operands (cdna guide 5.7 item 1; gemm3.hip does the same)
  u32x2 ra_lo[4][2], ra_hi[4][2], rb_lo[2][2], rb_hi[2][2];
  auto km_fix_a = [&]() {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        asm volatile("" : "+v"(ra_lo[mt][kh]), "+v"(ra_hi[mt][kh]));
        xa[mt][kh] = tr_join(ra_lo[mt][kh], ra_hi[mt][kh]);
      }
  };
  auto km_fix_b = [&](bf16x8 (&wb)[2][2]) {
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        asm volatile("" : "+v"(rb_lo[n2][kh]), "+v"(rb_hi[n2][kh]));
        wb[n2][kh] = tr_join(rb_lo[n2][kh], rb_hi[n2][kh]);
      }
  };
  auto read_a = [&](const char* slot) {
    if constexpr (KM) {
      const unsigned sa = lds0 + (unsigned)(slot - smem);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        ra_lo[mt][0] = v6_ds_tr<0>(sa + km_a[mt]); ra_hi[mt][0] = v6_ds_tr<4 * 256>(sa + km_a[mt]);
        ra_lo[mt][1] = v6_ds_tr<32 * 256>(sa + km_a[mt]); ra_hi[mt][1] = v6_ds_tr<36 * 256>(sa + km_a[mt]);
      }
      return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xa[mt][kh] = *reinterpret_cast<const bf16x8*>(slot + ((a_base ^ (kh << KHX)) + mt * 2048));
  };
  auto read_b = [&](const char* slot, bf16x8 (&wb)[2][2]) {
    if constexpr (KM) {
      const unsigned sa = lds0 + (unsigned)(slot - smem);
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        rb_lo[n2][0] = v6_ds_tr<0>(sa + km_b[n2]); rb_hi[n2][0] = v6_ds_tr<4 * 256>(sa + km_b[n2]);
        rb_lo[n2][1] = v6_ds_tr<32 * 256>(sa + km_b[n2]); rb_hi[n2][1] = v6_ds_tr<36 * 256>(sa + km_b[n2]);
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wb[nt][kh] = *reinterpret_cast<const bf16x8*>(slot + ((b_base ^ (kh << KHX)) + nt * 2048));
  };
  auto cat8 = [](const bf16x8& lo, const bf16x8& hi) -> i32x8_t {
    const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
    return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
#define V6_MMA(I, J, WB)                                                                                         \
  do {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    if constexpr (F8) {                                                                                          \
      _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                           \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                         \
          acc[(I) * 4 + mt][(J) * 2 + nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                    \
              cat8(WB[nt][0], WB[nt][1]), cat8(xa[mt][0], xa[mt][1]), acc[(I) * 4 + mt][(J) * 2 + nt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
    } else {                                                                                                     \
      _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                           \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                         \
          _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                       \
            acc[(I) * 4 + mt][(J) * 2 + nt] =                                                                    \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[nt][kh], xa[mt][kh], acc[(I) * 4 + mt][(J) * 2 + nt], 0, 0, 0); \
    }                                                                                                            \
    __builtin_amdgcn_s_setprio(0);                                                                               \
  } while (0)
  // first barrier of a phase: this wave's LDS reads are retired, then everybody meets
#define V6_SYNC_A()                                     \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                       \
    V6_FENCE();                                         \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
#define V6_SYNC_B()                     \
  do {                                  \
    V6_FENCE();                         \
    __builtin_amdgcn_s_barrier();       \
    V6_FENCE();                         \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

  int nt = F8 ? p.K / 128 : (p.K + BK6 - 1) / BK6;          // K tiles of 128 bytes per row (k-major: of 64 k rows, the last one ragged)
  if constexpr (KM) {
    krem = p.K & (BK6 - 1);
    tail_tile = krem ? nt - 1 : -1;
  }
  int sk_slice = -1, sk_j = 0;       // (SPLIT) this workgroup's K slice of tile sk_full + sk_j; -1: a whole tile
  // two-phase variants keep THREE K-tile buffers (3 x 48 KiB): tile t+2 is staged whole while tile t is multiplied, into the
  // buffer tile t-1 left a full tile ago. Slots: A0 A1 B (256 x 128) or A B0 B1 (128 x 256).
  auto stage_tile = [&](int t, char* buf) {
    if constexpr (AH == 2 && BH == 1) {
      stageA(offA[0], t, buf + 0 * HALF);
      stageB(offB[0], t, buf + 2 * HALF);
      stageA(offA[1], t, buf + 1 * HALF);
    } else if constexpr (AH == 1 && BH == 2) {
      stageA(offA[0], t, buf + 0 * HALF);
      stageB(offB[0], t, buf + 1 * HALF);
      stageB(offB[1], t, buf + 2 * HALF);
    }
  };
  bool prefetched = false;         // this tile's first K tiles were requested during the previous tile's epilogue (and have landed)
  int tile = blockIdx.x;           // (< ntile: the grid never exceeds the tile count)
  if constexpr (SPLIT) {
    if (p.sk_s > 1 && tile >= p.sk_full) {
      // workgroup sk_full + s * R8 + j, R8 = sk_rem rounded up to a multiple of 8: the slices of a tile run on the XCD whose L2 holds the
      // panels its neighbours in the tile order share (workgroup id mod 8 = XCD under round-robin placement; speed only)
      const int r8 = (p.sk_rem + 7) & ~7;
      const int r = tile - p.sk_full;
      sk_slice = r / r8;
      sk_j = r - sk_slice * r8;
      if (sk_j >= p.sk_rem) return;                  // (padding ids of a slice group)
      tile = p.sk_full + sk_j;
      const int per = (nt + p.sk_s - 1) / p.sk_s;
      const int kt0 = sk_slice * per;
      if constexpr (KM) tail_tile = krem ? nt - 1 - kt0 : -1;     // (relative to this slice; out of its range for all but the last slice)
      nt = nt - kt0 < per ? nt - kt0 : per;        // (the host keeps every slice non-empty)
      A += KM ? (long)kt0 * BK6 * p.lda : (long)kt0 * BK6;
      B += KM ? (long)kt0 * BK6 * p.ldb : (long)kt0 * BK6;
    }
  }
  // Start-up stagger: every tile of a launch costs the same, so the CUs run in lockstep and reach their epilogues - the C-tile stores, the
  // residual reads - in the same microseconds: bursts of 32-64 MB against an otherwise idle memory system, with the MFMA pipes waiting
  // (tools/gemm_stamps.py: "issue stores" 13.7 us per tile with a residual). Workgroups of the first round wait (workgroup % 8) * stagger
  // ticks (workgroup % 8 = the XCD under round-robin placement; only speed depends on it): the offsets then persist from round to round.
  if (p.stagger > 0 && blockIdx.x < 256 && blockIdx.z == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)((blockIdx.x & 7) * p.stagger);
    while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
  }
  do {                             // one pass for the 256 x 256 kernel (PERSIST is a compile-time false there: no loop is generated)
  tile_coords_id(p, tile, BM, BN, m0, n0);
  V6_STAMP(0);
  if (p.stamps && tid == 0)
    p.stamps[(long)tile * 8 + 7] = ((unsigned long long)blockIdx.x << 32) | __builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID, 4 bits */) |
                                   ((unsigned long long)(__builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID, 32 bits */) & 0xff00u) << 8);   // CU / SH / SE id -> bits 16-23
  if (HAS_LN && LN_EARLY && p.ln_in && tid < BM) ln_row(p, m0 + tid, ln_mu, ln_rs);
#pragma unroll
  for (int i = 0; i < AH * 4; ++i)
#pragma unroll
    for (int j = 0; j < BH * 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- prologue: all of tile 0, then the half-tiles of tile 1 that the steady state stages ahead of a tile's last phase
  if (!prefetched) set_offsets(m0, n0);
  if constexpr (AH == 2 && BH == 2) {
    stageA(offA[0], 0, smem + 0 * HALF);
    stageB(offB[0], 0, smem + 2 * HALF);
    stageB(offB[1], 0, smem + 3 * HALF);
    stageA(offA[1], 0, smem + 1 * HALF);
    if (nt > 1) {
      stageA(offA[0], 1, smem + BUF + 0 * HALF);
      stageB(offB[0], 1, smem + BUF + 2 * HALF);
      stageB(offB[1], 1, smem + BUF + 3 * HALF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else if (!prefetched) {
    stage_tile(0, smem);
    if (nt > 1) {
      stage_tile(1, smem + BUF);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  V6_FENCE();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) {   // waves 4-7 run one barrier behind
    __builtin_amdgcn_s_barrier();
    V6_FENCE();
    __builtin_amdgcn_sched_barrier(0);
  }
  V6_STAMP(1);      // first operands have landed: main loop starts

  auto ktile = [&](int t, char* cur, char* nxt) {
    const bool s1 = t + 1 < nt, s2 = t + 2 < nt;
    if constexpr (AH == 2 && BH == 2) {
      // phase 1
      read_b(cur + 2 * HALF, wb0);
      read_a(cur + 0 * HALF);
      if (s1) stageA(offA[1], t + 1, nxt + 1 * HALF);
      V6_SYNC_A();
      if constexpr (KM) { km_fix_b(wb0); km_fix_a(); __builtin_amdgcn_sched_barrier(0); }
      V6_MMA(0, 0, wb0);
      V6_SYNC_B();
      // phase 2
      read_b(cur + 3 * HALF, wb1);
      if (s2) stageA(offA[0], t + 2, cur + 0 * HALF);
      V6_SYNC_A();
      if constexpr (KM) { km_fix_b(wb1); __builtin_amdgcn_sched_barrier(0); }
      V6_MMA(0, 1, wb1);
      V6_SYNC_B();
      // phase 3
      read_a(cur + 1 * HALF);
      if (s2) stageB(offB[0], t + 2, cur + 2 * HALF);
      V6_SYNC_A();
      if constexpr (KM) { km_fix_a(); __builtin_amdgcn_sched_barrier(0); }
      V6_MMA(1, 1, wb1);
      V6_SYNC_B();
      // phase 4
      if (s2) {
        stageB(offB[1], t + 2, cur + 3 * HALF);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      V6_SYNC_A();
      V6_MMA(1, 0, wb0);
      V6_SYNC_B();
    } else if constexpr (AH == 2) {
      // 256 x 128: { A0, B -> quadrant (0,0) ; stage A0, B of t+2 }  { A1 -> quadrant (1,0) ; stage A1 of t+2 ; vmcnt }
      // (`nxt` is the buffer of tile t+2 here: last read a whole tile ago, so no slot is restaged near its readers)
      read_b(cur + 2 * HALF, wb0);
      read_a(cur + 0 * HALF);
      if (s2) {
        stageA(offA[0], t + 2, nxt + 0 * HALF);
        stageB(offB[0], t + 2, nxt + 2 * HALF);
      }
      V6_SYNC_A();
      V6_MMA(0, 0, wb0);
      V6_SYNC_B();
      read_a(cur + 1 * HALF);
      if (s2) {
        stageA(offA[1], t + 2, nxt + 1 * HALF);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile t+1 (staged during tile t-1) has landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      V6_SYNC_A();
      V6_MMA(1, 0, wb0);
      V6_SYNC_B();
    } else {
      // 128 x 256: { A, B0 -> (0,0) ; stage A, B0 of t+2 }  { B1 -> (0,1) ; stage B1 of t+2 ; vmcnt }
      read_b(cur + 1 * HALF, wb0);
      read_a(cur + 0 * HALF);
      if (s2) {
        stageA(offA[0], t + 2, nxt + 0 * HALF);
        stageB(offB[0], t + 2, nxt + 1 * HALF);
      }
      V6_SYNC_A();
      V6_MMA(0, 0, wb0);
      V6_SYNC_B();
      read_b(cur + 2 * HALF, wb1);
      if (s2) {
        stageB(offB[1], t + 2, nxt + 2 * HALF);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      V6_SYNC_A();
      V6_MMA(0, 1, wb1);
      V6_SYNC_B();
    }
  };
  if constexpr (AH == 2 && BH == 2) {
    for (int t = 0; t < nt; t += 2) {
      ktile(t, smem, smem + BUF);
      if (t + 1 < nt) ktile(t + 1, smem + BUF, smem);
    }
  } else {
    for (int t = 0; t < nt; t += 3) {
      ktile(t, smem, smem + 2 * BUF);
      if (t + 1 < nt) ktile(t + 1, smem + BUF, smem);
      if (t + 2 < nt) ktile(t + 2, smem + 2 * BUF, smem + BUF);
    }
  }
  if (wr == 0) {   // balance the stagger barrier
    __builtin_amdgcn_s_barrier();
    V6_FENCE();
  }

  // ---- epilogue
  V6_STAMP(2);      // main loop done
  if constexpr (SPLIT) {
    if (sk_slice >= 0) {
      // partial tiles in the accumulators' own order: quad q of thread tid at [q][tid] (a wave instruction moves 1 KiB contiguous)
      float* ws = p.sk_ws + (long)sk_j * (p.sk_s - 1) * (BM * BN);
      if (sk_slice < p.sk_s - 1) {
        // publish (cdna guide, Guideline 16 R1 / price list "publish-large"): WRITE-THROUGH (sc1) 16-byte stores - a 256 KiB partial tile
        // behind plain stores + an agent-scope release costs the write-back of everything the XCD's L2 holds dirty (the neighbours' C
        // tiles included) - every storing wave drains its own stores, the workgroup meets, ONE lane counts the slice in
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(ws + (long)sk_slice * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < AH * 4; ++i)
#pragma unroll
          for (int j = 0; j < BH * 2; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (tid * 4 + (i * BH * 2 + j) * 2048) * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.sk_cnt + sk_j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      // the last slice (highest workgroup id of the tile: dispatched last) adds the others' partials, then runs the usual epilogue.
      // Bounded wait: all sk_rem * sk_s <= #CUs split workgroups can be resident together and the unsplit tiles ahead of them wait
      // for nobody, so the count always arrives; the bound only turns a broken launch into an error word instead of a hang.
      if (tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(p.sk_cnt + sk_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p.sk_s - 1)) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > p.sk_spin) {
            __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // sticky, host-visible
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      // 16 quads (16 KiB per wave) in flight per batch: the partials come from another XCD's write-back, a 2-4 us round trip each
      // (4 in flight: 8 dependent round trips per partial tile, +25 us per split tile)
      for (int sl = 0; sl < p.sk_s - 1; ++sl) {
        const float* src = ws + (long)sl * (BM * BN) + tid * 4;
#pragma unroll
        for (int i0 = 0; i0 < AH * 4; i0 += 4) {
          f32x4 t[4][BH * 2];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < BH * 2; ++j) t[i][j] = *reinterpret_cast<const f32x4*>(src + ((i0 + i) * BH * 2 + j) * 2048);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < BH * 2; ++j) acc[i0 + i][j] += t[i][j];
        }
      }
    }
  }
  const int next_tile = tile + (int)gridDim.x;
  const bool more = PERSIST && next_tile < ntile;
  if constexpr (!OUT_F32) {      // (the host launches the bf16 instantiations only when host_staged_ok holds)
    // through LDS (the operand buffers are dead: every wave passed the last phase's barrier), out as whole rows
    // folded LayerNorm: rstd (acc - mu c) per lane-owned row, before the usual epilogue. The 256 x 256 kernel keeps the rows' (mu, rstd)
    // pairs in 2 KiB of LDS ABOVE the C image (its 128 KiB operand buffers are all the image needs) and applies them quad by quad inside
    // the staging loop: as a pass of its own over the 128 accumulator registers it spilled 80 of them in every launch. The two-phase
    // variants (64 accumulator registers, LDS full at 160 KiB) pass the pairs through the C image's space first.
    constexpr bool LN_FUSED = HAS_LN && (AH + BH == 4);
    const bool ln_on = HAS_LN && p.ln_in;
    const float2* lnp = reinterpret_cast<const float2*>(smem + 2 * BUF);      // (LN_FUSED only)
    f32x4 cc[BH * 2];
    if (ln_on) {
      __syncthreads();                                       // operand buffers are dead for every wave
      if (tid < BM) {
        float mu_t = ln_mu, rs_t = ln_rs;
        if (!LN_EARLY) ln_row(p, m0 + tid, mu_t, rs_t);
        reinterpret_cast<float2*>(LN_FUSED ? smem + 2 * BUF : smem_c)[tid] = float2{mu_t, rs_t};
      }
#pragma unroll
      for (int j = 0; j < BH; ++j)
#pragma unroll
        for (int nt2 = 0; nt2 < 2; ++nt2) cc[j * 2 + nt2] = ln_colsum(p, n0 + j * 128 + wc * 32 + nt2 * 16 + 4 * fq);
      __syncthreads();
      if (!LN_FUSED) {
#pragma unroll
        for (int i = 0; i < AH; ++i)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const float2 v = reinterpret_cast<const float2*>(smem_c)[i * 128 + wr * 64 + mt * 16 + fr];
#pragma unroll
            for (int q = 0; q < BH * 2; ++q) acc[i * 4 + mt][q] = ln_apply(acc[i * 4 + mt][q], v.x, v.y, cc[q]);
          }
        __syncthreads();                                       // before the C image overwrites the pairs
      }
      __builtin_amdgcn_sched_barrier(0);                     // (keeps the bias / LayerScale loads below from being hoisted into this block)
    }
    f32x4 bias_r[BH][2], cs_r[BH][2];
#pragma unroll
    for (int j = 0; j < BH; ++j)
#pragma unroll
      for (int nt2 = 0; nt2 < 2; ++nt2) {
        int n = n0 + j * 128 + wc * 32 + nt2 * 16 + 4 * fq;
        n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);          // columns past N are never stored
        if (p.bias) bias_r[j][nt2] = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.colscale) cs_r[j][nt2] = *reinterpret_cast<const f32x4*>(p.colscale + n);
      }
    EpiPre pre;
    pre.on = false;
    if constexpr (EK == 0) staged_prefetch<BM, BN>(p, coff, roff, m0, n0, tid, 512, pre);
    if (more) {
      // the next tile's first two K tiles: requested AFTER this epilogue's own operands (vmcnt counts in order: a wait for a younger
      // load would also wait for these), into the two buffers below the C image
      __builtin_amdgcn_sched_barrier(0);
      int nm0, nn0;
      tile_coords_id(p, next_tile, BM, BN, nm0, nn0);
      set_offsets(nm0, nn0);
      stage_tile(0, smem);
      if (nt > 1) stage_tile(1, smem + BUF);
      __builtin_amdgcn_sched_barrier(0);
    }
    // e4m3 operands: the per-token and per-output-channel scales of this lane's rows / column quads
    float f8r[AH * 4];
    f32x4 f8c[BH * 2];
    if constexpr (F8) {
#pragma unroll
      for (int i = 0; i < AH * 4; ++i) {
        int m = m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fr;
        m = m < p.M ? m : p.M - 1;
        f8r[i] = p.f8_rs ? p.f8_rs[m] : 1.f;
      }
#pragma unroll
      for (int q = 0; q < BH * 2; ++q) {
        const int li = (q >> 1) * 128 + wc * 32 + (q & 1) * 16 + 4 * fq;
        int n = n0 + li;
        if constexpr (EK == 4) n = li < BN / 2 ? (n0 >> 1) + li : (p.N >> 1) + (n0 >> 1) + li - BN / 2;       // (the weight row a tile column multiplies)
        n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);
        f8c[q] = p.f8_cs ? *reinterpret_cast<const f32x4*>(p.f8_cs + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
    }
    // fused q|k|v store: the rows' output offsets and RoPE table rows, one thread per row, in 4 KiB above the LayerNorm pairs (never
    // operand space); published by the barrier between staging and the row stores
    // ... and the whole cos / sin tables (33 rows x 64 B each at 448 x 448) in 8 KiB above that: the store's table reads were two
    // dependent global round trips per thread and tile with nothing to hide them behind
    const char* rowinfo = nullptr;
    const char* ropetab = nullptr;
    if constexpr (EK == 1 && AH + BH == 4) {
      if (p.epi == 1) {
        char* ri = smem + 2 * BUF + 2048;
        if (tid < BM) *reinterpret_cast<int4*>(ri + 16 * tid) = vit_row_info(p, m0 + tid);
        rowinfo = ri;
        if (p.vit.use_rope && p.vit.rope_rows <= 64) {
          char* tb = ri + 4096;
          const int nch = p.vit.rope_rows * 4, half = tid >> 8, ch = tid & 255;       // 16-byte chunks per table; threads 0-255: cos, 256-511: sin
          if (ch < nch)
            *reinterpret_cast<u32x4*>(tb + 4096 * half + 16 * ch) =
                *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(half ? p.vit.sin : p.vit.cos) + 16 * ch);
          ropetab = tb;
        }
      }
    }
    auto stage_all = [&](auto act_tag, auto mode_tag, auto ln_tag) {
      constexpr int ACT = decltype(act_tag)::value, MODE = decltype(mode_tag)::value;
      constexpr bool LN = LN_FUSED && decltype(ln_tag)::value;
#pragma unroll
      for (int i = 0; i < AH; ++i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float2 lv = float2{0.f, 1.f};
          if (LN) lv = lnp[i * 128 + wr * 64 + mt * 16 + fr];
#pragma unroll
          for (int j = 0; j < BH; ++j)
#pragma unroll
            for (int nt2 = 0; nt2 < 2; ++nt2) {
              f32x4 a = acc[i * 4 + mt][j * 2 + nt2];
              if (LN) a = ln_apply(a, lv.x, lv.y, cc[j * 2 + nt2]);
              if constexpr (F8) a = a * f8r[i * 4 + mt] * f8c[j * 2 + nt2];
              stage_quad<BN, ACT, MODE>(p, smem_c, i * 128 + wr * 64 + mt * 16 + fr, j * 128 + wc * 32 + nt2 * 16 + 4 * fq, a, bias_r[j][nt2], cs_r[j][nt2]);
            }
          if (LN_FUSED) __builtin_amdgcn_sched_barrier(0);      // one 16-row group at a time (register pressure)
        }
    };
    // activation, bias / LayerScale / alpha and the LayerNorm fold are dispatched ONCE per tile (gemm_common.h: VQ3_STAGE_DISPATCH); the
    // fused q|k|v and SwiGLU epilogues never carry an activation or LayerScale
#define V6_STAGE(ACT_, MODE_)                                                                                                     \
  do {                                                                                                                            \
    if constexpr (EK == 3) stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::true_type{}); \
    else if constexpr (EK == 1) {                                                                                                 \
      if (ln_on) stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::true_type{});          \
      else stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::false_type{});               \
    } else stage_all(std::integral_constant<int, ACT_>{}, std::integral_constant<int, MODE_>{}, std::false_type{});               \
  } while (0)
    if constexpr (EK == 2 || EK == 4) V6_STAGE(0, 0);            // (host: no bias, no LayerScale, alpha == 1)
    else if constexpr (EK == 1) {
      if (p.bias && p.alpha == 1.f) V6_STAGE(0, 1);
      else V6_STAGE(0, -1);
    } else VQ3_STAGE_DISPATCH(p, V6_STAGE);
#undef V6_STAGE
    V6_STAMP(6);    // this wave's quads are in the C image
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile's K tiles 0 and 1 has landed
    __syncthreads();
    V6_STAMP(3);    // C image staged (and the next tile's first operands landed)
    staged_store<BM, BN, (EK == 3 ? 0 : EK)>(p, smem_c, coff, roff, m0, n0, tid, 512, &pre, rowinfo, ropetab);     // (EK == 4: gate|up and act leave together)
    V6_STAMP(4);    // row stores issued
  } else {
    if (more) {
      int nm0, nn0;
      tile_coords_id(p, next_tile, BM, BN, nm0, nn0);
      set_offsets(nm0, nn0);
      stage_tile(0, smem);
      if (nt > 1) stage_tile(1, smem + BUF);
    }
#pragma unroll
    for (int i = 0; i < AH; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + i * 128 + wr * 64 + mt * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < BH; ++j)
#pragma unroll
          for (int nt2 = 0; nt2 < 2; ++nt2) {
            const int n = n0 + j * 128 + wc * 32 + nt2 * 16 + 4 * fq;
            if (n >= p.N) continue;
            store_quad<OUT_F32>(p, coff, roff, m, n, acc[i * 4 + mt][j * 2 + nt2]);
          }
      }
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  prefetched = more;
  if (p.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (diagnostic only: thread 0's own stores have been acknowledged)
    p.stamps[(long)tile * 8 + 5] = __builtin_amdgcn_s_memrealtime();
  }
  if (more) __syncthreads();      // everybody's DMA pieces are visible; the C image's readers are done before K tile 2 lands on it
  tile = next_tile;
  } while (PERSIST && tile < ntile);
}

// per-stream workspace of the last-round K split: f32 partial tiles + arrival counts (launches of one stream run in order)
struct SkWs { float* ws = nullptr; unsigned* cnt = nullptr; };
std::mutex g_sk_mutex;
std::map<hipStream_t, SkWs> g_sk;
// One error word per process in host-mapped pinned memory: a reducer whose bounded wait expired stores 1 there (system scope) and the
// host reads it WITHOUT synchronising anything. It is never part of the per-launch memset: a give-up inside a training step stays
// visible until somebody reads it with clear (gemm_split_poll(true) / vq3_gemm_split_status) - the trainer does, once per optimiser step.
unsigned* g_sk_err = nullptr;
unsigned g_sk_spin = 1u << 23;
vq3_ws_provider_t g_sk_provider = nullptr;      // caller-owned memory for the per-stream workspaces (vq3_gemm_workspace_provider)
constexpr size_t SK_WS_BYTES = (size_t)192 * 256 * 256 * 4;          // rem * (slices - 1) <= 192 partial tiles (48 MiB)
constexpr size_t SK_CNT_BYTES = SK_MAX_TILES * sizeof(unsigned);
bool sk_workspace(hipStream_t s, float** ws, unsigned** cnt, unsigned** err, unsigned* spin) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  SkWs& w = g_sk[s];
  if (!w.ws) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
    if (!g_sk_err) {
      void* h = nullptr;
      if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
      g_sk_err = (unsigned*)h;
      *(volatile unsigned*)g_sk_err = 0u;
    }
    bool ok = false;
    if (g_sk_provider) {
      // one block from the caller's allocator (kept alive by the caller for the life of the process): partial tiles, then the counts
      int dev = 0;
      (void)hipGetDevice(&dev);
      char* blk = (char*)g_sk_provider((int64_t)(SK_WS_BYTES + SK_CNT_BYTES), dev, 1);
      if (blk) { w.ws = (float*)blk; w.cnt = (unsigned*)(blk + SK_WS_BYTES); ok = true; }
    } else {
      // a plain C caller that registered no provider: the library's own allocation, once per stream
      ok = hipMalloc(&w.ws, SK_WS_BYTES) == hipSuccess && hipMalloc(&w.cnt, SK_CNT_BYTES) == hipSuccess;
      if (!ok) (void)hipGetLastError();
    }
    if (!ok) {
      static bool said = false;
      if (!said) fprintf(stderr, "[vq3 gemm] no memory for a split-K workspace (48 MiB): cfg 25 launches run as cfg 20 on this stream (same result, "
                                 "another order of the f32 sums)\n");
      said = true;
      w.ws = nullptr;
      return false;
    }
  }
  // the arrival counts are zeroed on the caller's stream before every launch that uses them (a memset node when captured); the error
  // word is NOT (it is sticky)
  static_assert(SK_CNT_BYTES % 16 == 0, "count block: multiple of 16 bytes");
  if (hipMemsetAsync(w.cnt, 0, SK_CNT_BYTES, s) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  *ws = w.ws; *cnt = w.cnt; *err = g_sk_err; *spin = g_sk_spin;
  return true;
}

// The split applies when the tile grid leaves a last round at most half full: R = tiles % CUs (or all tiles, when there are fewer than
// CUs) tiles are cut into S = min(CUs / R, 4) K slices of at least 8 K tiles each. Returns S (0: does not apply).
int sk_plan(const GemmParams& p, int nbatch, int ncu, int* full, int* rem, int ktile = BK6) {
  if (nbatch != 1 || p.out_f32 || (p.epi != 0 && p.epi != 1)) return 0;      // plain, LayerNorm-fold and fused q|k|v epilogues (the reducer runs them on the summed tile)
  static int off = -1;
  if (off < 0) { const char* e = getenv("VQ3_GEMM_SPLIT"); off = (e && atoi(e) == 0) ? 1 : 0; }      // VQ3_GEMM_SPLIT=0: A/B runs
  if (off) return 0;
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int r = tiles % ncu;
  if (r == 0 || r > ncu / 2 || r > SK_MAX_TILES) return 0;
  int sl = ncu / r;
  sl = sl > 4 ? 4 : sl;
  const int nt = p.K / ktile;
  if (sl > nt / 8) sl = nt / 8;
  while (sl >= 2 && (long)r * (sl - 1) > 192) --sl;
  // every slice non-empty: per = ceil(nt / sl) slices cover nt with the last one holding nt - (sl - 1) per >= 1 K tiles
  while (sl >= 2 && nt - (sl - 1) * ((nt + sl - 1) / sl) < 1) --sl;
  if (sl < 2) return 0;
  *full = tiles - r; *rem = r;
  return sl;
}

template <int AH, int BH>
int launch_v6(GemmParams& p, int nbatch, hipStream_t stream, bool split = false) {
  // (two-phase variants: two K-tile buffers of the NEXT tile + the C image of the current one, or three K-tile buffers: 160 KiB)
  constexpr int SMEM = (AH + BH == 4) ? 2 * 4 * HALF + 2048 + 4096 + 8192 : 2 * 3 * HALF + 128 * AH * 128 * BH * 2;   // (+ 2 KiB: LayerNorm pairs above the C image, + 4 KiB: the fused q|k|v store's row table, + 8 KiB: its cos / sin tables)
  static_assert(SMEM <= 160 * 1024 && (AH + BH == 4 || SMEM >= 3 * (AH + BH) * HALF), "LDS budget");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<AH, BH, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if constexpr (AH + BH == 4)
      if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if constexpr (AH + BH == 4)
      if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if constexpr (AH + BH == 4)
      if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vq3_set_error("gemm v6: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 2;
    }
    attr_done = true;
  }
  // bf16 output that cannot take the LDS-staged epilogue (rows not 16-byte aligned): the loader-ring kernel keeps the per-quad path
  if (p.epi == 3 && (p.out_f32 || !host_staged_ok(p) || (p.N >> 1) % (64 * BH) != 0)) {
    vq3_set_error("gemm v6: the SwiGLU-forward epilogue needs the staged bf16 path and I %% %d == 0", 64 * BH);
    return 1;
  }
  if (!p.out_f32 && !host_staged_ok(p)) return launch_gemm_v2(p, AH == 2 ? 11 : 13, nbatch, stream);
  p.mtiles = (p.M + 128 * AH - 1) / (128 * AH);
  p.ntiles = (p.N + 128 * BH - 1) / (128 * BH);
  choose_tile_order(p, 128 * AH, 128 * BH, 1);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    (void)hipGetLastError();
    ncu = n / 8 * 8;                 // persistent workgroups stride over the tile order by a multiple of 8: they stay on their XCD's slice
  }
  if (const char* sp = getenv("VQ3_GEMM_STAMP_PTR")) p.stamps = (unsigned long long*)strtoull(sp, nullptr, 0);   // diagnostics only
  static int stagger_env = -1;
  if (stagger_env < 0) { const char* e = getenv("VQ3_V6_STAGGER"); stagger_env = e ? atoi(e) : 0; }
  p.stagger = stagger_env;
  int nwg = p.mtiles * p.ntiles;
  if constexpr (AH + BH == 4) {
    if (split) {
      if (!host_staged_ok(p)) return -1;
      int full = 0, rem = 0;
      const int sl = sk_plan(p, nbatch, ncu, &full, &rem);
      if (sl < 2 || !sk_workspace(stream, &p.sk_ws, &p.sk_cnt, &p.sk_err, &p.sk_spin)) return -1;
      p.sk_full = full; p.sk_rem = rem; p.sk_s = sl;
      const dim3 sgrid(full + ((rem + 7) & ~7) * sl, 1, 1);
      if (p.epi == 1) hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 1, true>), sgrid, dim3(512), SMEM, stream, p);
      else if (p.ln_in) hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 3, true>), sgrid, dim3(512), SMEM, stream, p);
      else hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 0, true>), sgrid, dim3(512), SMEM, stream, p);
      return 0;
    }
  }
  if (AH + BH < 4 && nwg > ncu && getenv("VQ3_V6_PERSIST") == nullptr) nwg = ncu;      // (VQ3_V6_PERSIST=0: one tile per workgroup, for A/B runs)
  else if (AH + BH < 4 && nwg > ncu && atoi(getenv("VQ3_V6_PERSIST")) != 0) nwg = ncu;
  dim3 grid(nwg, 1, nbatch);
  if (p.out_f32)
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, true, 0>), grid, dim3(512), SMEM, stream, p);
  else if (p.epi == 1)
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, false, 1>), grid, dim3(512), SMEM, stream, p);
  else if (p.epi == 2)
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, false, 2>), grid, dim3(512), SMEM, stream, p);
  else if (p.epi == 3)
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, false, 4>), grid, dim3(512), SMEM, stream, p);
  else if (p.ln_in)
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, false, 3>), grid, dim3(512), SMEM, stream, p);
  else
    hipLaunchKernelGGL((gemm_v6_kernel<AH, BH, false, 0>), grid, dim3(512), SMEM, stream, p);
  return 0;
}

}  // namespace

// e4m3 operands on the 256 x 256 kernel (config C5): plain / SwiGLU-forward / SwiGLU-backward epilogues, last round split along K where
// that applies. Returns -1 (nothing launched) outside the contract: the caller takes the loader-ring kernel of gemm_fp8.hip.
int launch_gemm_v6_f8(GemmParams& p, hipStream_t stream) {
  constexpr int SMEM = 2 * 4 * HALF + 2048;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      vq3_set_error("gemm v6 (e4m3): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  if (p.out_f32 || !host_staged_ok(p) || p.K % 128 != 0 || p.ln_in || p.st_out) return -1;
  if (p.epi == 3 && (p.N >> 1) % 128 != 0) return -1;
  if (p.epi == 1) return -1;
  p.mtiles = (p.M + 255) / 256;
  p.ntiles = (p.N + 255) / 256;
  choose_tile_order(p, 256, 256, 1);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    (void)hipGetLastError();
    ncu = n / 8 * 8;
  }
  p.stagger = 0;
  const dim3 grid(p.mtiles * p.ntiles, 1, 1);
  if (p.epi == 2) hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 2, false, true>), grid, dim3(512), SMEM, stream, p);
  else if (p.epi == 3) hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 4, false, true>), grid, dim3(512), SMEM, stream, p);
  else {
    int full = 0, rem = 0;
    const int sl = sk_plan(p, 1, ncu, &full, &rem, 128);
    if (sl >= 2 && sk_workspace(stream, &p.sk_ws, &p.sk_cnt, &p.sk_err, &p.sk_spin)) {
      p.sk_full = full; p.sk_rem = rem; p.sk_s = sl;
      hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 0, true, true>), dim3(full + ((rem + 7) & ~7) * sl, 1, 1), dim3(512), SMEM, stream, p);
    } else {
      hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 0, false, true>), grid, dim3(512), SMEM, stream, p);
    }
  }
  return 0;
}

// both operands k-major on the 256 x 256 kernel (the weight-gradient products; cfg 106, with the last round split: 107). -1 = outside
// its contract (nothing launched): the caller keeps gemm3.hip's kernels.
int launch_gemm_v6_km(GemmParams& p, int nbatch, hipStream_t stream, bool split) {
  constexpr int SMEM = 2 * 4 * HALF + 2048;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_v6_kernel<2, 2, false, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      vq3_set_error("gemm v6 (k-major): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  if (nbatch != 1 || p.out_f32 || p.epi != 0 || p.ln_in || p.nsplit != 1 || !host_staged_ok(p) || p.M % 8 || p.N % 8 || p.K % 8 || p.M < 8 || p.N < 8)
    return -1;
  // (byte offsets of a tile's k rows are 32-bit: 63 rows of either operand must stay below 4 GiB)
  if ((long)64 * p.lda * 2 + (long)p.M * 2 >= (1l << 32) || (long)64 * p.ldb * 2 + (long)p.N * 2 >= (1l << 32)) return -1;
  p.mtiles = (p.M + 255) / 256;
  p.ntiles = (p.N + 255) / 256;
  choose_tile_order(p, 256, 256, 1);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    (void)hipGetLastError();
    ncu = n / 8 * 8;
  }
  p.stagger = 0;
  // (the last-round K split is not offered here: with the k-major fragment addresses on top of the split's state the 256-register kernel
  // spills 133 registers, 26 scratch instructions per main-loop trip - built and measured once: the down-projection weight gradient,
  // 380 tiles = 256 + 124 x 2 halves, 1871 us against 571 unsplit)
  if (split) return -1;
  hipLaunchKernelGGL((gemm_v6_kernel<2, 2, false, 0, false, false, true>), dim3(p.mtiles * p.ntiles, 1, 1), dim3(512), SMEM, stream, p);
  return 0;
}

int gemm_split_plan(const GemmParams& p, int nbatch, int ncu, int* full, int* rem) { return sk_plan(p, nbatch, ncu, full, rem); }
// Host read of the sticky error word: no synchronisation (it says what the launches COMPLETED so far reported). clear: reset it.
int gemm_split_poll(bool clear) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  if (!g_sk_err) return 0;
  const unsigned v = __atomic_load_n(g_sk_err, __ATOMIC_ACQUIRE);
  if (v && clear) __atomic_store_n(g_sk_err, 0u, __ATOMIC_RELEASE);
  return v ? 1 : 0;
}
void gemm_split_set_provider(vq3_ws_provider_t fn) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  g_sk_provider = fn;
}
void gemm_split_set_spin_bound(unsigned polls) {
  std::lock_guard<std::mutex> lock(g_sk_mutex);
  g_sk_spin = polls;
}
// Synchronises `stream`, then reads AND clears the word: 1 = some split launch of this process (any stream) gave up since the last read.
int gemm_split_gave_up(hipStream_t stream) {
  if (hipStreamSynchronize(stream) != hipSuccess) {
    (void)hipGetLastError();
    return -2;
  }
  return gemm_split_poll(true);
}

// shape: 0 = 256 x 256, 1 = 256 x 128, 2 = 128 x 256, 3 = 256 x 256 with the last round split along K (-1 where that does not apply)
int launch_gemm_v6(GemmParams& p, int shape, int nbatch, hipStream_t stream) {
  if (shape == 3) return launch_v6<2, 2>(p, nbatch, stream, true);
  if (shape == 1) return launch_v6<2, 1>(p, nbatch, stream);
  if (shape == 2) return launch_v6<1, 2>(p, nbatch, stream);
  return launch_v6<2, 2>(p, nbatch, stream);
}

}  // namespace vq3gemm
