// VGGT aggregator kernels (multi-view ViT): patch im2col with fused ImageNet normalisation, per-head q/k
// LayerNorm + 2-D RoPE + head split, and a flash-style attention forward (head_dim 64, non-causal) with LDS-staged,
// XOR-swizzled K / V^T tiles and v_mfma_f32_32x32x16_bf16 chains kept in registers:
//   S^T = K . Q^T        (A = K tile rows, B = Q fragment held in VGPRs for the whole kernel)
//   O^T += V^T . P^T     (the S^T accumulator, exponentiated and packed to bf16, IS the B operand: no LDS round trip)
// so every softmax statistic of a query row lives on the lane that owns that query column.
#include <cstdlib>

#include <type_traits>

#include "common.h"
#include "vq3_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ im2col
// images f32 [NI, 3, H, W] in [0,1]; out bf16 [NI*Hp*Wp, Kp], k = c*p*p + ky*p + kx, columns >= 3*p*p zero.
// The reference casts images to bf16 first and normalises in bf16 (two rounded ops).
__global__ __launch_bounds__(256) void im2col_norm_kernel(const float* __restrict__ img, bf16_t* __restrict__ out,
                                                          int H, int W, int p, int Hp, int Wp, int Kp, float m0,
                                                          float m1, float m2, float s0, float s1, float s2) {
  const long patch = blockIdx.x;  // ni*Hp*Wp + py*Wp + px
  const int px = (int)(patch % Wp), py = (int)((patch / Wp) % Hp);
  const long ni = patch / ((long)Wp * Hp);
  const int K = 3 * p * p;
  bf16_t* o = out + patch * (long)Kp;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < K) {
      const int c = k / (p * p), r = k - c * p * p, ky = r / p, kx = r - ky * p;
      const float x = rbf(img[((ni * 3 + c) * H + (py * p + ky)) * (long)W + (px * p + kx)]);
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
      const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      v = rbf(rbf(x - rbf(mean)) / rbf(sd));
    }
    o[k] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------ q/k prep
// qkv bf16 [T, 3*C] (columns: q heads | k heads | v heads, head-major) -> Q,K,V bf16 [G, NH, N, 64], T = G*N.
// Optional LayerNorm over the 64 head features (qk_norm) and 2-D RoPE: the first 32 features rotate with the
// patch row (y), the last 32 with the patch column (x), rotate-half inside each 32-block; tokens < patch_start of a
// frame sit at position 0 (identity). cos/sin tables bf16 [maxpos+1, 32].
__global__ __launch_bounds__(256) void vit_qkprep_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ qn_w,
                                                         const float* __restrict__ qn_b, const float* __restrict__ kn_w,
                                                         const float* __restrict__ kn_b, const bf16_t* __restrict__ cs,
                                                         const bf16_t* __restrict__ sn, bf16_t* __restrict__ Q,
                                                         bf16_t* __restrict__ K, bf16_t* __restrict__ V, int N, int NH,
                                                         int P, int patch_start, int Wp, int use_norm, int use_rope,
                                                         float eps) {
  // A wave handles two heads per pass (lanes 0-31 / 32-63), a lane two adjacent features (one 4-byte access):
  // the LayerNorm sums stay inside a 32-lane half, the rotate-half partner of feature e is e^16 = lane^8.
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const long t = blockIdx.x;
  const long g = t / N;
  const int n = (int)(t - g * N);
  const int C = NH * 64;
  const bf16_t* row = qkv + t * 3L * C;
  int py = 0, px = 0;
  if (use_rope) {
    const int tp = n % P;
    if (tp >= patch_start) {
      py = (tp - patch_start) / Wp + 1;
      px = (tp - patch_start) % Wp + 1;
    }
  }
  const int e0 = 2 * l32;                       // features e0, e0+1 of the head
  const int pos = l32 < 16 ? py : px;
  const int f0 = e0 & 31;
  float c0 = 1.f, c1 = 1.f, s0 = 0.f, s1 = 0.f;
  if (use_rope) {
    c0 = bf2f(cs[pos * 32 + f0]); c1 = bf2f(cs[pos * 32 + f0 + 1]);
    s0 = bf2f(sn[pos * 32 + f0]); s1 = bf2f(sn[pos * 32 + f0 + 1]);
  }
  const bool neg = (l32 & 8) == 0;              // (e & 16) == 0 -> rotate-half takes -x[e+16]
  for (int hp = wid; hp < (3 * NH) / 2; hp += 4) {
    const int hh = 2 * hp + half;
    const int which = hh / NH, h = hh - which * NH;
    const uint32_t raw = *reinterpret_cast<const uint32_t*>(row + which * C + h * 64 + e0);
    float x0 = bf2f((bf16_t)(raw & 0xffff)), x1 = bf2f((bf16_t)(raw >> 16));
    bf16_t* dst = (which == 0 ? Q : (which == 1 ? K : V)) + ((g * NH + h) * (long)N + n) * 64 + e0;
    if (which < 2) {
      if (use_norm) {
        float sum = x0 + x1;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.f / 64.f);
        const float d0 = x0 - mean, d1 = x1 - mean;
        float sq = d0 * d0 + d1 * d1;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rs = rsqrtf(sq * (1.f / 64.f) + eps);
        const float* w = which == 0 ? qn_w : kn_w;
        const float* b = which == 0 ? qn_b : kn_b;
        x0 = rbf(d0 * rs * w[e0] + b[e0]);
        x1 = rbf(d1 * rs * w[e0 + 1] + b[e0 + 1]);
      }
      if (use_rope) {
        const float p0 = __shfl_xor(x0, 8, 64), p1 = __shfl_xor(x1, 8, 64);
        const float r0 = neg ? -p0 : p0, r1 = neg ? -p1 : p1;
        x0 = rbf(rbf(x0 * c0) + rbf(r0 * s0));
        x1 = rbf(rbf(x1 * c1) + rbf(r1 * s1));
      }
    }
    *reinterpret_cast<uint32_t*>(dst) = pack2bf(x0, x1);
  }
}

// Same contract with 4 features (8 bytes) per lane: 16 lanes per head, four heads per wave pass - half the memory
// instructions of the 2-feature form. Needs NH % 4 == 0. LayerNorm sums stay inside a 16-lane quarter, the rotate-half
// partner of feature e is e ^ 16 = lane ^ 4 (same element slot).
__global__ __launch_bounds__(256) void vit_qkprep4_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ qn_w,
                                                          const float* __restrict__ qn_b, const float* __restrict__ kn_w,
                                                          const float* __restrict__ kn_b, const bf16_t* __restrict__ cs,
                                                          const bf16_t* __restrict__ sn, bf16_t* __restrict__ Q,
                                                          bf16_t* __restrict__ K, bf16_t* __restrict__ V, int N, int NH,
                                                          int P, int patch_start, int Wp, int use_norm, int use_rope,
                                                          float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l16 = lane & 15, quarter = lane >> 4;
  const long t = blockIdx.x;
  const long g = t / N;
  const int n = (int)(t - g * N);
  const int C = NH * 64;
  const bf16_t* row = qkv + t * 3L * C;
  int py = 0, px = 0;
  if (use_rope) {
    const int tp = n % P;
    if (tp >= patch_start) {
      py = (tp - patch_start) / Wp + 1;
      px = (tp - patch_start) % Wp + 1;
    }
  }
  const int e0 = 4 * l16;                       // features e0 .. e0+3 of the head
  const int pos = l16 < 8 ? py : px;
  const int f0 = e0 & 31;
  float c[4] = {1.f, 1.f, 1.f, 1.f}, sv[4] = {0.f, 0.f, 0.f, 0.f};
  if (use_rope) {
    const u32x2 cr = *reinterpret_cast<const u32x2*>(cs + pos * 32 + f0);
    const u32x2 sr = *reinterpret_cast<const u32x2*>(sn + pos * 32 + f0);
    c[0] = bf2f((bf16_t)(cr[0] & 0xffff)); c[1] = bf2f((bf16_t)(cr[0] >> 16));
    c[2] = bf2f((bf16_t)(cr[1] & 0xffff)); c[3] = bf2f((bf16_t)(cr[1] >> 16));
    sv[0] = bf2f((bf16_t)(sr[0] & 0xffff)); sv[1] = bf2f((bf16_t)(sr[0] >> 16));
    sv[2] = bf2f((bf16_t)(sr[1] & 0xffff)); sv[3] = bf2f((bf16_t)(sr[1] >> 16));
  }
  const bool neg = (l16 & 4) == 0;              // (e & 16) == 0 -> rotate-half takes -x[e+16]
  for (int hq = wid; hq < (3 * NH) / 4; hq += 4) {
    const int hh = 4 * hq + quarter;
    const int which = hh / NH, h = hh - which * NH;
    const u32x2 raw = *reinterpret_cast<const u32x2*>(row + which * C + h * 64 + e0);
    float x[4] = {bf2f((bf16_t)(raw[0] & 0xffff)), bf2f((bf16_t)(raw[0] >> 16)), bf2f((bf16_t)(raw[1] & 0xffff)),
                  bf2f((bf16_t)(raw[1] >> 16))};
    bf16_t* dst = (which == 0 ? Q : (which == 1 ? K : V)) + ((g * NH + h) * (long)N + n) * 64 + e0;
    if (which < 2) {
      if (use_norm) {
        float sum = (x[0] + x[1]) + (x[2] + x[3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.f / 64.f);
        float d[4], sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[j] = x[j] - mean; sq = fmaf(d[j], d[j], sq); }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rs = rsqrtf(sq * (1.f / 64.f) + eps);
        const f32x4 w = *reinterpret_cast<const f32x4*>((which == 0 ? qn_w : kn_w) + e0);
        const f32x4 b = *reinterpret_cast<const f32x4*>((which == 0 ? qn_b : kn_b) + e0);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = rbf(d[j] * rs * w[j] + b[j]);
      }
      if (use_rope) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pj = __shfl_xor(x[j], 4, 64);
          const float rj = neg ? -pj : pj;
          x[j] = rbf(rbf(x[j] * c[j]) + rbf(rj * sv[j]));
        }
      }
    }
    u32x2 o;
    o[0] = pack2bf(x[0], x[1]);
    o[1] = pack2bf(x[2], x[3]);
    *reinterpret_cast<u32x2*>(dst) = o;
  }
}

// ------------------------------------------------------------------------------------------------ flash attention
// Q, K, V bf16 [NB, N, 64] as the q/k-prep kernel writes them (no transposed copy of V exists); O bf16 token-major:
// O[(g*N + q) * ldo + h*64 + d] with NB = G*NH, g = nb / NH, h = nb % NH.
//
// Workgroup = 4 waves x QB 32-query blocks; K/V tiles of 64 keys, register-staged (loads issued before the tile's MFMAs,
// LDS stores after them), two LDS stages, one barrier per tile. Per wave and tile:
//   S^T = K . Q'^T - m        32x32x16 MFMA, key on the row, QUERY ON THE LANE. Q' = Q * (scale * log2 e), rounded to bf16
//                             once per workgroup, and the accumulator starts at -m (the lane's running max), so the
//                             exponentiation is ONE v_exp_f32 per score: no multiply, no subtraction.
//   rel = max_k S^T           how far this tile's scores exceed the running max (lane-local + one cross-half exchange).
//   P^T = exp2(S^T)           in the STALE scale m; packed to bf16 it is the B operand of the next product as it stands
//                             (accumulator-as-operand: k order 16s + 8(j>>2) + 4h + (j&3), matched on the V side).
//   O^T += V^T . P^T          the V^T fragment (A operand) comes from the ROW-MAJOR V tile through ds_read_b64_tr_b16:
//                             lane group (h, g16) reads keys 16u + 4h .. +3 (and +8) x d-columns 32db + 16 g16 .. +15.
//   deferred rescale          only when some lane's rel exceeds THR (2^THR of head-room in P, bf16 keeps f32's exponent)
//                             are O, l multiplied by 2^-rel and m advanced; tile 0 and rel > 100 (overflow guard) take
//                             the textbook order (rescale first, subtract, then exponentiate).
// LDS images: K rows of 128 B with 16-byte chunks XOR (row >> 1) & 7 (ds_read_b128, lane = key row: conflict-free);
// V rows of 128 B with chunk bit 2 XOR (row >> 1) & 1, so the 4 key rows x 64 B a 32-lane half reads cover all 64 banks.
typedef __bf16 fa_bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 fa_bf8 __attribute__((ext_vector_type(8)));
typedef float fa_f32x8 __attribute__((ext_vector_type(8)));
constexpr int FA_KV = 64;
constexpr float FA_THR = 6.0f;
#ifndef FA_LSUM_MFMA
#define FA_LSUM_MFMA 1
#endif
#ifndef FA_QK_OVERLAP
#define FA_QK_OVERLAP 1
#endif
#ifndef FA_PRIO
#define FA_PRIO 0      // 1: s_setprio 1 around the MFMA bursts (experiment: make EXTRA=-DFA_PRIO=1)
#endif
#if FA_PRIO
#define FA_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define FA_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define FA_PRIO_HI() ((void)0)
#define FA_PRIO_LO() ((void)0)
#endif
#ifndef FA_DMA
#define FA_DMA 1       // K / V tiles by LDS-DMA (0: through staging registers, round 2-4's form)
#endif
#ifndef FA_LSUM_VAR1
#define FA_LSUM_VAR1 0 // 1: row sums on the matrix pipe in the VAR 1 kernel too (spills 8 registers even with the DMA staging: off)
#endif
#ifndef FA_X
#define FA_X 0         // diagnostic builds only (wrong results): 1 the exponentials become multiplies, 2 no row sums, 3 no growth scan, 4 no barrier between key tiles
#endif
#if FA_X == 1
#define FA_EXP2(x) ((x) * 1.0009765625f)
#else
#define FA_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
#ifndef FA_MIXMODE
#define FA_MIXMODE 2   // P.V sections, scheduler hint: 2 = "the MFMAs of a chunk first, then the next chunk's VALU work"; 0 = "one MFMA, six VALU, ..." (2.5 % slower)
#endif
#ifndef FA_NOMAX_LSUM
#define FA_NOMAX_LSUM 0 // the kernels without a running maximum, two query blocks per wave: row sums on the matrix pipe (0: packed VALU adds)
#endif
#ifndef FA_DOT2SUM
#define FA_DOT2SUM 1   // VALU row sums as v_dot2c_f32_bf16 on the PACKED P words the P.V MFMA reads (one issue slot per two values, and the
#endif                 // sums are those of the bf16-rounded weights the numerator uses); 0: packed f32 adds in tree form at the exponentials
#ifndef FA_QK_HINT
#define FA_QK_HINT 0
#endif

__device__ __forceinline__ int fa_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int fa_vswz(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

// ---- the ragged last rows of a sequence (<= 16 queries past the last full 256-row block; N = 1029 = 5 special tokens + 1024 patches
// leaves 5) - keys split over the four waves. As a fifth four-wave block those rows held a workgroup slot for a whole key loop, as a
// one-wave workgroup (flash_attn_hd64_kernel<1, 64>) they walked 17 key tiles one dependent round trip after the other (81 us beside
// a 289 us main launch at 48 x 16 pairs; tools/bench_flash.py: 384 us together against 330-340 us without the tail rows). As a launch
// of its own this body takes 35 us - what reading every pair's K and V once more costs (202 MB at 5.8 TB/s) - so it runs as ONE MORE
// workgroup per pair of the main launch (flash_attn_hd64_kernel, tail_begin), dispatched behind the pair's exact blocks while their
// K / V rows are in L2. tools/bench_flash_tail.py (40 launches each behind a 320 MB flush, three alternating rounds, median / best us):
// in-launch workgroup 326-341 / 307, side-stream launch 335-341 / 313-319, fifth four-wave block 340-342 / 320-323. The main kernel has
// no registers for a third query block of its own (252 of 256), which is what would remove the second pass over K / V. Here wave w
// takes the 32-key tiles w, w + 4, ... on its own: K fragments straight from global memory (a 16-key S^T tile's A operand is 16 whole
// 128-byte rows), V through a wave-private LDS tile read back transposed (ds_read_b64_tr_b16), no workgroup barrier inside the loop,
// the next tile's rows requested before this one is multiplied; textbook online softmax on 16-query tiles (v_mfma_f32_16x16x32_bf16,
// query on the lane as in the main kernel: S^T = K . Q'^T, O^T += V^T . P^T), and one merge of the four (max, sum, O^T) partials
// through LDS at the end. Q' = bf16(Q * scale * log2 e) and exp2 as in the main kernel.
constexpr int FT_TK = 32;                 // keys per step and wave
constexpr int FT_VP = 128 + 32;           // V row pitch in LDS (bytes): the 4 rows x 32 B of a transposed read fall into different banks
constexpr int FT_STAGE = FT_TK * FT_VP;   // 5 KiB per wave: ONE stage - a tile's rows are written after the previous tile's transposed reads were
                                          // issued, and a wave's LDS operations complete in the order they were issued
typedef short ft_s16x4 __attribute__((ext_vector_type(4)));

constexpr int FT_SMEM = 4 * FT_STAGE > 4 * 4 * 64 * 16 + 512 ? 4 * FT_STAGE : 4 * 4 * 64 * 16 + 512;      // staging, then the merge area (16.5 KiB)
__device__ __forceinline__ void flash_tail_body(char* smem, const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int N, int NH, long ldo,
                                                float scale_log2e, int q_begin, int o_rows, long nb) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const bf16_t* Qb = Q + nb * (long)N * 64;
  const bf16_t* Kb = K + nb * (long)N * 64;
  const bf16_t* Vb = V + nb * (long)N * 64;
  const int nq = N - q_begin;                                   // 1 .. 16 valid query rows
  // Q' fragments (B operand): lane (fr, g) holds Q'[q_begin + fr][32 s + 8 g .. + 8]
  bf16x8 qf[2];
  {
    const int qr = q_begin + (fr < nq ? fr : nq - 1);
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const bf16x8 raw = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * 64 + 32 * sidx + 8 * g);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[sidx][j] = (short)f2bf(bf2f((bf16_t)raw[j]) * scale_log2e);
    }
  }
  char* const mine = smem + wid * FT_STAGE;
  const int nt = (N + FT_TK - 1) / FT_TK;
  // staging of this wave's V tile: 32 rows x 8 chunks of 16 B = 4 per lane (row = lane / 8 + 8 i, chunk = lane % 8);
  // K fragments: [tile half][k step]: lane (fr, g) holds K[32 t + 16 half + fr][32 s + 8 g .. + 8]. Rows past N: zeros.
  // A wave's tiles are one dependent chain (load -> scores -> max -> P -> P.V), and with one tile requested ahead every step waited out
  // most of a memory round trip (9 tiles in ~25 us): THREE tiles are in flight, in a ring of register slots (unrolled by 3).
  constexpr int FT_DEPTH = 3;
  u32x4 vreg[FT_DEPTH][4];
  bf16x8 kf[FT_DEPTH][2][2];
  auto load_tile = [&](int t, u32x4 (&vr)[4], bf16x8 (&kr)[2][2]) {
    const int k0 = t * FT_TK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = k0 + (lane >> 3) + 8 * i;
      vr[i] = u32x4{0u, 0u, 0u, 0u};
      if (row < N) vr[i] = *reinterpret_cast<const u32x4*>(Vb + (long)row * 64 + (lane & 7) * 8);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int row = k0 + 16 * hf + fr;
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        kr[hf][sidx] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (row < N) kr[hf][sidx] = *reinterpret_cast<const bf16x8*>(Kb + (long)row * 64 + 32 * sidx + 8 * g);
      }
    }
  };
  f32x4 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;                         // m: the same in the four lanes of a query; l: this lane's keys only
  auto step = [&](int t, u32x4 (&vr)[4], bf16x8 (&kr)[2][2]) {
    // this tile's V rows into the wave's LDS tile (behind the previous tile's transposed reads in the wave's LDS queue: a wave's LDS
    // operations complete in issue order), its scores from the K fragments, then the slot is refilled with tile t + 12
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(mine + ((lane >> 3) + 8 * i) * FT_VP + (lane & 7) * 16) = vr[i];
    asm volatile("" ::: "memory");
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[0][sidx], qf[sidx], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[1][sidx], qf[sidx], s1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 4 * FT_DEPTH < nt) load_tile(t + 4 * FT_DEPTH, vr, kr);
    const int k0 = t * FT_TK + 4 * g;                            // lane's keys: k0 + r (tile half 0), k0 + 16 + r (half 1)
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (k0 + r >= N) s0[r] = -INFINITY;
      if (k0 + 16 + r >= N) s1[r] = -INFINITY;
      tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m_run, tmax);                         // finite: every tile of the loop holds at least one key < N
    const float alpha = __builtin_amdgcn_exp2f(m_run - mn);      // first tile: exp2(-inf) = 0 on O = l = 0
    m_run = mn;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(s0[r] - mn);
      s1[r] = __builtin_amdgcn_exp2f(s1[r] - mn);
      ps += s0[r] + s1[r];
    }
    l_run = l_run * alpha + ps;
    const u32x4 pw = {pack2bf(s0[0], s0[1]), pack2bf(s0[2], s0[3]), pack2bf(s1[0], s1[1]), pack2bf(s1[2], s1[3])};
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);           // contraction slot 8 g + j = key 4 g + j (half 0), 16 + 4 g + j - 4 (half 1)
    // V^T fragments in the same key order: lane fr of group g addresses row 4 g + (fr >> 2), 8-byte piece fr & 3 of a 4-row x 32-byte
    // block and receives column fr of it
    const char* vb = mine + (4 * g + (fr >> 2)) * FT_VP + (fr & 3) * 8;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const ft_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ft_s16x4*)(vb + d * 32));
      const ft_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ft_s16x4*)(vb + 16 * FT_VP + d * 32));
      const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[d][r] *= alpha;
      oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[d], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int j = 0; j < FT_DEPTH; ++j)
    if (wid + 4 * j < nt) load_tile(wid + 4 * j, vreg[j], kf[j]);
  for (int t = wid; t < nt; t += 4 * FT_DEPTH) {
    step(t, vreg[0], kf[0]);
    if (t + 4 < nt) step(t + 4, vreg[1], kf[1]);
    if (t + 8 < nt) step(t + 8, vreg[2], kf[2]);
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  // ---- merge the four waves' partials: O area [wave][d tile][lane] f32x4, then (m, l) per [wave][query]
  __syncthreads();
  f32x4* oarea = reinterpret_cast<f32x4*>(smem);
  float* marea = reinterpret_cast<float*>(smem + 4 * 4 * 64 * 16);
#pragma unroll
  for (int d = 0; d < 4; ++d) oarea[(wid * 4 + d) * 64 + lane] = oacc[d];
  if (g == 0) { marea[wid * 16 + fr] = m_run; marea[64 + wid * 16 + fr] = l_run; }
  __syncthreads();
  float mw[4], M = -INFINITY;
#pragma unroll
  for (int v = 0; v < 4; ++v) { mw[v] = marea[v * 16 + fr]; M = fmaxf(M, mw[v]); }
  float L = 0.f, wgt[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    wgt[v] = (mw[v] > -INFINITY) ? __builtin_amdgcn_exp2f(mw[v] - M) : 0.f;      // a wave without a tile: no contribution
    L += marea[64 + v * 16 + fr] * wgt[v];
  }
  f32x4 o = {0.f, 0.f, 0.f, 0.f};                                // wave w finishes depth columns 16 w + 4 g .. + 3 of query fr
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const f32x4 x = oarea[(v * 4 + wid) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += x[r] * wgt[v];
  }
  if (fr < nq) {
    const float inv = 1.f / L;
    const long gi = nb / NH;
    const int hd = (int)(nb % NH);
    bf16_t* orow = O + (gi * o_rows + q_begin + fr) * ldo + hd * 64 + 16 * wid + 4 * g;
    *reinterpret_cast<u32x2*>(orow) = u32x2{pack2bf(o[0] * inv, o[1] * inv), pack2bf(o[2] * inv, o[3] * inv)};
  }
}

__global__ __launch_bounds__(256) void flash_tail_hd64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int N, int NH, long ldo,
                                                               float scale_log2e, int q_begin, int o_rows) {
  __shared__ __attribute__((aligned(16))) char smem[FT_SMEM];
  flash_tail_body(smem, Q, K, V, O, N, NH, ldo, scale_log2e, q_begin, o_rows, (long)blockIdx.x);
}

// NT threads per workgroup: 256 (four waves, 128 * QB query rows) or 64 - ONE wave for the ragged last rows of a sequence whose
// length is a multiple of 256 plus a few (1029 = 5 special tokens + 1024 patches): as a fifth four-wave block those 5 rows held a
// workgroup slot (222 VGPRs x 4 waves, 32 KiB LDS) for a whole key loop in every (sample, head) pair; as a one-wave workgroup on a
// second stream they run beside the 4 exact blocks. Query rows q_begin <= q < q_end are processed, keys 0 .. N - 1.
// VAR (two query blocks per wave only): 0 = row sums of P on the matrix pipe; 1 = keys 32-63 multiplied while keys 0-31 are scanned and
// exponentiated (row sums on the VALU: both together do not fit 256 registers)
// NOMAX (the caller vouches for |score| <= FA_NOMAX_BOUND in the exponent's units, e.g. from the weights of a q / k LayerNorm in front of
// the attention): softmax is shift-invariant and bf16 / f32 keep their relative precision at any scale, so with every 2^score inside
// 2^+-90 the running maximum is not needed at all - no -m fill of the score accumulators (the first K.Q MFMA starts from the inline
// constant 0), no growth scan, no deferred rescale, no textbook-order branch: per 64-key tile and wave 32 v_mov + 32 v_max3 + the
// decision go away beside 64 v_exp_f32 (the VALU is this kernel's busier pipe). Sums stay below 2^90 * N, far from the f32 range.
template <int QB, int NT, int VAR = 0, bool NOMAX = false>
__global__ __launch_bounds__(NT, 2) void flash_attn_hd64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                                 int N, int NH, long ldo, float scale_log2e, int q_begin, int q_end, int o_rows,
                                                                 int tail_begin, int nblk) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * FA_KV * 128];   // 2 stages x (K tile | V tile)
  static_assert(FT_SMEM <= 2 * 2 * FA_KV * 128, "the tail rows' staging / merge area fits the key loop's LDS");
  // XCD-aware placement (nblk > 0: a 1-D grid of nblk * pairs workgroups). The hardware deals consecutive workgroup ids round-robin over
  // the 8 XCDs, each with a private 4 MiB L2: with (query block, pair) = (blockIdx.x, blockIdx.y) the nblk blocks of ONE (sample, head)
  // pair - which all walk the same K and V rows - ran on nblk different XCDs, every one fetching the pair's K / V through the fabric on
  // its own (N = 1029: 5 x 263 KB per pair, 1.26 GB per 960-pair launch for 0.25 GB of distinct rows). Here workgroup w is the
  // (w / 8)-th item of XCD w % 8's contiguous slice of the (pair-major) item list, so a pair's blocks are dispatched together on one
  // XCD and share its L2. Bijective for any grid size (same construction as gemm_common.h: tile_coords_id); only speed depends on it.
  int bx = blockIdx.x, by = blockIdx.y, nbx = gridDim.x;
  if (nblk > 0) {
    const int total = gridDim.x, w = blockIdx.x;
    const int xcd = w & 7, idx = w >> 3, qd = total >> 3, rm = total & 7;
    const int sidx = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    by = sidx / nblk;
    bx = sidx - by * nblk;
    nbx = nblk;
  }
  // tail_begin >= 0: the LAST workgroup of every (sample, head) pair takes the <= 16 ragged rows tail_begin .. N - 1 with its keys split
  // over the four waves (flash_tail_body). It is dispatched right behind the pair's exact blocks, so its K / V rows come from the L2 they
  // are filling - as a launch of its own the tail rows cost one more read of every K and V from HBM (35 us at 48 x 16 pairs)
  if constexpr (NT == 256 && QB == 2) {      // (the host only asks for it with two query blocks per wave; kept out of the 168-register QB = 1 build)
    if (tail_begin >= 0 && bx == nbx - 1) {
      flash_tail_body(smem, Q, K, V, O, N, NH, ldo, scale_log2e, tail_begin, o_rows, (long)by);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const long nb = by;
  const int q0 = q_begin + bx * (NT / 2 * QB) + wid * (32 * QB);
  const bf16_t* Qb = Q + nb * (long)N * 64;
  const bf16_t* Kb = K + nb * (long)N * 64;
  const bf16_t* Vb = V + nb * (long)N * 64;

  f32x16 o0[QB], o1[QB];
  float m_run[QB], l_run[QB];
  // -m replicated over an accumulator-shaped vector: the first QK MFMA of a tile takes it as C and writes the scores to
  // other registers, so no per-tile fill of the accumulator is needed (m only changes on the rare deferred rescale).
  // One query block per wave only: with two the 16 extra registers per block would push the kernel past 256 VGPRs.
  constexpr bool NEGM = (QB == 1) && !NOMAX;
  f32x16 negm[NEGM ? QB : 1];
  // Row sums of P on the MATRIX pipe: one more MFMA per 16-key chunk with an all-ones A operand leaves sum_k P[k][q] in every register
  // of the query's lane (both half-waves: the MFMA sums over all 16 keys of the chunk) - the sums of the bf16-rounded P, i.e. exactly the
  // weights the numerator uses. PMC (profiles/r3_flash_pmc.txt): the VALU is the busier pipe of this kernel (61 % against 43 %) and the
  // two overlap little; this moves 68 of ~300 vector instructions per tile (the adds + the cross-half exchange) to 8 MFMAs.
  constexpr bool LSUM = FA_LSUM_MFMA && QB == 2 && (NOMAX ? (FA_NOMAX_LSUM != 0) : (VAR == 0 || FA_LSUM_VAR1));      // (one query block per wave: 168 registers = three waves per SIMD; the 20 extra would cost the third)
  constexpr bool MSUM = LSUM;                              // row sums on the matrix pipe: no VALU partial sums
  f32x16 lacc[LSUM ? QB : 1];
  bf16x8 ones8;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones8[j] = (short)0x3F80;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[qb][i] = 0.f; o1[qb][i] = 0.f; }
    m_run[qb] = 0.f; l_run[qb] = 0.f;
    if (NEGM) {
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[qb][i] = 0.f;
    }
    if (LSUM) {
#pragma unroll
      for (int i = 0; i < 16; ++i) lacc[qb][i] = 0.f;
    }
  }

  // staging: 512 16-byte chunks per tile and operand, NS = 512 / NT per thread: row = tid >> 3 (+ NT / 8 per step), chunk = tid & 7.
  // Raw buffer loads: the per-lane offset is fixed for the whole key loop and the tile advances through the SCALAR offset, so a tile
  // costs no address arithmetic on the VALU (the 64-bit pointer form spent ~30 vector instructions per tile on it - beside 64
  // v_exp_f32 and 32 MFMAs); rows past N read as zero through the descriptor's range check (their scores are masked below).
  constexpr int NS = 512 / NT, RSTEP = NT / 8;
  const int srow0 = tid >> 3, sch = tid & 7;
  const int nt = (N + FA_KV - 1) / FA_KV;
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kb), 0, N * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Vb), 0, N * 128, 0x00020000);
#if FA_DMA
  // K / V tiles go global -> LDS directly (buffer_load ... lds: 1 KiB per wave instruction = 8 rows of 128 B, lane p writes position p):
  // no staging registers, no LDS store instructions. The images' XOR swizzles move to the SOURCE side - position c of row r holds data
  // chunk c ^ s(r) - and the tile still advances through the scalar offset; rows past N read as zero through the descriptor.
  int goffK[NS], goffV[NS];                 // per-lane byte offsets into the (sample, head)'s K / V
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int row = srow0 + RSTEP * i;
    goffK[i] = row * 128 + ((sch ^ ((row >> 1) & 7)) << 4);
    goffV[i] = row * 128 + ((sch ^ (((row >> 1) & 1) << 2)) << 4);
  }
  const int wrow0 = (tid >> 6) * 8;         // first row of this wave's 8-row piece (step i: + RSTEP * i)
  auto dma_tile = [&](int t, int buf) {
    const int soff = t * (FA_KV * 128);
    char* sb = smem + buf * (2 * FA_KV * 128);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass of hipcc has no such builtin: with it in sight the kernel's launch stub is silently dropped)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)(sb + (wrow0 + RSTEP * i) * 128), 16, goffK[i], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)(sb + FA_KV * 128 + (wrow0 + RSTEP * i) * 128), 16, goffV[i], soff, 0, 0);
#else
      (void)sb; (void)soff; (void)goffK[i]; (void)goffV[i]; (void)rsK; (void)rsV; (void)wrow0;
#endif
    }
  };
  auto load_tile = [&](int t, int buf) { dma_tile(t, buf); };
  auto store_tile = [&](int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };      // this wave's pieces have landed (the barrier publishes them)
#else
  u32x4 kreg[NS], vreg[NS];
  int goff[NS], kst[NS], vst[NS];          // per-lane byte offsets: in the (sample, head)'s K / V, in the LDS K image, in the LDS V image
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int row = srow0 + RSTEP * i;
    goff[i] = row * 128 + sch * 16;
    kst[i] = fa_swz(row, sch);
    vst[i] = FA_KV * 128 + fa_vswz(row, sch);
  }
  auto load_tile = [&](int t, int) {
    const int soff = t * (FA_KV * 128);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, goff[i], soff, 0));
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, goff[i], soff, 0));
    }
  };
  auto store_tile = [&](int buf) {
    char* sb = smem + buf * (2 * FA_KV * 128);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      *reinterpret_cast<u32x4*>(sb + kst[i]) = kreg[i];
      *reinterpret_cast<u32x4*>(sb + vst[i]) = vreg[i];
    }
  };
#endif
  int koff0[4], koff1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    koff0[s] = fa_swz(r, 2 * s + h);
    koff1[s] = fa_swz(32 + r, 2 * s + h);
  }
  // transposed V reads: lane li = lane & 15 of its 16-lane group supplies row q = li >> 2, columns 4 p .. 4 p + 3 (p = li & 3)
  // of the block [keys 16u + 4h .. +3] x [d 32 db + 16 g16 .. +15]; +8 keys = +1024 B, +16 keys (next u) = +2048 B
  unsigned voff[2];
  {
    const int li = lane & 15, g16 = (lane >> 4) & 1, vq = li >> 2, vp = li & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
      voff[db] = FA_KV * 128 + fa_vswz(4 * h + vq, 4 * db + 2 * g16 + (vp >> 1)) + 8 * (vp & 1);
  }

  // Prologue: the first K / V tile AND the Q rows are requested together (two independent round trips in flight at once; the Q rows
  // used to be loaded, waited for and converted before the first tile was even requested: ~1 us of a 25 us workgroup at N = 1029)
  load_tile(0, 0);
  // Q' fragments (B operand): lane (r,h) holds Q'[q0 + 32 qb + r][16 s + 8 h .. +8]
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qr = q0 + 32 * qb + r;
    qr = qr < q_end ? qr : q_end - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * 64 + 16 * s + 8 * h);
  }
  store_tile(0);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qb][s][j] = (short)f2bf(bf2f((bf16_t)qf[qb][s][j]) * scale_log2e);
  __syncthreads();

  // a wave whose 32 * QB query rows all lie past N (the ragged last block: N = 1029 leaves 5 rows for wave 0 and none for waves
  // 1-3) only helps to stage K / V: its SIMD's MFMA and VALU slots go to the other workgroup on the CU
  const bool active = q0 < q_end;
  // the key loop runs two tiles per iteration so that the LDS stage is a compile-time constant: every fragment read and staging
  // store then carries its stage offset as an immediate
  auto tile_body = [&](const int t, auto stage_tag) {
    constexpr int STG = decltype(stage_tag)::value;
    const bool more = t + 1 < nt;
    const char* sb = smem + STG * (2 * FA_KV * 128);
    const unsigned sbase = lds_base + (unsigned)(STG * (2 * FA_KV * 128));
    if (more) load_tile(t + 1, STG ^ 1);
    if (active) {
    // ---- S^T = K . Q'^T - m
    f32x16 s0[QB], s1[QB];
    if (!NEGM) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[qb][i] = NOMAX ? 0.f : -m_run[qb]; s1[qb][i] = NOMAX ? 0.f : -m_run[qb]; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(sb + koff0[s]);
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(sb + koff1[s]);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        if (NEGM && s == 0) {
          s0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[qb][s], negm[qb], 0, 0, 0);
          s1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[qb][s], negm[qb], 0, 0, 0);
        } else {
          s0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[qb][s], s0[qb], 0, 0, 0);
          s1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[qb][s], s1[qb], 0, 0, 0);
        }
      }
    }
    const int kbase = t * FA_KV;
    // first V^T fragments: issued now, waited for after the exponentials (the reads hide under the softmax chain)
    u32x2 va[2][2], vb[2][2];
#define FA_VISSUE(U, BUFI)                                                                                              \
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"                        \
               "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"                            \
               : "=v"(va[BUFI][0]), "=v"(va[BUFI][1]), "=v"(vb[BUFI][0]), "=v"(vb[BUFI][1])                             \
               : "v"(sbase + voff[0]), "v"(sbase + voff[1]), "i"((U) * 2048), "i"((U) * 2048 + 1024)                    \
               : "memory")
    // wait until all but the newest NEWER LDS operations of this wave are done; names the destinations so that no use of them
    // can be scheduled above the wait (cdna guide 5.7 item 1, form ii)
#define FA_VWAIT(BUFI, NEWER)                                                                                          \
  do {                                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                              \
                 : "+v"(va[BUFI][0]), "+v"(va[BUFI][1]), "+v"(vb[BUFI][0]), "+v"(vb[BUFI][1])                          \
                 : "n"(NEWER));                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
    float rel[QB];
    bool slow = (t == 0) && !NOMAX;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (kbase + FA_KV > N) {   // wave-uniform: key tail of the last tile
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key0 = kbase + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key0 >= N) s0[qb][i] = -INFINITY;
          if (key0 + 32 >= N) s1[qb][i] = -INFINITY;
        }
      }
      if (NOMAX) {
        rel[qb] = 0.f;
      } else if (t == 0) {
        // exact float maximum (may be negative): the first tile sets the scale
        float mx = fmaxf(s0[qb][0], s1[qb][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[qb][i], s1[qb][i]));
        rel[qb] = fmaxf(mx, __shfl_xor(mx, 32, 64));
      } else {
        // steady state: only growth above the stale max matters, and for that the INTEGER order of the bit patterns is
        // enough (positive floats order like ints, every negative float - and -inf - is a negative int). Integer max
        // needs no canonicalising v_max per MFMA output the way fmaxf does.
        // (__float_as_int, not __builtin_bit_cast: hipcc 7.2 folds a bit_cast of a vector ELEMENT to element 0)
        int mi = max(__float_as_int(s0[qb][0]), __float_as_int(s1[qb][0]));
#pragma unroll
        for (int i = 1; i < 16; ++i) mi = max(max(mi, __float_as_int(s0[qb][i])), __float_as_int(s1[qb][i]));      // one v_max3_i32 each
        const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);   // [0] = lower half's value, [1] = upper half's
        mi = max((int)sw[0], (int)sw[1]);
        rel[qb] = mi > 0 ? __int_as_float(mi) : 0.f;
        slow = slow || !(rel[qb] <= 100.f);                                        // also catches NaN
      }
    }
    if (!NOMAX && __any(slow)) {
      // textbook order (first tile; overflow guard): advance the max first, rescale, subtract, then exponentiate
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float adv = (t == 0 || rel[qb] > FA_THR) ? rel[qb] : 0.f;
        const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-adv);   // tile 0: O = l = 0, and 0 * 2^+big would be NaN
        m_run[qb] += adv;
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o0[qb][i] *= alpha; o1[qb][i] *= alpha;
          s0[qb][i] -= adv; s1[qb][i] -= adv;
          if (LSUM) lacc[qb][i] *= alpha;
        }
        if (NEGM) {
#pragma unroll
          for (int i = 0; i < 16; ++i) negm[qb][i] = -m_run[qb];
        }
        rel[qb] = 0.f;
      }
    }
    FA_VISSUE(0, 0);
    // ---- P^T = exp2(S^T) in four chunks of 16 keys, O^T += V^T . P^T chunk by chunk: the exponentials of chunk u + 1 sit between
    // the MFMAs of chunk u (an MFMA holds the SIMD's issue port for 8 of its 32 cycles: the VALU work rides in the other 24), and
    // the V^T fragments of step u + 1 are in flight while step u multiplies
    f32x2_t ps[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) ps[qb] = f32x2_t{0.f, 0.f};
#define FA_PSADD4(P, A, B, C, D) (P) += (f32x2_t{(A), (B)} + f32x2_t{(C), (D)})     /* (FA_DOT2SUM == 0) packed adds, half of them on the accumulator */
#define FA_EXP(U)                                                                                                       \
  do {                                                                                                                  \
    _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) {                                                                 \
      _Pragma("unroll") for (int j = 0; j < 8; j += 4) {                                                                \
        constexpr int e_ = 8 * ((U) & 1);                                                                               \
        if ((U) < 2) { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) s0[qb][e_ + j + jj] = FA_EXP2(s0[qb][e_ + j + jj]);              \
                       if (!MSUM && FA_X != 2 && !FA_DOT2SUM) FA_PSADD4(ps[qb], s0[qb][e_ + j], s0[qb][e_ + j + 1], s0[qb][e_ + j + 2], s0[qb][e_ + j + 3]); } \
        else { _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) s1[qb][e_ + j + jj] = FA_EXP2(s1[qb][e_ + j + jj]);                      \
               if (!MSUM && FA_X != 2 && !FA_DOT2SUM) FA_PSADD4(ps[qb], s1[qb][e_ + j], s1[qb][e_ + j + 1], s1[qb][e_ + j + 2], s1[qb][e_ + j + 3]); }   \
      }                                                                                                                 \
    }                                                                                                                   \
  } while (0)
#define FA_PV(U, BUFI)                                                                                                  \
  do {                                                                                                                  \
    const u32x4 ta = {va[BUFI][0][0], va[BUFI][0][1], va[BUFI][1][0], va[BUFI][1][1]};                                  \
    const u32x4 tb = {vb[BUFI][0][0], vb[BUFI][0][1], vb[BUFI][1][0], vb[BUFI][1][1]};                                  \
    FA_PRIO_HI();                                                                                                       \
    _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) {                                                                 \
      fa_f32x8 pv;                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) pv[j] = (U) < 2 ? s0[qb][8 * ((U) & 1) + j] : s1[qb][8 * ((U) & 1) + j]; \
      const fa_bf8 pb = __builtin_convertvector(pv, fa_bf8);          /* four v_cvt_pk_bf16_f32 */                     \
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);                                                                 \
      o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ta), pf, o0[qb], 0, 0, 0);            \
      o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tb), pf, o1[qb], 0, 0, 0);            \
      if (LSUM) lacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones8, pf, lacc[qb], 0, 0, 0);                       \
      if (!MSUM && FA_DOT2SUM && FA_X != 2) {      /* (sub-vectors by shufflevector: hipcc 7.2 folds a bit_cast of a vector ELEMENT to element 0) */ \
        const fa_bf2 one2 = {(__bf16)1.0f, (__bf16)1.0f};                                                               \
        float sa = ps[qb][0], sb = ps[qb][1];                                                                           \
        sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(pb, pb, 0, 1), one2, sa, false);                   \
        sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(pb, pb, 2, 3), one2, sb, false);                   \
        sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(pb, pb, 4, 5), one2, sa, false);                   \
        sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(pb, pb, 6, 7), one2, sb, false);                   \
        ps[qb] = f32x2_t{sa, sb};                                                                                       \
      }                                                                                                                 \
    }                                                                                                                   \
    FA_PRIO_LO();                                                                                                       \
  } while (0)
    // one MFMA, then a few of the next chunk's VALU instructions, ... (scheduler hint for the region up to the next wait)
#if FA_MIXMODE == 0
#define FA_MIX()                                                                                                        \
  do {                                                                                                                  \
    _Pragma("unroll") for (int g = 0; g < 2 * QB; ++g) {                                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                \
      __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);                                                                \
    }                                                                                                                   \
  } while (0)
#else
#define FA_MIX()                                                                                                        \
  do {                                                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 3 * QB, 0);                                                             \
    __builtin_amdgcn_sched_group_barrier(0x006, 64, 0);                                                                 \
  } while (0)
#endif
    FA_EXP(0);
    FA_VISSUE(1, 1);
    FA_VWAIT(0, 4);
    FA_PV(0, 0);
    FA_EXP(1);
    FA_MIX();
    FA_VISSUE(2, 0);
    FA_VWAIT(1, 4);
    FA_PV(1, 1);
    FA_EXP(2);
    FA_MIX();
    FA_VISSUE(3, 1);
    FA_VWAIT(0, 4);
    FA_PV(2, 0);
    FA_EXP(3);
    FA_MIX();
    FA_VWAIT(1, 0);
    FA_PV(3, 1);
    if (!MSUM) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float pst = ps[qb][0] + ps[qb][1];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pst), __float_as_uint(pst), false, false);
        l_run[qb] += __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
    }
    // ---- deferred rescale: only when some lane's scores outgrew the stale max by more than THR
    bool grow = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) grow = grow || (rel[qb] > FA_THR);
    if (!NOMAX && __any(grow)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float adv = rel[qb] > FA_THR ? rel[qb] : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-adv);
        m_run[qb] += adv;
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[qb][i] *= alpha; o1[qb][i] *= alpha; if (LSUM) lacc[qb][i] *= alpha; }
        if (NEGM) {
#pragma unroll
          for (int i = 0; i < 16; ++i) negm[qb][i] = -m_run[qb];
        }
      }
    }
    }   // active
    if (more) store_tile(STG ^ 1);
#if FA_X != 4
    __syncthreads();
#endif
  };
  // Steady-state tile (not the first, no key tail; two query blocks per wave): the scores of keys 32-63 are multiplied WHILE keys
  // 0-31 are scanned for growth and exponentiated - the deferred maximum makes that legal: the exponentials use the stale scale, the
  // growth test only decides about a rescale afterwards. The overflow guard (a score more than 2^100 above the stale maximum) is
  // evaluated on both halves before any P reaches a P.V product; in that rare case the first half's scores are multiplied again
  // (its registers hold exponentials by then) and the tile takes the textbook order.
  auto tile_fast = [&](const int t, auto stage_tag) {
    constexpr int STG = decltype(stage_tag)::value;
    const bool more = t + 1 < nt;
    const char* sb = smem + STG * (2 * FA_KV * 128);
    const unsigned sbase = lds_base + (unsigned)(STG * (2 * FA_KV * 128));
    if (more) load_tile(t + 1, STG ^ 1);
    if (active) {
      f32x16 s0[QB], s1[QB];
      u32x2 va[2][2], vb[2][2];
      auto qk0 = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int i = 0; i < 16; ++i) s0[qb][i] = -m_run[qb];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + koff0[sidx]);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][sidx], s0[qb], 0, 0, 0);
        }
      };
      qk0();
      // ---- keys 32-63 on the matrix pipe; keys 0-31: growth scan + exponentials on the VALU between those MFMAs
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s1[qb][i] = -m_run[qb];
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const bf16x8 kg = *reinterpret_cast<const bf16x8*>(sb + koff1[sidx]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kg, qf[qb][sidx], s1[qb], 0, 0, 0);
      }
      int mi[QB];
      f32x2_t ps[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        ps[qb] = f32x2_t{0.f, 0.f};
        mi[qb] = max(__float_as_int(s0[qb][0]), __float_as_int(s0[qb][1]));
#pragma unroll
        for (int i = 2; i < (FA_X == 3 ? 2 : 16); i += 2) mi[qb] = max(max(mi[qb], __float_as_int(s0[qb][i])), __float_as_int(s0[qb][i + 1]));
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) s0[qb][i + ii] = FA_EXP2(s0[qb][i + ii]);
          if (!MSUM && FA_X != 2 && !FA_DOT2SUM) FA_PSADD4(ps[qb], s0[qb][i], s0[qb][i + 1], s0[qb][i + 2], s0[qb][i + 3]);
        }
      }
      // (the exponentials above are NOT issued here in the generated code: the textbook-order branch below recomputes s0, which makes them
      // dead on that path, and LLVM sinks them into the fast path's successor block; pinning them here measured 3 % slower)
#if FA_QK_HINT
#pragma unroll
      for (int gidx = 0; gidx < 4 * QB; ++gidx) {           // one MFMA, then six of the 24 VALU instructions per MFMA, ...
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
      }
#endif
      // ---- keys 32-63: growth scan; both halves decide
      const int kbase = t * FA_KV;
      const bool special = (t == 0) || (kbase + FA_KV > N);          // wave-uniform: first tile (sets the scale) / key tail
      float rel[QB];
      bool slow = special;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int i = 0; i < (FA_X == 3 ? 2 : 16); i += 2) mi[qb] = max(max(mi[qb], __float_as_int(s1[qb][i])), __float_as_int(s1[qb][i + 1]));
        const auto sw = __builtin_amdgcn_permlane32_swap(mi[qb], mi[qb], false, false);
        const int mm = max((int)sw[0], (int)sw[1]);
        rel[qb] = mm > 0 ? __int_as_float(mm) : 0.f;
        slow = slow || !(rel[qb] <= 100.f);                                        // also catches NaN
      }
      if (__any(slow)) {
        // textbook order (first tile, key tail, overflow guard): the first half's scores again (its registers hold exponentials),
        // mask, exact maximum, advance, rescale, subtract, exponentiate
        qk0();
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          if (kbase + FA_KV > N) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int key0 = kbase + (i & 3) + 8 * (i >> 2) + 4 * h;
              if (key0 >= N) s0[qb][i] = -INFINITY;
              if (key0 + 32 >= N) s1[qb][i] = -INFINITY;
            }
          }
          if (special) {
            float mx = fmaxf(s0[qb][0], s1[qb][0]);                                // exact float maximum (may be negative)
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[qb][i], s1[qb][i]));
            rel[qb] = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (t > 0 && !(rel[qb] > FA_THR)) rel[qb] = rel[qb] > 0.f ? rel[qb] : 0.f;     // (tail tile: growth below the threshold is deferred as usual)
          }
          const float adv = (t == 0 || rel[qb] > FA_THR) ? rel[qb] : 0.f;
          const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-adv);         // tile 0: O = l = 0, and 0 * 2^+big would be NaN
          m_run[qb] += adv;
          l_run[qb] *= alpha;
          ps[qb] = f32x2_t{0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            o0[qb][i] *= alpha; o1[qb][i] *= alpha;
            if (LSUM) lacc[qb][i] *= alpha;
              s0[qb][i] = __builtin_amdgcn_exp2f(s0[qb][i] - adv);
            if (!MSUM && !FA_DOT2SUM) ps[qb][i & 1] += s0[qb][i];
            s1[qb][i] -= adv;
          }
          rel[qb] = (t > 0 && !(adv > 0.f)) ? rel[qb] : 0.f;
        }
      }
      FA_VISSUE(0, 0);
      FA_VISSUE(1, 1);
      FA_VWAIT(0, 4);
      FA_PV(0, 0);
      FA_EXP(2);
      FA_MIX();
      FA_VISSUE(2, 0);
      FA_VWAIT(1, 4);
      FA_PV(1, 1);
      FA_EXP(3);
      FA_MIX();
      FA_VISSUE(3, 1);
      FA_VWAIT(0, 4);
      FA_PV(2, 0);
      FA_VWAIT(1, 0);
      FA_PV(3, 1);
      if (!MSUM) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float pst = ps[qb][0] + ps[qb][1];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pst), __float_as_uint(pst), false, false);
          l_run[qb] += __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
      }
      bool grow = false;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) grow = grow || (rel[qb] > FA_THR);
      if (__any(grow)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float adv = rel[qb] > FA_THR ? rel[qb] : 0.f;
          const float alpha = __builtin_amdgcn_exp2f(-adv);
          m_run[qb] += adv;
          l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) { o0[qb][i] *= alpha; o1[qb][i] *= alpha; if (LSUM) lacc[qb][i] *= alpha; }
        }
      }
    }
    if (more) store_tile(STG ^ 1);
#if FA_X != 4
    __syncthreads();
#endif
  };
  constexpr bool FAST = (QB == 2) && FA_QK_OVERLAP && VAR == 1 && !NOMAX;
  for (int t = 0; t < nt; t += 2) {
    if constexpr (FAST) {
      tile_fast(t, std::integral_constant<int, 0>{});
      if (t + 1 < nt) tile_fast(t + 1, std::integral_constant<int, 1>{});
    } else {
      tile_body(t, std::integral_constant<int, 0>{});
      if (t + 1 < nt) tile_body(t + 1, std::integral_constant<int, 1>{});
    }
  }

  // ---- epilogue: lane owns query q0 + 32*qb + r, d = 32*db + (i&3) + 8*(i>>2) + 4h. Straight from the registers a wave instruction
  // would write 32 rows x 16 B - 32 partially written 128-byte lines per instruction, 32 instructions per wave. The K / V stages are dead
  // (every wave passed the last tile's barrier), so the wave lays its 32 * QB rows x 128 B out in a private slice of them - 16-byte chunk
  // XOR (row >> 1) & 7: the 8-byte quad writes of 16 rows x 2 halves cover all 64 banks - and stores WHOLE rows: 16 B per lane, 8 lanes
  // per row, 8 rows per instruction (cdna guide: attention, "O staged through LDS and stored as whole rows").
  if (active) {
    constexpr int RW = 32 * QB;                                 // rows of this wave
    // (the thread index is made opaque here: every address below is formed AFTER the key loop - hoisted in front of it they cost the
    // loop 4-11 registers: a spill in the VAR = 1 build, the third wave per SIMD in the QB = 1 build)
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int lane_e = te & 63, wid_e = __builtin_amdgcn_readfirstlane(te >> 6);
    const int r = lane_e & 31, h = lane_e >> 5, lane = lane_e;
    const int q0 = q_begin + bx * (NT / 2 * QB) + wid_e * (32 * QB);
    char* ob = smem + wid_e * (RW * 128);
    const long g = nb / NH;
    const int hd = (int)(nb % NH);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float inv = 1.f / (LSUM ? lacc[qb][0] : l_run[qb]);
      const int row = 32 * qb + r;
      const int sw = (row >> 1) & 7;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        u32x2 w0, w1;
        w0[0] = pack2bf(o0[qb][4 * i4 + 0] * inv, o0[qb][4 * i4 + 1] * inv);
        w0[1] = pack2bf(o0[qb][4 * i4 + 2] * inv, o0[qb][4 * i4 + 3] * inv);
        w1[0] = pack2bf(o1[qb][4 * i4 + 0] * inv, o1[qb][4 * i4 + 1] * inv);
        w1[1] = pack2bf(o1[qb][4 * i4 + 2] * inv, o1[qb][4 * i4 + 3] * inv);
        *reinterpret_cast<u32x2*>(ob + row * 128 + ((i4 ^ sw) << 4) + 8 * h) = w0;            // d = 8 i4 + 4 h .. + 3
        *reinterpret_cast<u32x2*>(ob + row * 128 + (((4 + i4) ^ sw) << 4) + 8 * h) = w1;      // d = 32 + 8 i4 + 4 h .. + 3
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's own writes (no other wave touches the slice)
    const int lr = lane >> 3, lc = lane & 7;
#pragma unroll
    for (int pass = 0; pass < RW / 8; ++pass) {
      const int row = pass * 8 + lr;
      const int q = q0 + row;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ob + row * 128 + ((lc ^ ((row >> 1) & 7)) << 4));
      if (q < q_end) *reinterpret_cast<u32x4*>(O + (g * o_rows + q) * ldo + hd * 64 + lc * 8) = v;
    }
  }
}

}  // namespace

extern "C" int vq3_im2col_norm(const float* images, void* patches, int32_t NI, int32_t H, int32_t W, int32_t p,
                               int32_t Kp, const float* mean3_host, const float* std3_host, void* stream) {
  VQ3_CHECK_ARG(images && patches && mean3_host && std3_host, "im2col_norm: null pointer");
  VQ3_CHECK_ARG(NI > 0 && p > 0 && H % p == 0 && W % p == 0 && Kp >= 3 * p * p, "im2col_norm: bad shape");
  const int Hp = H / p, Wp = W / p;
  hipLaunchKernelGGL(im2col_norm_kernel, dim3((unsigned)((long)NI * Hp * Wp)), dim3(256), 0, (hipStream_t)stream, images,
                     (bf16_t*)patches, H, W, p, Hp, Wp, Kp, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0],
                     std3_host[1], std3_host[2]);
  VQ3_CHECK_LAUNCH("im2col_norm");
  return 0;
}

extern "C" int vq3_vit_qkprep(const void* qkv, const float* qn_w, const float* qn_b, const float* kn_w,
                              const float* kn_b, const void* cos, const void* sin, void* Q, void* K, void* V, int64_t T,
                              int32_t N, int32_t NH, int32_t head_dim, int32_t tokens_per_frame, int32_t patch_start,
                              int32_t Wp, int32_t use_norm, int32_t use_rope, float eps, void* stream) {
  VQ3_CHECK_ARG(qkv && Q && K && V, "vit_qkprep: null pointer");
  VQ3_CHECK_ARG(head_dim == 64, "vit_qkprep: head_dim must be 64, got %d", head_dim);
  VQ3_CHECK_ARG(T > 0 && N > 0 && T % N == 0 && NH > 0 && NH % 2 == 0, "vit_qkprep: bad shape (NH must be even)");
  VQ3_CHECK_ARG(!use_norm || (qn_w && qn_b && kn_w && kn_b), "vit_qkprep: norm weights missing");
  VQ3_CHECK_ARG(!use_rope || (cos && sin && tokens_per_frame > 0 && Wp > 0), "vit_qkprep: rope tables missing");
  const char* force2 = getenv("VQ3_VIT_QKPREP_VEC2");   // tests: compare the two lane layouts on the same input
  const bool vec4 = !(force2 && force2[0] == '1') && NH % 4 == 0 && (!use_norm || (((uintptr_t)qn_w | (uintptr_t)qn_b | (uintptr_t)kn_w | (uintptr_t)kn_b) % 16 == 0)) &&
                    (!use_rope || (((uintptr_t)cos | (uintptr_t)sin) % 8 == 0));
  if (vec4)
    hipLaunchKernelGGL(vit_qkprep4_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, qn_w,
                       qn_b, kn_w, kn_b, (const bf16_t*)cos, (const bf16_t*)sin, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V, N, NH,
                       tokens_per_frame, patch_start, Wp, use_norm, use_rope, eps);
  else
    hipLaunchKernelGGL(vit_qkprep_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, qn_w,
                       qn_b, kn_w, kn_b, (const bf16_t*)cos, (const bf16_t*)sin, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V, N, NH,
                       tokens_per_frame, patch_start, Wp, use_norm, use_rope, eps);
  VQ3_CHECK_LAUNCH("vit_qkprep");
  return 0;
}

constexpr float FA_NOMAX_BOUND = 90.f;      // |score| * log2(e) up to which the kernels without a running maximum are used

static int flash_attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N, int32_t q_rows,
                               int32_t head_dim, int64_t ldo, float scale, void* stream, float score_bound = -1.f) {
  VQ3_CHECK_ARG(Q && K && V && O, "flash_attn_fwd: null pointer");
  VQ3_CHECK_ARG(q_rows > 0 && q_rows <= N, "flash_attn_fwd: q_rows must be in 1..N, got %d", q_rows);
  VQ3_CHECK_ARG(head_dim == 64, "flash_attn_fwd: head_dim must be 64, got %d", head_dim);
  VQ3_CHECK_ARG(G > 0 && NH > 0 && N > 0, "flash_attn_fwd: bad shape");
  VQ3_CHECK_ARG((long)G * NH <= 65535, "flash_attn_fwd: too many (group, head) pairs");
  VQ3_CHECK_ARG(ldo >= (long)NH * 64 && ldo % 8 == 0, "flash_attn_fwd: bad ldo (rows leave as 16-byte pieces: ldo %% 8 == 0)");
  VQ3_CHECK_ARG(((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) % 16 == 0, "flash_attn_fwd: misaligned operand");
  static int qb_forced = -1;
  if (qb_forced < 0) { const char* e = getenv("VQ3_FLASH_QB"); qb_forced = e ? atoi(e) : 0; }
  // two query blocks per wave (every K / V fragment read feeds two MFMA chains) once there are enough rows to fill them
  int qb = qb_forced ? qb_forced : (q_rows >= 512 ? 2 : 1);
  if (!qb_forced && qb == 2) {
    // 512 workgroup slots (2 per CU at 2 query blocks per wave): a grid of 1 .. 2 rounds whose last round is mostly empty - one
    // 8 232-token sample: 33 x 16 = 528 workgroups - runs better as 3-per-CU workgroups of half the rows (602 -> 670 TF/s)
    const double r = (double)((q_rows + 255) / 256) * G * NH / 512.0;
    if (r > 1.0 && r < 2.0 && (double)(long)(r + 0.999999) - r > 0.4) qb = 1;
  }
  const float sl2 = scale * 1.44269504088896340736f;
  int q_main = q_rows;
  bool forked = false;
  int tail_begin = -1;
  // ragged tail of <= 32 rows past a multiple of 256 (N = 1029 = 5 special tokens + 1024 patches leaves 5), two query blocks per wave:
  //   mode 2 (default, tail <= 16): the four-wave launch takes the exact blocks and ONE MORE workgroup per (sample, head) pair runs the tail
  //          rows with the keys split over its four waves (flash_tail_body), reading K / V out of the L2 the pair's other blocks fill;
  //   mode 1 (tail <= 32, >= 512 pairs): the tail as a launch of its own on a side stream, forked from / joined to `stream` by events
  //          (keys split over four waves up to 16 rows, VQ3_FLASH_TAIL_KSPLIT=0 or 17-32 rows: one wave per pair);
  //   mode 0: the tail rows as a fifth four-wave block of the main launch (three of its waves only help to stage K / V).
  // Measured at 48 x 16 pairs x 1029 (tools/bench_flash.py, us): see DESIGN.md section 3 "Flash attention".
  const int tail = N % 256;
  static int tail_mode = -1, tail_ksplit = -1;
  static hipStream_t side = nullptr;
  static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  if (tail_mode < 0) { const char* e = getenv("VQ3_FLASH_TAIL_MODE"); tail_mode = e ? atoi(e) : 2; }
  if (tail_ksplit < 0) { const char* e = getenv("VQ3_FLASH_TAIL_KSPLIT"); tail_ksplit = e ? atoi(e) : 1; }
  const bool ragged = q_rows == N && qb == 2 && tail > 0 && tail <= 32 && N > 256;
  if (ragged && tail_mode == 2 && tail <= 16) {
    q_main = N - tail;
    tail_begin = q_main;
  } else if (ragged && tail_mode >= 1 && (long)G * NH >= 512) {
    if (!side) {
      if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        side = nullptr;
      }
    }
    if (side) {
      q_main = N - tail;
      forked = true;
      (void)hipEventRecord(ev_fork, (hipStream_t)stream);
      (void)hipStreamWaitEvent(side, ev_fork, 0);
      if (tail <= 16 && tail_ksplit)
        hipLaunchKernelGGL(flash_tail_hd64_kernel, dim3(G * NH), dim3(256), 0, side, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V,
                           (bf16_t*)O, N, NH, (long)ldo, sl2, q_main, N);
      else
        hipLaunchKernelGGL((flash_attn_hd64_kernel<1, 64>), dim3(1, G * NH), dim3(64), 0, side, (const bf16_t*)Q, (const bf16_t*)K,
                           (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, q_main, N, N, -1, 0);
      (void)hipEventRecord(ev_join, side);
    }
  }
  dim3 grid((q_main + 128 * qb - 1) / (128 * qb) + (tail_begin >= 0 ? 1 : 0), G * NH);
  // XCD-aware 1-D grid (see the kernel): VQ3_FLASH_XCD=0 keeps the (block, pair) grid for A/B runs
  static int xcd_env = -1;
  if (xcd_env < 0) { const char* e = getenv("VQ3_FLASH_XCD"); xcd_env = e ? atoi(e) : 1; }
  int nblk = 0;
  if (xcd_env && grid.x > 1 && (long)grid.x * grid.y < (1L << 30)) {
    nblk = (int)grid.x;
    grid = dim3(grid.x * grid.y, 1);
  }
  // variant: measured (tools/bench_flash.py, 6 x 16 heads): 8232 keys 920 (0) / 951 (1) TF/s, 1029 keys 563 / 560 - the overlap pays once the
  // key loop is long; VQ3_FLASH_VAR pins one
  static int fvar_env = -2;
  if (fvar_env == -2) { const char* e = getenv("VQ3_FLASH_VAR"); fvar_env = e ? atoi(e) : -1; }
  const int fvar = fvar_env >= 0 ? fvar_env : (N >= 2048 ? 1 : 0);
  // the caller's bound on |scale * q.k| (vq3_flash_attn_fwd_bounded; < 0 = none): small enough -> the kernels without a running maximum
  static int nomax_env = -1;
  if (nomax_env < 0) { const char* e = getenv("VQ3_FLASH_NOMAX"); nomax_env = e ? atoi(e) : 1; }
  const bool nomax = nomax_env && score_bound >= 0.f && score_bound * 1.44269504088896340736f <= FA_NOMAX_BOUND;
  if (nomax && qb == 2)
    hipLaunchKernelGGL((flash_attn_hd64_kernel<2, 256, 0, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, 0, q_main, q_rows, tail_begin, nblk);
  else if (nomax)
    hipLaunchKernelGGL((flash_attn_hd64_kernel<1, 256, 0, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, 0, q_main, q_rows, -1, nblk);
  else if (qb == 2 && fvar == 1)
    hipLaunchKernelGGL((flash_attn_hd64_kernel<2, 256, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, 0, q_main, q_rows, tail_begin, nblk);
  else if (qb == 2)
    hipLaunchKernelGGL((flash_attn_hd64_kernel<2, 256>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, 0, q_main, q_rows, tail_begin, nblk);
  else
    hipLaunchKernelGGL((flash_attn_hd64_kernel<1, 256>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, N, NH, (long)ldo, sl2, 0, q_main, q_rows, -1, nblk);
  if (forked) (void)hipStreamWaitEvent((hipStream_t)stream, ev_join, 0);
  VQ3_CHECK_LAUNCH("flash_attn_fwd");
  return 0;
}

extern "C" int vq3_flash_attn_fwd(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH,
                                  int32_t N, int32_t head_dim, int64_t ldo, float scale, void* stream) {
  return flash_attn_fwd_impl(Q, K, V, O, G, NH, N, N, head_dim, ldo, scale, stream);
}

// The same two calls with a promise: |scale * q . k| <= score_bound for every (query, key) pair (e.g. derived from the weights of the
// q / k LayerNorms in front of the attention; q_rows = N for all queries). A bound of at most 90 / log2(e) = 62.4 selects kernels that keep no
// running maximum (exact: softmax is shift-invariant, and every 2^score then lies inside the bf16 / f32 range with room for the sums);
// a larger bound, or a negative one (= none), runs the general kernels. A WRONG promise can overflow - the caller owns it.
extern "C" int vq3_flash_attn_fwd_bounded(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N,
                                          int32_t q_rows, int32_t head_dim, int64_t ldo, float scale, float score_bound, void* stream) {
  return flash_attn_fwd_impl(Q, K, V, O, G, NH, N, q_rows, head_dim, ldo, scale, stream, score_bound);
}

// The same attention for the first q_rows queries of every group only (all N keys): O[(g*q_rows + n)*ldo + h*64 + d], n < q_rows. The
// last global block of the aggregator feeds only the first num_vis_tokens rows of each sample to the projector (vggt_qwen3_vlm.py:148-156).
extern "C" int vq3_flash_attn_fwd_rows(const void* Q, const void* K, const void* V, void* O, int32_t G, int32_t NH, int32_t N,
                                       int32_t q_rows, int32_t head_dim, int64_t ldo, float scale, void* stream) {
  return flash_attn_fwd_impl(Q, K, V, O, G, NH, N, q_rows, head_dim, ldo, scale, stream);
}
