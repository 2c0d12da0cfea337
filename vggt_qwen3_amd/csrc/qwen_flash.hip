// Fused causal GQA attention for Qwen3 (head_dim 128) - forward and backward - replacing the batched
// QK^T GEMM / softmax / PV GEMM chain of modeling_qwen3.py:185-207 (eager) / SDPA and its autograd backward.
// At Stage-1 lengths (L = 200) attention is < 1 % of the FLOPs but was 7 launches and ~270 MB of score traffic per
// layer; here scores never leave registers.
//
// MFMA 32x32x16 bf16 throughout, one wave per 32-row block, no LDS in the main loops:
//   forward / dQ kernels: S^T = K Q^T (key rows, query columns): the C layout puts the QUERY on the lane, so the online
//     softmax state, LSE and delta are per-lane scalars, and P^T / dS^T feed the next MFMA straight from the accumulator
//     registers (k order permuted: element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)).
//   dK/dV kernel: S = Q K^T (query rows, key columns): the KEY is on the lane, dK^T / dV^T accumulate in registers over
//     the q-blocks and (through LDS) over the 4 query heads of the GQA group - no atomics.
// The operand that would need a transposed tile (V^T, K^T, dO^T, Q^T as the A matrix) is gathered instead: MFMA row r of
// d-block db stands for feature d = 4 r + db, so one 8-byte load per key/query row feeds all four d-blocks.
// Layouts: Q, dQ [B, Hq, L, 128]; K, V, dK, dV [B, Hkv, L, 128]; O, dO token-major rows (b*L + q) with row stride ld,
// head h at column h*128; LSE (log2 domain, scaled) and Delta f32 [B, Hq, L].
#include <cstdlib>
#include "common.h"
#include "vq3_hip.h"

namespace {

constexpr int D = 128;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int rho(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(v[8 * s + j]);
  return o;
}
// Gather of the A operand of "X^T . (acc)" for all four d-blocks: row key(j) contributes 4 consecutive bf16 at column 4r.
// Done in two phases so the loads can be issued a whole tile ahead of their use.
__device__ __forceinline__ void gather_load(const bf16_t* base, long rowstride, int row0, int h, int r, int maxrow,
                                            u32x2 (&g)[2][8]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int row = row0 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      row = row < maxrow ? row : maxrow;
      g[s][j] = *reinterpret_cast<const u32x2*>(base + (long)row * rowstride + 4 * r);
    }
}
__device__ __forceinline__ void gather_split(const u32x2 (&g)[8], bf16x8 (&a)[4]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[0][j] = (short)(g[j][0] & 0xffff);
    a[1][j] = (short)(g[j][0] >> 16);
    a[2][j] = (short)(g[j][1] & 0xffff);
    a[3][j] = (short)(g[j][1] >> 16);
  }
}
__device__ __forceinline__ void rows_load(const bf16_t* base, long rowstride, int row, int h, int maxrow, bf16x8 (&f)[8]) {
  const bf16_t* p = base + (long)(row < maxrow ? row : maxrow) * rowstride + 8 * h;
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = ld8(p + 16 * t);
}
// lanes 0..31 (and their mirrors 32..63) vote on "key kb*32 + r is a real, attended key"
__device__ __forceinline__ unsigned key_bits(const uint8_t* km, int kb, int r, int L) {
  const int key = kb * 32 + r;
  const bool ok = key < L && km[key < L ? key : L - 1] != 0;
  return (unsigned)(__ballot(ok) & 0xffffffffull);
}

// The attended-key bits of every tile a wave will walk, computed ONCE into the wave's own LDS words (a keymask byte load inside
// the tile loop is a dependent global load, and its s_waitcnt vmcnt(0) sits on the tile's critical path). 256 keys = 8 tiles per
// round trip: four independent byte loads per lane.
constexpr int MAX_TILES = 256;     // L <= 8192
__device__ __forceinline__ void key_bits_all(const uint8_t* km, int ntiles, int L, unsigned* mine) {
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < ntiles * 32; base += 256) {
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = base + 64 * j + lane;
      ok[j] = key < L && km[key < L ? key : L - 1] != 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long m = __ballot(ok[j]);
      if (lane == 0) {
        mine[(base >> 5) + 2 * j] = (unsigned)m;
        mine[(base >> 5) + 2 * j + 1] = (unsigned)(m >> 32);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave reads back what it wrote (same wave: no barrier needed)
}
// Tiles without a single attended key contribute exact zeros (every probability is 0): they are dropped from the walk - as the
// causal upper triangle always was. With the reference's collator the attention mask covers only the ~20-50 text positions of a
// 200-column row, so most key tiles of a padded batch go. Returns the number of tiles kept; `list` (the wave's own LDS words)
// receives their indices in ascending order.
__device__ __forceinline__ int compact_tiles(const unsigned* bits, int ntiles, unsigned* list) {
  int n = 0;
  for (int kb = 0; kb < ntiles; ++kb) {
    if (bits[kb] != 0u) {
      if ((threadIdx.x & 63) == 0) list[n] = (unsigned)kb;
      ++n;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  return n;
}
// One dword per 128-byte line of a 32-row x 256-byte tile (64 lanes = its 64 lines): pulls the NEXT tile of the walk into this
// XCD's L2 a whole tile ahead, so that its real loads are L2 hits instead of a ~2 us cross-XCD round trip. Costs one register.
__device__ __forceinline__ unsigned touch_tile(const bf16_t* base, long rowstride, int row0, int maxrow) {
  const int lane = threadIdx.x & 63;
  int row = row0 + (lane >> 1);
  row = row < maxrow ? row : maxrow;
  return *reinterpret_cast<const unsigned*>(base + (long)row * rowstride + (lane & 1) * 64);
}

// ---- LDS staging helpers (the dK/dV pass's private stages, the fat kernels' K / V slots)
constexpr int DKV_TILE = 32 * 256;                       // bytes of a 32-row x 128-feature bf16 tile
constexpr int DKV_STAGE = 2 * DKV_TILE + 256;            // Q | dO | 32 LSE + 32 delta floats
__device__ __forceinline__ void dma_tile(const bf16_t* base, long rowstride, int row0, int maxrow, char* dst, int lane) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 4 * p + (lane >> 4);
    const int lchunk = (lane & 15) ^ (row & 15);
    int grow = row0 + row;
    grow = grow < maxrow ? grow : maxrow;
    const char* g = reinterpret_cast<const char*>(base + (long)grow * rowstride) + lchunk * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 lds_row16(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 256 + ((chunk ^ (row & 15)) << 4));
}
// the gather of gather_load / gather_split from an LDS tile: 4 consecutive features 4r .. 4r+3 of 8 rows per half
__device__ __forceinline__ void lds_gather(const char* tile, int sp, int h, int r, bf16x8 (&a)[4]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 16 * sp + 8 * (j >> 2) + 4 * h + (j & 3);
    const u32x2 g = *reinterpret_cast<const u32x2*>(tile + row * 256 + (((r >> 1) ^ (row & 15)) << 4) + ((r & 1) << 3));
    a[0][j] = (short)(g[0] & 0xffff);
    a[1][j] = (short)(g[0] >> 16);
    a[2][j] = (short)(g[1] & 0xffff);
    a[3][j] = (short)(g[1] >> 16);
  }
}


// Output tiles leave through LDS as whole rows. The accumulator layout gives a lane 8-byte pieces (4 features) of ONE row, 16 of them
// 16 bytes apart in pairs: stored straight from registers every store instruction scatters 64 pieces over 32 rows - 1 024 partial-line
// write requests per 8-KiB tile, and for these short kernels the request rate of the L2 write path is what they wait for. put: the
// piece goes to the wave's 8-KiB tile (32 rows x 256 B, 16-byte chunks XOR (row & 15)); flush: 8 passes of 4 whole rows, 16 B per lane.
__device__ __forceinline__ void tile_put8(char* tile, int row, int piece, const u32x2& w) {      // piece = feature / 4
  *reinterpret_cast<u32x2*>(tile + row * 256 + (((piece >> 1) ^ (row & 15)) << 4) + ((piece & 1) << 3)) = w;
}
__device__ __forceinline__ void tile_flush(const char* tile, bf16_t* base, long rowstride, int row0, int L, int lane) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 4 * p + (lane >> 4);
    const int phys = lane & 15, logical = phys ^ (row & 15);
    const u32x4 v = *reinterpret_cast<const u32x4*>(tile + row * 256 + (phys << 4));
    if (row0 + row < L) *reinterpret_cast<u32x4*>(base + (long)(row0 + row) * rowstride + logical * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward
// grid: B * Hkv * nqb blocks of 4 waves; wave g = query head hk*G + g (G = Hq / Hkv <= 4; spare waves idle)
// One 32-row query block of one query head: the wave's whole forward. LDS_KV = false: K / V tiles come from global memory (rows +
// 8-byte gathers, all loads of a tile issued before its first use); true: they sit in LDS slots (slot li = the li-th attended tile of
// the sample: [K tile | V tile], 16-byte chunks XOR (row & 15), staged by dma_tile) and a step has no global load at all. Same
// arithmetic in the same order either way.
template <bool LDS_KV>
__device__ __forceinline__ void fwd_block(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                          bf16_t* __restrict__ O, float* __restrict__ LSE, int L, int Hq, int Hkv, long ldo, float scale,
                                          int b, int hk, int hq, int qb, const unsigned* bits_g, const unsigned* list_g, int nlist,
                                          const char* slots, char* otile, const bf16x8* qpre = nullptr) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int q = qb * 32 + r, qc = q < L ? q : L - 1;
  const bf16_t* Qr = Q + (((long)b * Hq + hq) * L + qc) * D + 8 * h;
  const bf16_t* Kb = K + ((long)b * Hkv + hk) * L * D;
  const bf16_t* Vb = V + ((long)b * Hkv + hk) * L * D;
  bf16x8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = qpre ? qpre[s] : ld8(Qr + 16 * s);      // (qpre: the caller requested the rows a query block ahead)
  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2e = scale * LOG2E;
  for (int li = 0; li < nlist; ++li) {
    const int kb = (int)list_g[li];
    // every load of the tile is issued before its first use: one memory round trip per tile; the next attended tile is touched
    // into L2 meanwhile
    unsigned tk = 0, tv = 0;
    bf16x8 kc[8];
    u32x2 vc[2][8];
    const char* slot = slots + li * (2 * 32 * 256);
    if (LDS_KV) {
#pragma unroll
      for (int t = 0; t < 8; ++t) kc[t] = *reinterpret_cast<const bf16x8*>(slot + r * 256 + (((2 * t + h) ^ (r & 15)) << 4));
    } else {
      const int nkb = (int)list_g[li + 1 < nlist ? li + 1 : li];
      tk = touch_tile(Kb, D, nkb * 32, L - 1); tv = touch_tile(Vb, D, nkb * 32, L - 1);
      rows_load(Kb, D, kb * 32 + r, h, L - 1, kc);
      gather_load(Vb, D, kb * 32, h, r, L - 1, vc);
    }
    f32x16 s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[t], qf[t], s, 0, 0, 0);
    const unsigned bits = bits_g[kb];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = rho(i, h);
      const bool ok = ((bits >> kl) & 1u) && (kb * 32 + kl <= q);
      s[i] = ok ? s[i] : -INFINITY;
      mx = fmaxf(mx, s[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sl2e);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], sl2e, -m_safe));
      ps += s[i];
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      bf16x8 vf[4];
      if (LDS_KV) lds_gather(slot + 32 * 256, sp, h, r, vf);
      else gather_split(vc[sp], vf);
      const bf16x8 pf = pack8(s, sp);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], pf, o[db], 0, 0, 0);
    }
    if (!LDS_KV) asm volatile("" ::"v"(tk), "v"(tv));   // keeps the touch loads alive; their wait lands here, after the tile's MFMAs
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    u32x2 w;
    w[0] = pack2bf(o[0][i] * inv, o[1][i] * inv);
    w[1] = pack2bf(o[2][i] * inv, o[3][i] * inv);
    tile_put8(otile, r, rho(i, h), w);
  }
  tile_flush(otile, O + (long)b * L * ldo + (long)hq * D, ldo, qb * 32, L, lane);
  if (h == 0 && q < L) LSE[((long)b * Hq + hq) * L + q] = l_run > 0.f ? m_run + __builtin_amdgcn_logf(l_run) : INFINITY;
}

// grid: B * Hkv * nqb blocks of 4 waves; wave g = query head hk*G + g (G = Hq / Hkv <= 4; spare waves idle)
__global__ __launch_bounds__(256) void qwen_flash_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                            const bf16_t* __restrict__ V, const uint8_t* __restrict__ keymask,
                                                            bf16_t* __restrict__ O, float* __restrict__ LSE, int L, int Hq,
                                                            int Hkv, long ldo, float scale) {
  __shared__ unsigned sbits[4][MAX_TILES];
  __shared__ unsigned slist[4][MAX_TILES];
  __shared__ __attribute__((aligned(16))) char ostage[4][32 * 256];
  const int g = threadIdx.x >> 6;
  const int nqb = (L + 31) / 32;
  // heaviest blocks first: block ids are handed out in order, so all (b, kv-head) pairs of the last query block (the
  // most key tiles) start before any lighter one and the light blocks fill the tail
  const int nbh = gridDim.x / nqb;                            // B * Hkv
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);
  const int hk = (blockIdx.x % nbh) % Hkv, b = (blockIdx.x % nbh) / Hkv;
  const int G = Hq / Hkv, hq = hk * G + g;
  if (g >= G) return;
  key_bits_all(keymask + (long)b * L, qb + 1, L, sbits[g]);
  const int nlist = compact_tiles(sbits[g], qb + 1, slist[g]);
  fwd_block<false>(Q, K, V, O, LSE, L, Hq, Hkv, ldo, scale, b, hk, hq, qb, sbits[g], slist[g], nlist, nullptr, ostage[g]);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
// One 32-row query block of one query head in the dQ pass (LDS_KV as in fwd_block: K / V tiles from global memory or from the
// workgroup's LDS slots; same arithmetic either way).
template <bool LDS_KV>
__device__ __forceinline__ void dq_block(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                         const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                         float* __restrict__ Delta, bf16_t* __restrict__ dQ, int L, int Hq, int Hkv, long ldo, long lddo,
                                         float scale, int b, int hk, int hq, int qb, const unsigned* bits_g, const unsigned* list_g,
                                         int nlist, const char* slots, char* otile) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int q = qb * 32 + r, qc = q < L ? q : L - 1;
  const bf16_t* Qr = Q + (((long)b * Hq + hq) * L + qc) * D + 8 * h;
  const bf16_t* Or = O + ((long)b * L + qc) * ldo + (long)hq * D + 8 * h;
  const bf16_t* dOr = dO + ((long)b * L + qc) * lddo + (long)hq * D + 8 * h;
  const bf16_t* Kb = K + ((long)b * Hkv + hk) * L * D;
  const bf16_t* Vb = V + ((long)b * Hkv + hk) * L * D;
  bf16x8 qf[8], dof[8];
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = ld8(Qr + 16 * s);
    dof[s] = ld8(dOr + 16 * s);
    const bf16x8 of = ld8(Or + 16 * s);
#pragma unroll
    for (int j = 0; j < 8; ++j) delta = fmaf(bf2f((bf16_t)dof[s][j]), bf2f((bf16_t)of[j]), delta);
  }
  delta += __shfl_xor(delta, 32, 64);
  const float lse = LSE[((long)b * Hq + hq) * L + qc];
  f32x16 dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;
  const float sl2e = scale * LOG2E;
  for (int li = 0; li < nlist; ++li) {
    const int kb = (int)list_g[li];
    unsigned tk = 0, tv = 0;
    bf16x8 kc[8], vc[8];
    u32x2 gc[2][8];
    const char* slot = slots + li * (2 * 32 * 256);
    if (LDS_KV) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        kc[t] = *reinterpret_cast<const bf16x8*>(slot + r * 256 + (((2 * t + h) ^ (r & 15)) << 4));
        vc[t] = *reinterpret_cast<const bf16x8*>(slot + 32 * 256 + r * 256 + (((2 * t + h) ^ (r & 15)) << 4));
      }
    } else {
      const int nkb = (int)list_g[li + 1 < nlist ? li + 1 : li];
      tk = touch_tile(Kb, D, nkb * 32, L - 1); tv = touch_tile(Vb, D, nkb * 32, L - 1);
      rows_load(Kb, D, kb * 32 + r, h, L - 1, kc);
      rows_load(Vb, D, kb * 32 + r, h, L - 1, vc);
      gather_load(Kb, D, kb * 32, h, r, L - 1, gc);
    }
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[t], qf[t], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[t], dof[t], dp, 0, 0, 0);
    }
    const unsigned bits = bits_g[kb];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kl = rho(i, h);
      const bool ok = ((bits >> kl) & 1u) && (kb * 32 + kl <= q);
      const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], sl2e, -lse)) : 0.f;
      s[i] = p * (dp[i] - delta) * scale;          // dS^T
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      bf16x8 kf[4];
      if (LDS_KV) lds_gather(slot, sp, h, r, kf);
      else gather_split(gc[sp], kf);
      const bf16x8 dsf = pack8(s, sp);
#pragma unroll
      for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[db], dsf, dq[db], 0, 0, 0);
    }
    if (!LDS_KV) asm volatile("" ::"v"(tk), "v"(tv));
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    u32x2 w;
    w[0] = pack2bf(dq[0][i], dq[1][i]);
    w[1] = pack2bf(dq[2][i], dq[3][i]);
    tile_put8(otile, r, rho(i, h), w);
  }
  tile_flush(otile, dQ + ((long)b * Hq + hq) * L * D, D, qb * 32, L, lane);
  if (h == 0 && q < L) Delta[((long)b * Hq + hq) * L + q] = delta;
}

__global__ __launch_bounds__(256) void qwen_flash_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, const uint8_t* __restrict__ keymask,
                                                               const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ LSE, float* __restrict__ Delta,
                                                               bf16_t* __restrict__ dQ, int L, int Hq, int Hkv, long ldo,
                                                               long lddo, float scale) {
  __shared__ unsigned sbits[4][MAX_TILES];
  __shared__ unsigned slist[4][MAX_TILES];
  __shared__ __attribute__((aligned(16))) char ostage[4][32 * 256];
  const int g = threadIdx.x >> 6;
  const int nqb = (L + 31) / 32;
  const int nbh = gridDim.x / nqb;
  const int qb = nqb - 1 - (int)(blockIdx.x / nbh);           // heaviest first (see the forward kernel)
  const int hk = (blockIdx.x % nbh) % Hkv, b = (blockIdx.x % nbh) / Hkv;
  const int G = Hq / Hkv, hq = hk * G + g;
  if (g >= G) return;
  key_bits_all(keymask + (long)b * L, qb + 1, L, sbits[g]);
  const int nlist = compact_tiles(sbits[g], qb + 1, slist[g]);
  dq_block<false>(Q, K, V, O, dO, LSE, Delta, dQ, L, Hq, Hkv, ldo, lddo, scale, b, hk, hq, qb, sbits[g], slist[g], nlist, nullptr, ostage[g]);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// grid: B * Hkv * nkb blocks; wave g = query head of the group; the four waves' accumulators meet in LDS.
__global__ __launch_bounds__(256) void qwen_flash_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, const uint8_t* __restrict__ keymask,
                                                                const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                                                const float* __restrict__ Delta, bf16_t* __restrict__ dK,
                                                                bf16_t* __restrict__ dV, int L, int Hq, int Hkv, long lddo,
                                                                float scale, int S, long part_stride) {
  __shared__ float red[4][16][64];     // one f32x16 accumulator block per wave
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nkb = (L + 31) / 32;
  // S workgroups share a key tile: workgroup sp takes the q-blocks kb + sp, kb + sp + S, ... and writes its own partial dK / dV
  // slab (summed by the q/k-prep backward, which reads them anyway) - the q-block walk of key tile 0 is the longest serial
  // chain of the layer's backward (7 steps at L = 200), this cuts it to ceil(7 / S)
  const int sp = blockIdx.x % S;
  const int bid = blockIdx.x / S;
  const int nbh = (gridDim.x / S) / nkb;
  const int kb = bid / nbh;                                     // kb = 0 walks the most q-blocks: heaviest first
  dK += sp * part_stride;
  dV += sp * part_stride;
  const int hk = (bid % nbh) % Hkv, b = (bid % nbh) / Hkv;
  const int G = Hq / Hkv, hq = hk * G + g;
  const bool active = g < G;
  const int key = kb * 32 + r, kc = key < L ? key : L - 1;
  const bf16_t* Kr = K + (((long)b * Hkv + hk) * L + kc) * D + 8 * h;
  const bf16_t* Vr = V + (((long)b * Hkv + hk) * L + kc) * D + 8 * h;
  const bool key_ok = key < L && keymask[(long)b * L + kc] != 0;
  if (__ballot(key_ok) == 0ull) {
    // no attended key in this tile (padding): dK = dV = 0 exactly, for every head of the group - same answer in all four waves
    if (key < L && g == 0) {
      bf16_t* dKz = dK + (((long)b * Hkv + hk) * L + key) * D + 64 * h;
      bf16_t* dVz = dV + (((long)b * Hkv + hk) * L + key) * D + 64 * h;
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        *reinterpret_cast<u32x4*>(dKz + 8 * c) = z;
        *reinterpret_cast<u32x4*>(dVz + 8 * c) = z;
      }
    }
    return;
  }
  bf16x8 kf[8], vf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) { kf[s] = ld8(Kr + 16 * s); vf[s] = ld8(Vr + 16 * s); }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
  const float sl2e = scale * LOG2E;
  if (active) {
    const bf16_t* Qh = Q + ((long)b * Hq + hq) * L * D;
    const bf16_t* dOh = dO + (long)b * L * lddo + (long)hq * D;
    const float* lse_h = LSE + ((long)b * Hq + hq) * L;
    const float* del_h = Delta + ((long)b * Hq + hq) * L;
    for (int qb = kb + sp; qb < nkb; qb += S) {
      // every load of the tile is issued before its first use: one memory round trip per tile instead of four; the next
      // q-block's Q and dO tiles are touched into L2 meanwhile
      const int nq = qb + S < nkb ? qb + S : qb;
      const unsigned tq = touch_tile(Qh, D, nq * 32, L - 1), tdo = touch_tile(dOh, lddo, nq * 32, L - 1);
      bf16x8 qrow[8], dorow[8];
      u32x2 gdo[2][8], gq[2][8];
      f32x4 lse4[4], del4[4];
      rows_load(Qh, D, qb * 32 + r, h, L - 1, qrow);
      rows_load(dOh, lddo, qb * 32 + r, h, L - 1, dorow);
      gather_load(dOh, lddo, qb * 32, h, r, L - 1, gdo);                 // dO^T rows d = 4r + db
      gather_load(Qh, D, qb * 32, h, r, L - 1, gq);                       // Q^T rows
#pragma unroll
      for (int c = 0; c < 4; ++c) {      // rows rho(4c .. 4c+3, h) = 8c + 4h .. +3 are consecutive
        // unconditional 16-byte loads (a branch here would drain the load queue): a group that starts past the end is
        // clamped (all its rows are masked), a group that straddles the end reads into the next row / the 4-float slack
        // the caller guarantees behind LSE and Delta - those rows are masked too
        int q0 = qb * 32 + 8 * c + 4 * h;
        q0 = q0 < L ? q0 : (L >= 4 ? L - 4 : 0);
        lse4[c] = *reinterpret_cast<const f32x4*>(lse_h + q0);
        del4[c] = *reinterpret_cast<const f32x4*>(del_h + q0);
      }
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qrow[t], kf[t], s, 0, 0, 0);      // S[q, key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dorow[t], vf[t], dp, 0, 0, 0);   // dP[q, key]
      }
      f32x16 p;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qg = qb * 32 + rho(i, h);
        const bool ok = key_ok && qg < L && key <= qg;
        const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], sl2e, -lse4[i >> 2][i & 3])) : 0.f;
        p[i] = pv;
        s[i] = pv * (dp[i] - del4[i >> 2][i & 3]) * scale;     // dS[q, key]
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        bf16x8 a[4];
        gather_split(gdo[sp], a);
        const bf16x8 pf = pack8(p, sp);
#pragma unroll
        for (int db = 0; db < 4; ++db) dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[db], pf, dv[db], 0, 0, 0);
        gather_split(gq[sp], a);
        const bf16x8 dsf = pack8(s, sp);
#pragma unroll
        for (int db = 0; db < 4; ++db) dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[db], dsf, dk[db], 0, 0, 0);
      }
      asm volatile("" ::"v"(tq), "v"(tdo));
    }
  }
  // sum the G query heads: 8 accumulator blocks (dk[0..3], dv[0..3]), each reduced through LDS; wave (blk & 3) keeps blk
  bf16_t* dKr = dK + (((long)b * Hkv + hk) * L + kc) * D;
  bf16_t* dVr = dV + (((long)b * Hkv + hk) * L + kc) * D;
  f32x16 keep[2];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) {
    const f32x16& mine = blk < 4 ? dk[blk] : dv[blk - 4];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[g][i][lane] = mine[i];
    __syncthreads();
    if ((blk & 3) == g) {
#pragma unroll
      for (int i = 0; i < 16; ++i) keep[blk >> 2][i] = red[0][i][lane] + red[1][i][lane] + red[2][i][lane] + red[3][i][lane];
    }
  }
  if (key >= L) return;
  // wave g holds d-block db = g of both dK^T and dV^T: element i is feature d = 4 rho(i) + g
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dKr[4 * rho(i, h) + g] = f2bf(keep[0][i]);
    dVr[4 * rho(i, h) + g] = f2bf(keep[1][i]);
  }
}


// ------------------------------------------------------------------------------------------------ backward: dK, dV with LDS-staged Q / dO
// Same mapping and arithmetic as qwen_flash_bwd_dkv_kernel; what changes is where a q-block's operands come from. There every step
// began with ~56 global loads per lane (rows of Q and dO, their 8-byte "transposed" gathers, LSE, delta) whose round trip nothing could
// hide: the kernel runs one wave per SIMD (accumulators for dK^T and dV^T alone are 128 registers), so a step cost ~4 us for 0.5 us of
// MFMA work. Here each wave owns two LDS stages of {Q tile, dO tile, LSE | delta} (32 rows x 256 B each, 16-byte chunks XOR (row & 15))
// filled by global_load_lds one q-block AHEAD (17 DMA instructions per stage, counted vmcnt - no barrier: a wave reads only what it
// staged itself), and both operand forms are LDS reads: the rows as ds_read_b128, the gathers as ds_read_b64 of the same tile.
__global__ __launch_bounds__(256) void qwen_flash_bwd_dkv_lds_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                    const bf16_t* __restrict__ V, const uint8_t* __restrict__ keymask,
                                                                    const bf16_t* __restrict__ dO, const float* __restrict__ LSE,
                                                                    const float* __restrict__ Delta, bf16_t* __restrict__ dK,
                                                                    bf16_t* __restrict__ dV, int L, int Hq, int Hkv, long lddo,
                                                                    float scale, int S, long part_stride) {
  extern __shared__ __attribute__((aligned(16))) char dkv_smem[];      // [4 waves][2 stages][DKV_STAGE], then red
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nkb = (L + 31) / 32;
  const int sp = blockIdx.x % S;
  const int bid = blockIdx.x / S;
  const int nbh = (gridDim.x / S) / nkb;
  const int kb = bid / nbh;                                     // kb = 0 walks the most q-blocks: heaviest first
  dK += sp * part_stride;
  dV += sp * part_stride;
  const int hk = (bid % nbh) % Hkv, b = (bid % nbh) / Hkv;
  const int G = Hq / Hkv, hq = hk * G + g;
  const bool active = g < G;
  const int key = kb * 32 + r, kc = key < L ? key : L - 1;
  const bf16_t* Kr = K + (((long)b * Hkv + hk) * L + kc) * D + 8 * h;
  const bf16_t* Vr = V + (((long)b * Hkv + hk) * L + kc) * D + 8 * h;
  const bool key_ok = key < L && keymask[(long)b * L + kc] != 0;
  if (__ballot(key_ok) == 0ull) {
    if (key < L && g == 0) {
      bf16_t* dKz = dK + (((long)b * Hkv + hk) * L + key) * D + 64 * h;
      bf16_t* dVz = dV + (((long)b * Hkv + hk) * L + key) * D + 64 * h;
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        *reinterpret_cast<u32x4*>(dKz + 8 * c) = z;
        *reinterpret_cast<u32x4*>(dVz + 8 * c) = z;
      }
    }
    return;
  }
  bf16x8 kf[8], vf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) { kf[s] = ld8(Kr + 16 * s); vf[s] = ld8(Vr + 16 * s); }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
  const float sl2e = scale * LOG2E;
  char* const mine = dkv_smem + __builtin_amdgcn_readfirstlane(g) * (2 * DKV_STAGE);      // wave-uniform: the DMA's LDS base goes through M0
  if (active) {
    const bf16_t* Qh = Q + ((long)b * Hq + hq) * L * D;
    const bf16_t* dOh = dO + (long)b * L * lddo + (long)hq * D;
    const float* lse_h = LSE + ((long)b * Hq + hq) * L;
    const float* del_h = Delta + ((long)b * Hq + hq) * L;
    auto issue = [&](int qb, char* st) {
      dma_tile(Qh, D, qb * 32, L - 1, st, lane);
      dma_tile(dOh, lddo, qb * 32, L - 1, st + DKV_TILE, lane);
      int q = qb * 32 + (lane & 31);
      q = q < L ? q : L - 1;
      const float* src = (lane < 32 ? lse_h : del_h) + q;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + 2 * DKV_TILE), 4, 0, 0);
    };
    int stage = 0;
    if (kb + sp < nkb) issue(kb + sp, mine);
    for (int qb = kb + sp; qb < nkb; qb += S) {
      char* st = mine + stage * DKV_STAGE;
      const bool more = qb + S < nkb;
      if (more) {
        issue(qb + S, mine + (stage ^ 1) * DKV_STAGE);
        asm volatile("s_waitcnt vmcnt(17)" ::: "memory");     // everything but the 17 DMA pieces just issued has landed (K / V rows included)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const char* qt = st;
      const char* dot = st + DKV_TILE;
      const float* ld = reinterpret_cast<const float*>(st + 2 * DKV_TILE);
      f32x4 lse4[4], del4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {      // rows rho(4c .. 4c+3, h) = 8c + 4h .. +3 are consecutive (rows past L hold row L-1's values: masked below)
        lse4[c] = *reinterpret_cast<const f32x4*>(ld + 8 * c + 4 * h);
        del4[c] = *reinterpret_cast<const f32x4*>(ld + 32 + 8 * c + 4 * h);
      }
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bf16x8 qrow = lds_row16(qt, r, 2 * t + h);
        const bf16x8 dorow = lds_row16(dot, r, 2 * t + h);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qrow, kf[t], s, 0, 0, 0);      // S[q, key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dorow, vf[t], dp, 0, 0, 0);   // dP[q, key]
      }
      f32x16 p;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qg = qb * 32 + rho(i, h);
        const bool ok = key_ok && qg < L && key <= qg;
        const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], sl2e, -lse4[i >> 2][i & 3])) : 0.f;
        p[i] = pv;
        s[i] = pv * (dp[i] - del4[i >> 2][i & 3]) * scale;     // dS[q, key]
      }
#pragma unroll
      for (int sp2 = 0; sp2 < 2; ++sp2) {
        bf16x8 a[4];
        lds_gather(dot, sp2, h, r, a);
        const bf16x8 pf = pack8(p, sp2);
#pragma unroll
        for (int db = 0; db < 4; ++db) dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[db], pf, dv[db], 0, 0, 0);
        lds_gather(qt, sp2, h, r, a);
        const bf16x8 dsf = pack8(s, sp2);
#pragma unroll
        for (int db = 0; db < 4; ++db) dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[db], dsf, dk[db], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this stage's reads are done before the NEXT iteration's DMA lands on the other one's successor
      stage ^= 1;
    }
  }
  // sum the G query heads (as in qwen_flash_bwd_dkv_kernel); the reduction scratch sits behind the stages
  float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(dkv_smem + 4 * 2 * DKV_STAGE);
  bf16_t* dKr = dK + (((long)b * Hkv + hk) * L + kc) * D;
  bf16_t* dVr = dV + (((long)b * Hkv + hk) * L + kc) * D;
  f32x16 keep[2];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) {
    const f32x16& mineacc = blk < 4 ? dk[blk] : dv[blk - 4];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) red[g][i][lane] = mineacc[i];
    __syncthreads();
    if ((blk & 3) == g) {
#pragma unroll
      for (int i = 0; i < 16; ++i) keep[blk >> 2][i] = red[0][i][lane] + red[1][i][lane] + red[2][i][lane] + red[3][i][lane];
    }
  }
  // wave g holds feature d = 4 rho(i) + g of both tiles: 2-byte pieces 8 bytes apart. Through two LDS tiles (the stages are dead: every
  // wave passed the reduction's barriers) they leave as whole 256-byte rows, 16 B per thread
  (void)dKr; (void)dVr;
  char* const tk = dkv_smem;
  char* const tv = dkv_smem + DKV_TILE;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = 4 * rho(i, h) + g;
    const int off = r * 256 + (((d >> 3) ^ (r & 15)) << 4) + ((d & 7) << 1);
    *reinterpret_cast<bf16_t*>(tk + off) = f2bf(keep[0][i]);
    *reinterpret_cast<bf16_t*>(tv + off) = f2bf(keep[1][i]);
  }
  __syncthreads();
  bf16_t* dKb = dK + ((long)b * Hkv + hk) * L * D;
  bf16_t* dVb = dV + ((long)b * Hkv + hk) * L * D;
#pragma unroll
  for (int pss = 0; pss < 2; ++pss) {
    const int idx = threadIdx.x + 256 * pss;                 // 512 chunks of 16 B per tile
    const int row = idx >> 4, phys = idx & 15, logical = phys ^ (row & 15);
    if (kb * 32 + row < L) {
      *reinterpret_cast<u32x4*>(dKb + (long)(kb * 32 + row) * D + logical * 8) = *reinterpret_cast<const u32x4*>(tk + row * 256 + (phys << 4));
      *reinterpret_cast<u32x4*>(dVb + (long)(kb * 32 + row) * D + logical * 8) = *reinterpret_cast<const u32x4*>(tv + row * 256 + (phys << 4));
    }
  }
}
constexpr int DKV_SMEM = 4 * 2 * DKV_STAGE + 4 * 16 * 64 * 4;

}  // namespace

static int flash_check(const char* who, int B, int L, int Hq, int Hkv, int Dh) {
  VQ3_CHECK_ARG(Dh == D, "%s: head_dim must be %d, got %d", who, D, Dh);
  VQ3_CHECK_ARG(B > 0 && L > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= 4,
                "%s: need 1..4 query heads per kv head (B=%d L=%d Hq=%d Hkv=%d)", who, B, L, Hq, Hkv);
  VQ3_CHECK_ARG((long)B * Hkv * ((L + 31) / 32) < (1l << 31), "%s: grid too large", who);
  VQ3_CHECK_ARG(L <= 32 * MAX_TILES, "%s: L=%d exceeds %d", who, L, 32 * MAX_TILES);
  return 0;
}

extern "C" int vq3_qwen_flash_fwd(const void* Q, const void* K, const void* V, const void* keymask, void* O, float* LSE,
                                  int32_t B, int32_t L, int32_t Hq, int32_t Hkv, int32_t Dh, int64_t ldo, float scale,
                                  void* stream) {
  VQ3_CHECK_ARG(Q && K && V && keymask && O && LSE, "qwen_flash_fwd: null pointer");
  if (flash_check("qwen_flash_fwd", B, L, Hq, Hkv, Dh)) return 1;
  VQ3_CHECK_ARG(ldo >= (long)Hq * D && ldo % 4 == 0, "qwen_flash_fwd: bad output row stride");
  const int nqb = (L + 31) / 32;
  hipLaunchKernelGGL(qwen_flash_fwd_kernel, dim3(B * Hkv * nqb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                     (const bf16_t*)K, (const bf16_t*)V, (const uint8_t*)keymask, (bf16_t*)O, LSE, L, Hq, Hkv, (long)ldo, scale);
  VQ3_CHECK_LAUNCH("qwen_flash_fwd");
  return 0;
}

static int launch_dkv(const void* Q, const void* K, const void* V, const void* keymask, const void* dO, const float* LSE, float* Delta,
                      void* dK, void* dV, int kv_parts, int B, int L, int Hq, int Hkv, long lddo, float scale, hipStream_t s);

extern "C" int vq3_qwen_flash_bwd(const void* Q, const void* K, const void* V, const void* keymask, const void* O,
                                  const void* dO, const float* LSE, float* Delta, void* dQ, void* dK, void* dV, int32_t kv_parts,
                                  int32_t B, int32_t L, int32_t Hq, int32_t Hkv, int32_t Dh, int64_t ldo, int64_t lddo,
                                  float scale, void* stream) {
  VQ3_CHECK_ARG(Q && K && V && keymask && O && dO && LSE && Delta && dQ && dK && dV, "qwen_flash_bwd: null pointer");
  if (flash_check("qwen_flash_bwd", B, L, Hq, Hkv, Dh)) return 1;
  VQ3_CHECK_ARG(ldo >= (long)Hq * D && lddo >= (long)Hq * D && ldo % 8 == 0 && lddo % 8 == 0, "qwen_flash_bwd: bad row strides");
  VQ3_CHECK_ARG(kv_parts >= 1 && kv_parts <= 4, "qwen_flash_bwd: kv_parts must be 1..4");
  const int nqb = (L + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(qwen_flash_bwd_dq_kernel, dim3(B * Hkv * nqb), dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K,
                     (const bf16_t*)V, (const uint8_t*)keymask, (const bf16_t*)O, (const bf16_t*)dO, LSE, Delta, (bf16_t*)dQ, L,
                     Hq, Hkv, (long)ldo, (long)lddo, scale);
  const int rc_dkv = launch_dkv(Q, K, V, keymask, dO, LSE, Delta, dK, dV, kv_parts, B, L, Hq, Hkv, (long)lddo, scale, s);
  if (rc_dkv) return rc_dkv;
  VQ3_CHECK_LAUNCH("qwen_flash_bwd");
  return 0;
}

static int launch_dkv(const void* Q, const void* K, const void* V, const void* keymask, const void* dO, const float* LSE, float* Delta,
                      void* dK, void* dV, int kv_parts, int B, int L, int Hq, int Hkv, long lddo, float scale, hipStream_t s) {
  const int nqb = (L + 31) / 32;
  // dK / dV pass: the LDS-staged kernel (operands of the next q-block in flight while this one multiplies); VQ3_QWEN_DKV_LDS=0 = the
  // register-staged one
  static int dkv_lds = -1;
  if (dkv_lds < 0) {
    const char* e = getenv("VQ3_QWEN_DKV_LDS");
    dkv_lds = e ? atoi(e) : 1;
    if (dkv_lds && hipFuncSetAttribute((const void*)qwen_flash_bwd_dkv_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_SMEM) != hipSuccess) {
      (void)hipGetLastError();
      dkv_lds = 0;
    }
  }
  if (dkv_lds)
    hipLaunchKernelGGL(qwen_flash_bwd_dkv_lds_kernel, dim3(B * Hkv * nqb * kv_parts), dim3(256), DKV_SMEM, s, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)V, (const uint8_t*)keymask, (const bf16_t*)dO, LSE, Delta, (bf16_t*)dK,
                       (bf16_t*)dV, L, Hq, Hkv, (long)lddo, scale, (int)kv_parts, (long)B * Hkv * L * D);
  else
    hipLaunchKernelGGL(qwen_flash_bwd_dkv_kernel, dim3(B * Hkv * nqb * kv_parts), dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)V, (const uint8_t*)keymask, (const bf16_t*)dO, LSE, Delta, (bf16_t*)dK, (bf16_t*)dV, L, Hq,
                       Hkv, (long)lddo, scale, (int)kv_parts, (long)B * Hkv * L * D);
  return 0;
}
