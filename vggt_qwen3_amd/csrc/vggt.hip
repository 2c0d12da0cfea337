// VGGT aggregator kernels (multi-view ViT): patch im2col with fused ImageNet normalisation, per-head q/k
// LayerNorm + 2-D RoPE + head split, and a flash-style attention forward (head_dim 64, non-causal) with LDS-staged,
// XOR-swizzled K / V^T tiles and v_mfma_f32_32x32x16_bf16 chains kept in registers:
//   S^T = K . Q^T        (A = K tile rows, B = Q fragment held in VGPRs for the whole kernel)
//   O^T += V^T . P^T     (the S^T accumulator, exponentiated and packed to bf16, IS the B operand: no LDS round trip)
// so every softmax statistic of a query row lives on the lane that owns that query column.
#include <cstdlib>

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
// Q,K bf16 [NB, N, 64]; Vt bf16 [NB, 64, Np] (V transposed, zero-padded to Np % 64 == 0); O bf16 token-major:
// O[(g*N + q) * ldo + h*64 + d] with NB = G*NH, g = nb / NH, h = nb % NH.
// Block = 4 waves x 32 query rows; KV tile = 64 keys. LDS: K tile [64][64] and V^T tile [64][64], 16-byte chunks
// XOR-swizzled with (row >> 1) & 7 so the fragment reads (lane = row) are conflict-free / 2-way.
constexpr int FA_KV = 64;
#ifndef VQ3_FA_DIAG
#define VQ3_FA_DIAG 0   // counters-only builds: 1 = no V^T reads, 2 = no K reads, 3 = no tile stores (outputs are then wrong)
#endif

__device__ __forceinline__ int fa_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// QB = 32-row query blocks per wave (1 or 2): with 2, every K / V^T fragment read from LDS feeds two MFMAs.
template <int QB>
__global__ __launch_bounds__(256) void flash_attn_hd64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                              int N, int Np, int NH, long ldo, float scale_log2e) {
  // two stages of (K tile | V^T tile): tile t+1 is written while tile t is read -> one barrier per tile
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * FA_KV * 128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const unsigned fa_lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the tiles
  const long nb = blockIdx.y;
  const int q0 = blockIdx.x * (128 * QB) + wid * (32 * QB);
  const bf16_t* Qb = Q + nb * (long)N * 64;
  const bf16_t* Kb = K + nb * (long)N * 64;
  const bf16_t* Vb = Vt + nb * 64L * Np;

  // Q fragments: B operand, lane (r,h) holds Q[q0 + 32*qb + r][16s + 8h .. +8]
  bf16x8 qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qr = q0 + 32 * qb + r;
    qr = qr < N ? qr : N - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(Qb + (long)qr * 64 + 16 * s + 8 * h);
  }
  f32x16 o0[QB], o1[QB];
  float m_run[QB], l_run[QB];   // running max in the scaled (log2) domain
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[qb][i] = 0.f; o1[qb][i] = 0.f; }
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
  }

  // staging: 512 16-byte chunks per tile, 2 per thread: chunk id c = tid + 256*i -> row c>>3, chunk c&7
  const int srow0 = tid >> 3, sch = tid & 7;
  const int nt = (N + FA_KV - 1) / FA_KV;
  u32x4 kreg[2], vreg[2];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + 32 * i;
      int kr = t * FA_KV + row;
      kr = kr < N ? kr : N - 1;
      kreg[i] = *reinterpret_cast<const u32x4*>(Kb + (long)kr * 64 + sch * 8);
      vreg[i] = *reinterpret_cast<const u32x4*>(Vb + (long)row * Np + t * FA_KV + sch * 8);
    }
  };
  auto store_tile = [&](int buf) {
    char* Ks = smem + buf * (2 * FA_KV * 128);
    char* Vs = Ks + FA_KV * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + 32 * i;
      *reinterpret_cast<u32x4*>(Ks + fa_swz(row, sch)) = kreg[i];
      // V^T is read back 8 bytes per lane by 32-lane groups on 64 banks: rows r and r+16 share a 16-byte slot (the chunk
      // swizzle has 8 positions), so rows with bit 4 set keep their two 8-byte halves swapped and a group covers all banks
      const u32x4 vv = (row & 16) ? u32x4{vreg[i][2], vreg[i][3], vreg[i][0], vreg[i][1]} : vreg[i];
      *reinterpret_cast<u32x4*>(Vs + fa_swz(row, sch)) = vv;
    }
  };
  // lane-constant LDS offsets of the fragment reads (relative to the stage base)
  int koff0[4], koff1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    koff0[s] = fa_swz(r, 2 * s + h);
    koff1[s] = fa_swz(32 + r, 2 * s + h);
  }
  int voffa[4][2], voffb[4][2];   // [sub*2+sp][run]
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int kb = 16 * u + 4 * h;
#pragma unroll
    for (int run = 0; run < 2; ++run) {
      const int hv = ((kb & 7) << 1) ^ (((r >> 4) & 1) << 3);   // 8-byte half, swapped for rows with bit 4 set (store_tile)
      voffa[u][run] = FA_KV * 128 + fa_swz(r, (kb + 8 * run) >> 3) + hv;
      voffb[u][run] = FA_KV * 128 + fa_swz(32 + r, (kb + 8 * run) >> 3) + hv;
    }
  }

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;
    const char* sb = smem + (t & 1) * (2 * FA_KV * 128);
    if (more) load_tile(t + 1);
    // ---- S^T = K . Q^T for the two 32-key sub-tiles (each K fragment feeds QB query blocks)
    f32x16 s0[QB], s1[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { s0[qb][i] = 0.f; s1[qb][i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 k0 = VQ3_FA_DIAG == 2 ? qf[0][s] : *reinterpret_cast<const bf16x8*>(sb + koff0[s]);
      const bf16x8 k1 = VQ3_FA_DIAG == 2 ? qf[0][s] : *reinterpret_cast<const bf16x8*>(sb + koff1[s]);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[qb][s], s0[qb], 0, 0, 0);
        s1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[qb][s], s1[qb], 0, 0, 0);
      }
    }
    // ---- online softmax (per query column = per lane pair (lane, lane^32)); masking only on the last tile
    const int kbase = t * FA_KV;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      if (kbase + FA_KV > N) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key0 = kbase + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key0 >= N) s0[qb][i] = -INFINITY;
          if (key0 + 32 >= N) s1[qb][i] = -INFINITY;
        }
      }
      float mx = fmaxf(s0[qb][0], s1[qb][0]);
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[qb][i], s1[qb][i]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx * scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s0[qb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[qb][i], scale_log2e, -m_new));
        s1[qb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[qb][i], scale_log2e, -m_new));
        ps += s0[qb][i] + s1[qb][i];
      }
      ps += __shfl_xor(ps, 32, 64);
      l_run[qb] = l_run[qb] * alpha + ps;
      m_run[qb] = m_new;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[qb][i] *= alpha; o1[qb][i] *= alpha; }
    }
    // ---- O^T += V^T . P^T : B operand = packed S^T registers 8s'..8s'+7 (k order: 16s' + 8(j>>2) + 4h + (j&3))
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // u = sub*2 + sp
      // Plain ds_read_b64 through inline asm: left to the compiler, the a/b reads (4 KiB apart) are merged into
      // ds_read2st64_b64, which is serviced on 32 banks in 16-lane groups at half the rate and made rows r, r+1 collide
      // (counters: every conflict cycle of this kernel, 32 % of its LDS time). The asm reads are waited for by hand.
      u32x2 a0, a1, b0, b1;
      if (VQ3_FA_DIAG == 1) {
        a0 = a1 = b0 = b1 = u32x2{(unsigned)u, (unsigned)lane};
      } else {
        const unsigned sbase = (unsigned)(size_t)(sb - smem) + fa_lds_base;
        // the four reads and their wait are ONE statement with early-clobber outputs: hipcc does not count LDS reads issued
        // from inline asm, so nothing may sit between them and the wait that could consume a0..b1 (cdna guide 5.7 item 1 form i)
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
                     : "v"(sbase + voffa[u][0]), "v"(sbase + voffa[u][1]), "v"(sbase + voffb[u][0]), "v"(sbase + voffb[u][1])
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      const u32x4 ta = {a0[0], a0[1], a1[0], a1[1]};
      const u32x4 tb = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(u < 2 ? s0[qb][8 * (u & 1) + j] : s1[qb][8 * (u & 1) + j]);
        o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ta), pf, o0[qb], 0, 0, 0);
        o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tb), pf, o1[qb], 0, 0, 0);
      }
    }
    if (more && VQ3_FA_DIAG != 3) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns query q0 + 32*qb + r, d = 32*db + (i&3) + 8*(i>>2) + 4h
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q0 + 32 * qb + r;
    if (q < N) {
      const float inv = 1.f / l_run[qb];
      const long g = nb / NH;
      const int hd = (int)(nb % NH);
      bf16_t* orow = O + (g * N + q) * ldo + hd * 64;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int d = 8 * i4 + 4 * h;
        u32x2 w0, w1;
        w0[0] = pack2bf(o0[qb][4 * i4 + 0] * inv, o0[qb][4 * i4 + 1] * inv);
        w0[1] = pack2bf(o0[qb][4 * i4 + 2] * inv, o0[qb][4 * i4 + 3] * inv);
        w1[0] = pack2bf(o1[qb][4 * i4 + 0] * inv, o1[qb][4 * i4 + 1] * inv);
        w1[1] = pack2bf(o1[qb][4 * i4 + 2] * inv, o1[qb][4 * i4 + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + d) = w0;
        *reinterpret_cast<u32x2*>(orow + 32 + d) = w1;
      }
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

extern "C" int vq3_flash_attn_fwd(const void* Q, const void* K, const void* Vt, void* O, int32_t G, int32_t NH,
                                  int32_t N, int32_t Np, int32_t head_dim, int64_t ldo, float scale, void* stream) {
  VQ3_CHECK_ARG(Q && K && Vt && O, "flash_attn_fwd: null pointer");
  VQ3_CHECK_ARG(head_dim == 64, "flash_attn_fwd: head_dim must be 64, got %d", head_dim);
  VQ3_CHECK_ARG(G > 0 && NH > 0 && N > 0 && Np >= N && Np % 64 == 0, "flash_attn_fwd: bad shape (Np %% 64)");
  VQ3_CHECK_ARG((long)G * NH <= 65535, "flash_attn_fwd: too many (group, head) pairs");
  VQ3_CHECK_ARG(ldo >= (long)NH * 64 && ldo % 4 == 0, "flash_attn_fwd: bad ldo");
    static int qb_forced = -1;
  if (qb_forced < 0) { const char* e = getenv("VQ3_FLASH_QB"); qb_forced = e ? atoi(e) : 0; }
  // measured (tools/bench_flash.py): QB=2 +3 % at N = 1029 (fewer, fuller workgroups), -11 % at N = 8232 (occupancy 2 vs 3)
  const int qb = qb_forced ? qb_forced : ((N >= 512 && N < 4096) ? 2 : 1);
  dim3 grid((N + 128 * qb - 1) / (128 * qb), G * NH);
  if (qb == 2)
    hipLaunchKernelGGL(flash_attn_hd64_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, N, Np, NH, (long)ldo,
                       scale * 1.44269504088896340736f);
  else
    hipLaunchKernelGGL(flash_attn_hd64_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q,
                       (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, N, Np, NH, (long)ldo,
                       scale * 1.44269504088896340736f);
  VQ3_CHECK_LAUNCH("flash_attn_fwd");
  return 0;
}
