// Shared by the GEMM kernels: launch parameters, the XCD-aware tile order and the fused epilogue.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "vq3_hip.h"

namespace vq3gemm {

// Fused epilogue of the VGGT qkv projection (vq3_gemm_vit_qkv): instead of C the tile is written as head-major Q, K, V with the
// per-head LayerNorm and 2-D RoPE of the attention applied on the way (what vq3_vit_qkprep does in a pass of its own).
struct VitQkvEpi {
  bf16_t *Q, *K, *V;                       // [T / N, NH, N, 64]
  const float *qn_w, *qn_b, *kn_w, *kn_b;  // [64] each (use_norm)
  const bf16_t *cos, *sin;                 // [maxpos + 1, 32] (use_rope)
  int N, NH, P, patch_start, Wp, use_norm, use_rope;
  int rope_rows;   // rows of the cos / sin tables the epilogue can ask for: max(patch rows, patch columns) + 1 (host; <= 64: the 256 x 256 kernel keeps both tables in LDS)
  int m_off;       // token index of this launch's row 0 (the row-tail launch of gemm.hip: launch_split_rows starts past 0)
  float eps;
};

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;      // [N] or null
  const float* colscale;  // [N] or null  (LayerScale gamma)
  const void* R;          // residual, same dtype as C, or null
  int M, N, K, lda, ldb, ldc, ldr;
  long sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  int nb2, b2divB;
  int mtiles, ntiles;
  int xm;          // XCD blocking of the tile grid along M (1, 2, 4 or 8)
  int nbw;         // band width (n-tiles) of the walk inside an XCD's rectangle: n-fastest inside a band, then m, bands in turn
  int kper;        // split-K: K range per blockIdx.y slice (multiple of 64); == K when not split
  int nsplit;      // number of K slices (gridDim.y); > 1 => f32 output combined with atomics
  int act;         // 0 none, 1 gelu(erf), 2 silu
  int out_f32;     // C / R dtype: 0 bf16, 1 f32
  int accumulate;  // C += result
  int vec_ok;      // C (and R) rows are 16B (f32) / 8B (bf16) aligned for 4-wide stores
  float alpha;
  int epi;         // 0 = C / R epilogue; 1 = VitQkvEpi; 2 = SwiGLU backward; 3 = SwiGLU forward (staged path only; 3: gemm6.hip only)
  VitQkvEpi vit;
  const bf16_t* sw_gu;   // epi == 2: saved gate|up pre-activations [M, 2N]
  bf16_t* sw_dgu;        //           d(gate|up) [M, 2N]; the GEMM result d(act) [M, N] is never written
                         // epi == 3: N = 2 I columns gate | up; sw_dgu = gate|up out [M, 2 I], C / ldc = act out [M, I] = silu(gate) * up
  // LayerNorm folded into the GEMM that consumes it (staged epilogue only): A holds the raw rows x, B = gamma o W, bias = b + W.beta,
  // ln_c[n] = sum_k B[n,k]; the row statistics arrive as ln_parts (sum, sum of squares) pairs per row. y = rstd (acc - mu c) + bias.
  const float* ln_in;    // [M, ln_parts, 2] or null
  const float* ln_c;     // [N]
  int ln_parts;
  float ln_eps;
  // ... and produced for the NEXT LayerNorm by the GEMM that writes its input: per row and 128 output columns the (sum, sum of squares)
  // of the stored bf16 values, st_out [M, N / 128, 2] (N % 128 == 0, every slot written by exactly one thread: no atomics, no zeroing)
  float* st_out;
  // diagnostics (tools/gemm_stamps.py; null in production): per tile 8 x u64 of s_memrealtime / s_memtime stamps, written by thread 0
  unsigned long long* stamps;
  // start-up stagger (gemm6.hip): workgroups of the launch's first round wait (blockIdx.x % 8) * stagger ticks of the 100 MHz clock
  int stagger;
  // K split of the LAST, partly filled round of tiles over the CUs it would leave idle (gemm6.hip, 256 x 256 kernel, cfg 25): workgroup
  // sk_full + s * roundup8(sk_rem) + j multiplies K slice s (of sk_s) of tile sk_full + j; slices 0 .. sk_s - 2 leave their f32 accumulators in sk_ws
  // and count themselves in sk_cnt[j], slice sk_s - 1 (dispatched last) waits for that count, adds them and runs the epilogue. sk_s <= 1: off
  int sk_full, sk_rem, sk_s;
  unsigned sk_spin;      // bound of the reducer's wait in polls (1 << 23 in production; tests shrink it to walk the give-up path)
  // e4m3 operands (gemm6.hip F8 instantiations): fp32 scale per A row (token) and per B row (output channel); null = 1
  const float* f8_rs;
  const float* f8_cs;
  float* sk_ws;
  unsigned* sk_cnt;      // [sk_rem] arrival counts, zeroed per launch
  unsigned* sk_err;      // STICKY error word (host-mapped, one per process): set when a bounded wait gave up; cleared only when read
};
constexpr int SK_MAX_TILES = 128;

// row statistics of the folded LayerNorm (contraction length K is the normalised width)
__device__ __forceinline__ void ln_row(const GemmParams& p, int m, float& mu, float& rs) {
  m = m < p.M ? m : p.M - 1;
  const float* st = p.ln_in + (long)m * p.ln_parts * 2;
  float sm = 0.f, sq = 0.f;
  if (p.ln_parts == 8) {                        // width 1024: four 16-byte loads in flight together, summed in slot order
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(st + 4 * i);
#pragma unroll
    for (int i = 0; i < 4; ++i) { sm += v[i][0]; sq += v[i][1]; sm += v[i][2]; sq += v[i][3]; }
  } else {
    for (int i = 0; i < p.ln_parts; ++i) {      // fixed order: bit-identical from run to run
      const float2 v = *reinterpret_cast<const float2*>(st + 2 * i);
      sm += v.x; sq += v.y;
    }
  }
  const float inv = 1.f / (float)p.K;
  mu = sm * inv;
  rs = rsqrtf(fmaxf(sq * inv - mu * mu, 0.f) + p.ln_eps);
}
// The kernels fetch the statistics one thread per tile row at their start (two registers through the main loop), pass the (mu, rstd)
// pairs through LDS at the epilogue and every lane reads its own rows' pair from there. Loaded per lane instead, the 16 lanes that
// share a row would each request its 64 bytes - 256 KB of load traffic per 256-row tile through the texture path (+3.7 us per tile).
__device__ __forceinline__ f32x4 ln_colsum(const GemmParams& p, int n) {
  n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);     // columns past N are never stored
  return *reinterpret_cast<const f32x4*>(p.ln_c + n);
}
__device__ __forceinline__ f32x4 ln_apply(const f32x4& a, float mu, float rs, const f32x4& c) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = rs * (a[r] - mu * c[r]);
  return o;
}

template <bool F32_RESULT = false>          // (a result that stays f32 takes the 1e-5 form of GELU, one rounded to bf16 the cheaper 2.5e-4 one)
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return F32_RESULT ? gelu_erf(v) : gelu_erf_b(v);
  if (act == 2) return silu_f(v);
  return v;
}

// XCD-aware tile order. Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own 4 MiB
// L2), so workgroup b runs on XCD b % 8 as its (b / 8)-th tile. Each XCD is given one rectangle of an xm x (8/xm)
// blocking of the tile grid. Inside its rectangle an XCD walks BANDS of nbw n-tiles: n-fastest inside a band, then down
// m, one band after the other - so the c tiles an XCD has in flight together (consecutive positions: 32 CUs x workgroups per
// CU) form a (c / nbw) x nbw block that shares c / nbw row panels of A and nbw panels of B. Round 2's m-fastest walk put up to
// 25-48 m-tiles of ONE n-tile in flight: every A panel was fetched again for every n-tile (fc1, 49 392 x 4096 x 1024: 1.68 GB
// through the fabric per launch for 0.11 GB of operands, L2 hit rate 0.58: profiles/r3_gemm_fetch_by_shape.txt). The host picks
// (xm, nbw) from a traffic model (choose_tile_order). Bijective for any grid size; placement only ever affects speed.
__host__ __device__ __forceinline__ void tile_coords_id(const GemmParams& p, int orig, int BM, int BN, int& m0, int& n0) {
  const int nwg = p.mtiles * p.ntiles;
  const int xcd = orig & 7, idx = orig >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  int s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // position in the region-major sequence
  const int xm = p.xm, xn = 8 / p.xm;
  const int mc = (p.mtiles + xm - 1) / xm, nc = (p.ntiles + xn - 1) / xn;
  int mt = 0, ntl = 0;
  for (int reg = 0; reg < 8; ++reg) {
    const int xi = reg % xm, xj = reg / xm;
    int mcnt = p.mtiles - xi * mc; mcnt = mcnt < 0 ? 0 : (mcnt > mc ? mc : mcnt);
    int ncnt = p.ntiles - xj * nc; ncnt = ncnt < 0 ? 0 : (ncnt > nc ? nc : ncnt);
    const int sz = mcnt * ncnt;
    if (s < sz) {
      const int nb = p.nbw < 1 ? 1 : (p.nbw > ncnt ? ncnt : p.nbw);
      const int full = (ncnt / nb) * nb;            // columns covered by whole bands
      int band0, bw, rr;
      if (s < full * mcnt) {
        const int per = mcnt * nb;
        const int b = s / per;
        band0 = b * nb; bw = nb; rr = s - b * per;
      } else {
        band0 = full; bw = ncnt - full; rr = s - full * mcnt;
      }
      const int mi = rr / bw;
      mt = xi * mc + mi;
      ntl = xj * nc + band0 + (rr - mi * bw);
      break;
    }
    s -= sz;
  }
  m0 = mt * BM;
  n0 = ntl * BN;
}
__device__ __forceinline__ void tile_coords(const GemmParams& p, int BM, int BN, int& m0, int& n0) {
  tile_coords_id(p, blockIdx.x, BM, BN, m0, n0);
}

// Epilogue for one lane-owned quad C[m][n..n+3] (m < M, n < N guaranteed by the caller):
//   v = alpha*acc; += bias; (round to bf16 if C is bf16); act; *= colscale; += R; += C if accumulate; store.
// The intermediate roundings reproduce the points where PyTorch materialises a bf16 tensor.
template <bool OUT_F32>
__device__ __forceinline__ void store_quad(const GemmParams& p, long coff, long roff, int m, int n, const f32x4& a) {
  typedef typename std::conditional<OUT_F32, float, bf16_t>::type out_t;
  out_t* C = reinterpret_cast<out_t*>(p.C) + coff;
  const out_t* R = p.R ? reinterpret_cast<const out_t*>(p.R) + roff : nullptr;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
  const bool full = (n + 3 < p.N);
  const int nv = full ? 4 : (p.N - n);
  if (p.bias) {
    if (full) {  // n % 4 == 0 and the vectors are 16-byte aligned (checked on the host)
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bv[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nv) v[r] += p.bias[n + r];
    }
  }
  if (!OUT_F32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
  }
  if (p.act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = apply_act<OUT_F32>(v[r], p.act);
      if (!OUT_F32) v[r] = rbf(v[r]);
    }
  }
  if (p.colscale) {
    if (full) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(p.colscale + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] *= cv[r];
        if (!OUT_F32) v[r] = rbf(v[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nv) {
          v[r] *= p.colscale[n + r];
          if (!OUT_F32) v[r] = rbf(v[r]);
        }
    }
  }
  out_t* cp = C + (long)m * p.ldc + n;
  const out_t* rp = R ? R + (long)m * p.ldr + n : nullptr;
  if (OUT_F32 && p.nsplit > 1) {   // split-K partial: C was zeroed by the caller, slices meet by f32 atomics
    for (int r = 0; r < nv; ++r) atomicAdd(reinterpret_cast<float*>(cp) + r, v[r]);
    return;
  }
  if (full && p.vec_ok) {
    if (OUT_F32) {
      if (rp) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      }
      if (p.accumulate) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += cv[r];
      }
      *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      if (rp) {
        const u32x2 rv = *reinterpret_cast<const u32x2*>(rp);
        v[0] = rbf(v[0] + bf2f((bf16_t)(rv[0] & 0xffff)));
        v[1] = rbf(v[1] + bf2f((bf16_t)(rv[0] >> 16)));
        v[2] = rbf(v[2] + bf2f((bf16_t)(rv[1] & 0xffff)));
        v[3] = rbf(v[3] + bf2f((bf16_t)(rv[1] >> 16)));
      }
      if (p.accumulate) {
        const u32x2 cv = *reinterpret_cast<const u32x2*>(cp);
        v[0] += bf2f((bf16_t)(cv[0] & 0xffff));
        v[1] += bf2f((bf16_t)(cv[0] >> 16));
        v[2] += bf2f((bf16_t)(cv[1] & 0xffff));
        v[3] += bf2f((bf16_t)(cv[1] >> 16));
      }
      u32x2 o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *reinterpret_cast<u32x2*>(cp) = o;
    }
  } else {
    for (int r = 0; r < nv; ++r) {
      float x = v[r];
      if (OUT_F32) {
        if (rp) x += reinterpret_cast<const float*>(rp)[r];
        if (p.accumulate) x += reinterpret_cast<const float*>(cp)[r];
        reinterpret_cast<float*>(cp)[r] = x;
      } else {
        if (rp) x = rbf(x + bf2f(reinterpret_cast<const bf16_t*>(rp)[r]));
        if (p.accumulate) x += bf2f(reinterpret_cast<const bf16_t*>(cp)[r]);
        reinterpret_cast<bf16_t*>(cp)[r] = f2bf(x);
      }
    }
  }
}

// bf16 epilogue with operands fetched BEFORE the last K step (bias / LayerScale column vectors and the residual quad): the
// loads' round trip to L2 hides under the final MFMAs instead of stalling the stores. Same arithmetic and rounding
// points as store_quad; the caller guarantees n + 3 < N, vec_ok, bf16 output, no split-K.
__device__ __forceinline__ void store_quad_pre(const GemmParams& p, long coff, int m, int n, const f32x4& a, const f32x4& bias_v,
                                               const f32x4& cs_v, const u32x2& res_v, const u32x2* c_pre = nullptr) {
  bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + coff + (long)m * p.ldc + n;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bias_v[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
  if (p.act == 1) {
    const f32x2_t g0 = gelu_erf2_b(f32x2_t{v[0], v[1]}), g1 = gelu_erf2_b(f32x2_t{v[2], v[3]});
    v[0] = rbf(g0[0]); v[1] = rbf(g0[1]); v[2] = rbf(g1[0]); v[3] = rbf(g1[1]);
  } else if (p.act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(apply_act(v[r], p.act));
  }
  if (p.colscale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(v[r] * cs_v[r]);
  }
  if (p.R) {
    v[0] = rbf(v[0] + bf2f((bf16_t)(res_v[0] & 0xffff)));
    v[1] = rbf(v[1] + bf2f((bf16_t)(res_v[0] >> 16)));
    v[2] = rbf(v[2] + bf2f((bf16_t)(res_v[1] & 0xffff)));
    v[3] = rbf(v[3] + bf2f((bf16_t)(res_v[1] >> 16)));
  }
  if (p.accumulate) {
    const u32x2 cv = c_pre ? *c_pre : *reinterpret_cast<const u32x2*>(cp);
    v[0] += bf2f((bf16_t)(cv[0] & 0xffff));
    v[1] += bf2f((bf16_t)(cv[0] >> 16));
    v[2] += bf2f((bf16_t)(cv[1] & 0xffff));
    v[3] += bf2f((bf16_t)(cv[1] >> 16));
  }
  u32x2 o;
  o[0] = pack2bf(v[0], v[1]);
  o[1] = pack2bf(v[2], v[3]);
  *reinterpret_cast<u32x2*>(cp) = o;
}

// ---- LDS-staged bf16 epilogue ------------------------------------------------------------------------------------------------
// The MFMA accumulator layout gives a lane 4 consecutive columns of one row: stored straight from registers a wave
// instruction writes 16 rows x 32 B - four times the L2 requests of a coalesced store, and at short K (the VGGT GEMMs:
// 16 K steps per tile) the output write is what the tile waits for (a 4096 x 4096 x 128 launch ran at 1.2 TB/s of C).
// Instead the finished values (alpha, bias, activation, LayerScale applied, rounded to bf16 where PyTorch rounds) go to an
// LDS image of the C tile - rows of BN bf16, 16-byte chunks XOR (row & 15): conflict-free for the 8-byte quad writes
// (16 lanes = 16 rows, one column group) and for the 16-byte row reads - and leave as whole rows: 16 B per lane,
// BN * 2 contiguous bytes per row, with the residual / accumulate operand read the same coalesced way.
// lane exchange inside groups of 8 lanes on the DPP path of the VALU (no LDS-permute traffic next to the staged tile's reads)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {          // all-reduce over the 8 lanes l & ~7 .. l | 7
  v += dpp_mov<0xB1>(v);                                  // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(v);                                  // quad_perm [2,3,0,1]: lane ^ 2
  return v + dpp_mov<0x141>(v);                           // row_half_mirror: lane i <-> 7 - i, i.e. the other quad's sum
}
__device__ __forceinline__ float sum16(float v) {         // all-reduce over the 16 lanes of a DPP row
  v = sum8(v);
  return v + dpp_mov<0x140>(v);                           // row_mirror: lane i <-> 15 - i, i.e. the other half-row's sum
}

constexpr int EPI_U = 8;     // rows per thread whose global operands are in flight together in the staged epilogues
template <int BN>
__device__ __forceinline__ int cstage_off(int row, int chunk) {
  constexpr int MASK = BN / 8 >= 16 ? 15 : BN / 8 - 1;     // rows of fewer than 16 chunks (BN = 64): 2-way on the quad writes
  return row * (BN * 2) + ((chunk ^ (row & MASK)) << 4);
}

// phase 1, per lane-owned quad (tile-local row ml, column nl..nl+3): same arithmetic and rounding points as store_quad_pre up
// to (and including) LayerScale; bias_v / cs_v are the lane's column vectors (prefetched), ignored when p.bias / p.colscale is null
// ACT >= 0: the activation is a compile-time constant (the caller dispatches on p.act ONCE, outside its quad loops: with the run-time
// form every one of a thread's 32 quads carries all three activation bodies, and the 256 x 256 kernel's epilogue was 22 000 instructions
// for an 8 000-instruction instruction cache line budget); -1: run-time p.act
// MODE >= 0: which of alpha / bias / LayerScale apply is a compile-time constant too - bit 0 bias, bit 1 LayerScale, bit 2 alpha != 1
// (the caller dispatches ONCE per tile on the combinations its launches use; -1: run-time, from p). Round 4: with the run-time form
// hipcc turned the wave-uniform `if (p.bias)` / `if (p.colscale)` into per-element v_cndmask selects and every intermediate rounding was
// followed by a second conversion at the final pack: 45 vector instructions per quad with no activation at all (stage C: 4-6 us of a
// 45 us tile at K = 1024). The roundings themselves are unchanged: after the bias, after the activation, after LayerScale - the LAST of
// them is the pack's own v_cvt_pk_bf16_f32.
template <int BN, int ACT = -1, int MODE = -1>
__device__ __forceinline__ void stage_quad(const GemmParams& p, char* smem, int ml, int nl, const f32x4& a, const f32x4& bias_v,
                                           const f32x4& cs_v) {
  const int act = ACT >= 0 ? ACT : p.act;
  const bool has_bias = MODE >= 0 ? (MODE & 1) != 0 : p.bias != nullptr;
  const bool has_cs = MODE >= 0 ? (MODE & 2) != 0 : p.colscale != nullptr;
  const bool unit_alpha = MODE >= 0 ? (MODE & 4) == 0 : false;
  f32x2_t v0 = {a[0], a[1]}, v1 = {a[2], a[3]};
  if (!unit_alpha) { v0 *= p.alpha; v1 *= p.alpha; }
  if (has_bias) { v0 += f32x2_t{bias_v[0], bias_v[1]}; v1 += f32x2_t{bias_v[2], bias_v[3]}; }
  if (act || has_cs) {
    v0 = rbf2(v0); v1 = rbf2(v1);
    if (act == 1) { v0 = gelu_erf2_b(v0); v1 = gelu_erf2_b(v1); }
    else if (act) { v0 = f32x2_t{apply_act(v0[0], act), apply_act(v0[1], act)}; v1 = f32x2_t{apply_act(v1[0], act), apply_act(v1[1], act)}; }
    if (has_cs) {
      if (act) { v0 = rbf2(v0); v1 = rbf2(v1); }
      v0 *= f32x2_t{cs_v[0], cs_v[1]}; v1 *= f32x2_t{cs_v[2], cs_v[3]};
    }
  }
  u32x2 o;
  o[0] = pack2bf(v0[0], v0[1]);
  o[1] = pack2bf(v1[0], v1[1]);
  *reinterpret_cast<u32x2*>(smem + cstage_off<BN>(ml, nl >> 3) + ((nl & 4) << 1)) = o;
}
// the (ACT, MODE) pairs the launches of a step use get a compiled form each; anything else takes the run-time form
#define VQ3_STAGE_DISPATCH(P, CALL)                                                                           \
  do {                                                                                                        \
    const int mode__ = ((P).bias ? 1 : 0) | ((P).colscale ? 2 : 0) | ((P).alpha != 1.f ? 4 : 0);              \
    if ((P).act == 0 && mode__ == 0) CALL(0, 0);                                                              \
    else if ((P).act == 0 && mode__ == 1) CALL(0, 1);                                                         \
    else if ((P).act == 1 && mode__ == 1) CALL(1, 1);                                                         \
    else if ((P).act == 0 && mode__ == 3) CALL(0, 3);                                                         \
    else CALL(-1, -1);                                                                                        \
  } while (0)

template <int BM, int BN>
__device__ __forceinline__ void vit_qkv_store(const GemmParams& p, const char* smem, int m0, int n0, int tid, int nthreads, const char* rowinfo = nullptr,
                                              const char* ropetab = nullptr);

// phase 2 (after a workgroup barrier): `nthreads` threads (tid 0 .. nthreads-1) move the BM x BN image out as whole rows,
// adding the residual (rounded, as PyTorch's bf16 add) and / or the old C (accumulate) on the way. Needs p.vec_ok, ldc % 8 == 0
// and a 16-byte aligned C (checked by staged_ok on the host side of the kernel).
// EPIK >= 0: only that epilogue kind is compiled in (the caller guarantees p.epi == EPIK); -1: run-time dispatch
// The residual / old-C operands of a thread's FIRST batch of rows (the rows staged_store hands it first), requested before the C tile is
// staged: their round trip - every CU of a round asks within the same microseconds - runs under the staging instead of after it
// (tools/gemm_stamps.py, proj 49 392 x 1024 x 1024 + residual: "issue stores" 13.8 us per tile against 3.2 us without a residual).
struct EpiPre {
  u32x4 v[EPI_U];      // the residual rows, or the old C rows of an accumulating launch (a launch with both is not prefetched)
  bool on;
};
template <int BM, int BN>
__device__ __forceinline__ void staged_load_rows(const GemmParams& p, const bf16_t* src, long ld, int m0, int n, int row0, int rpp,
                                                 u32x4 (&v)[EPI_U]) {
#pragma unroll
  for (int u = 0; u < EPI_U; ++u) {
    const int row = row0 + u * rpp, m = m0 + row;
    const long mc = (row < BM && m < p.M) ? m : p.M - 1;
    v[u] = *reinterpret_cast<const u32x4*>(src + mc * ld + n);
  }
}
template <int BM, int BN>
__device__ __forceinline__ void staged_prefetch(const GemmParams& p, long coff, long roff, int m0, int n0, int tid, int nthreads, EpiPre& pre) {
  constexpr int CPR = BN / 8;
  const int n = n0 + (tid % CPR) * 8;
  pre.on = p.epi == 0 && ((p.R != nullptr) != (p.accumulate != 0)) && n + 7 < p.N;
  if (!pre.on) return;
  if (p.R) staged_load_rows<BM, BN>(p, reinterpret_cast<const bf16_t*>(p.R) + roff, p.ldr, m0, n, tid / CPR, nthreads / CPR, pre.v);
  else staged_load_rows<BM, BN>(p, reinterpret_cast<const bf16_t*>(p.C) + coff, p.ldc, m0, n, tid / CPR, nthreads / CPR, pre.v);
}

template <int BM, int BN, int EPIK = -1>
__device__ __forceinline__ void staged_store(const GemmParams& p, const char* smem, long coff, long roff, int m0, int n0, int tid,
                                             int nthreads, const EpiPre* pre = nullptr, const char* rowinfo = nullptr, const char* ropetab = nullptr) {
  constexpr int CPR = BN / 8;                  // 16-byte chunks per row
  if constexpr (EPIK == 4) {
    // SwiGLU forward (modeling_qwen3.py:81-83): the tile's columns are [BN/2 gate | the same BN/2 up columns] (the kernel stages the
    // two weight row ranges side by side), so a thread holds gate and up of 8 features of one row: both leave to gate|up [M, 2 I] as
    // the backward reads them, act = bf16(bf16(silu(g)) * u) to C - arithmetic and rounding points of silu_mul_fwd_kernel.
    constexpr int HC = BN / 16;                // chunks per half row
    const int inter = p.N >> 1;
    const int rpp = nthreads / HC, c = tid % HC, n = (n0 >> 1) + c * 8;
    if (n >= inter) return;                    // I % 8 == 0 (host)
    bf16_t* act = reinterpret_cast<bf16_t*>(p.C) + coff;
    for (int row = tid / HC; row < BM; row += rpp) {
      const int m = m0 + row;
      if (m >= p.M) break;
      const u32x4 gv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row, c));
      const u32x4 uv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row, c + HC));
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float r[2];
#pragma unroll
        for (int hgh = 0; hgh < 2; ++hgh) {
          const float gf = bf2f((bf16_t)((gv[k] >> (16 * hgh)) & 0xffff));
          const float uf = bf2f((bf16_t)((uv[k] >> (16 * hgh)) & 0xffff));
          r[hgh] = rbf(silu_f(gf)) * uf;
        }
        o[k] = pack2bf(r[0], r[1]);
      }
      if (p.sw_dgu) {                            // (null under no_grad: nobody reads gate|up again)
        bf16_t* gp = p.sw_dgu + (long)m * p.N + n;
        *reinterpret_cast<u32x4*>(gp) = gv;
        *reinterpret_cast<u32x4*>(gp + inter) = uv;
      }
      *reinterpret_cast<u32x4*>(act + (long)m * p.ldc + n) = o;
    }
    return;
  }
  if ((EPIK < 0 || EPIK == 1) && p.epi == 1) {
    vit_qkv_store<BM, BN>(p, smem, m0, n0, tid, nthreads, rowinfo, ropetab);
    return;
  }
  if ((EPIK < 0 || EPIK == 2) && p.epi == 2) {
    // SwiGLU backward on the way out (modeling_qwen3.py:81-83 under autograd): the tile is d(act) = d(silu(g) * u); with the saved
    // g | u rows read the same coalesced way it leaves as dg = d * u * silu'(g) and du = d * silu(g) - arithmetic of silu_mul_bwd_kernel
    constexpr int CPR2 = BN / 8;
    const int rpp = nthreads / CPR2, c = tid % CPR2, n = n0 + c * 8;
    if (n >= p.N) return;                               // N % 8 == 0 (checked on the host)
    for (int row0 = tid / CPR2; row0 < BM; row0 += rpp * EPI_U) {
      // the saved operands of EPI_U rows are requested first: one L2 / HBM round trip per batch instead of one per row
      u32x4 gv[EPI_U], uv[EPI_U];
      bool ok[EPI_U];
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) {
        const int row = row0 + u * rpp, m = m0 + row;
        ok[u] = row < BM && m < p.M;
        const bf16_t* gp = p.sw_gu + (long)(ok[u] ? m : p.M - 1) * 2 * p.N + n;
        gv[u] = *reinterpret_cast<const u32x4*>(gp);
        uv[u] = *reinterpret_cast<const u32x4*>(gp + p.N);
      }
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) {
        if (!ok[u]) continue;
        const int row = row0 + u * rpp, m = m0 + row;
        const u32x4 dv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row, c));
        u32x4 og, ou;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float r_g[2], r_u[2];
#pragma unroll
          for (int hgh = 0; hgh < 2; ++hgh) {
            const float df = bf2f((bf16_t)((dv[k] >> (16 * hgh)) & 0xffff));
            const float gf = bf2f((bf16_t)((gv[u][k] >> (16 * hgh)) & 0xffff));
            const float uf = bf2f((bf16_t)((uv[u][k] >> (16 * hgh)) & 0xffff));
            const float sg = sigmoid_f(gf);
            r_g[hgh] = df * uf * (sg * (1.f + gf * (1.f - sg)));
            r_u[hgh] = df * (gf * sg);
          }
          og[k] = pack2bf(r_g[0], r_g[1]);
          ou[k] = pack2bf(r_u[0], r_u[1]);
        }
        bf16_t* op = p.sw_dgu + (long)m * 2 * p.N + n;
        *reinterpret_cast<u32x4*>(op) = og;
        *reinterpret_cast<u32x4*>(op + p.N) = ou;
      }
    }
    return;
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + coff;
  const bf16_t* R = p.R ? reinterpret_cast<const bf16_t*>(p.R) + roff : nullptr;
  const int rpp = nthreads / CPR;              // rows per pass
  const int c = tid % CPR;
  const int n = n0 + c * 8;
  if (n >= p.N) return;
  const bool fulln = n + 7 < p.N;
  if (fulln) {
    // residual / old-C rows of EPI_U passes are requested before any of them is used, and the NEXT batch's before this batch's stores
    // (which the compiler must assume alias them): the round trips overlap the arithmetic and the stores of the batch before
    u32x4 rv[EPI_U], cv[EPI_U];
    const bool pre_r = pre && pre->on && R, pre_c = pre && pre->on && !R;
    if (pre_r) {
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) rv[u] = pre->v[u];
    } else if (R) {
      staged_load_rows<BM, BN>(p, R, p.ldr, m0, n, tid / CPR, rpp, rv);
    }
    if (pre_c) {
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) cv[u] = pre->v[u];
    } else if (p.accumulate) {
      staged_load_rows<BM, BN>(p, C, p.ldc, m0, n, tid / CPR, rpp, cv);
    }
    for (int row0 = tid / CPR; row0 < BM; row0 += rpp * EPI_U) {
      u32x4 rn[EPI_U], cn[EPI_U];
      const bool nxt = row0 + rpp * EPI_U < BM;
      if (nxt && R) staged_load_rows<BM, BN>(p, R, p.ldr, m0, n, row0 + rpp * EPI_U, rpp, rn);
      if (nxt && p.accumulate) staged_load_rows<BM, BN>(p, C, p.ldc, m0, n, row0 + rpp * EPI_U, rpp, cn);
#pragma unroll
      for (int u = 0; u < EPI_U; ++u) {
        const int row = row0 + u * rpp, m = m0 + row;
        if (!(row < BM && m < p.M)) continue;
        const u32x4 sv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row, c));
        u32x4 o = sv;
        if (R || p.accumulate) {
          // (a pair per step; the LAST rounding is the pack's own conversion: rbf(x) packed again is x's packing)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            f32x2_t v = unpack2bf(sv[k]);
            if (R) {
              v += unpack2bf(rv[u][k]);
              if (p.accumulate) v = rbf2(v);
            }
            if (p.accumulate) v += unpack2bf(cv[u][k]);
            o[k] = pack2bf(v[0], v[1]);
          }
        }
        *reinterpret_cast<u32x4*>(C + (long)m * p.ldc + n) = o;
        if (p.st_out) {
          // (sum, sum of squares) of the 128 stored values around this thread's 8: the 16 lanes of a 16-lane row hold them
          float sm = 0.f, sq = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lo = bf2f((bf16_t)(o[k] & 0xffff)), hi = bf2f((bf16_t)(o[k] >> 16));
            sm += lo + hi;
            sq = fmaf(lo, lo, fmaf(hi, hi, sq));
          }
          sm = sum16(sm); sq = sum16(sq);
          if ((c & 15) == 0)
            *reinterpret_cast<float2*>(p.st_out + ((long)m * (p.N >> 7) + (n >> 7)) * 2) = float2{sm, sq};
        }
      }
      if (nxt) {
#pragma unroll
        for (int u = 0; u < EPI_U; ++u) { rv[u] = rn[u]; cv[u] = cn[u]; }
      }
    }
    return;
  }
  for (int row = tid / CPR; row < BM; row += rpp) {      // ragged last chunk of a row (N % 8 != 0): element-wise
    const int m = m0 + row;
    if (m >= p.M) break;
    const u32x4 sv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row, c));
    bf16_t* cp = C + (long)m * p.ldc + n;
    for (int k = 0; k < 8 && n + k < p.N; ++k) {
      float x = bf2f((bf16_t)((sv[k >> 1] >> ((k & 1) * 16)) & 0xffff));
      if (R) x = rbf(x + bf2f(R[(long)m * p.ldr + n + k]));
      if (p.accumulate) x += bf2f(cp[k]);
      cp[k] = f2bf(x);
    }
  }
}

// epi == 1: the rows of the staged tile leave as Q / K / V [g, head, token, 64]. A thread owns 8 consecutive features of one
// head; the head's 64 features are the 8 consecutive lanes around it (CPR is a multiple of 8 and tiles start at multiples of
// 64 columns), so the LayerNorm sums are three xor-shuffles and the rotate-half partner (feature e ^ 16) is lane ^ 2.
// Arithmetic and rounding points are those of vit_qkprep4_kernel (vggt.hip).
// x = q * d + r for 0 <= x < 2^22 and a small divisor, through the float reciprocal with one correction step either way
__device__ __forceinline__ void divmod_f(int x, int d, float inv, int& q, int& r) {
  q = (int)((float)x * inv);
  r = x - q * d;
  if (r < 0) { r += d; --q; }
  else if (r >= d) { r -= d; ++q; }
}

// What the fused q|k|v store needs to know about a ROW of the tile, worked out once per row (vit_row_info, one thread per row, into
// LDS) instead of once per (row, 16-byte chunk) by each of the BN / 8 threads that share the row: three divisions by run-time
// divisors through the float reciprocal are ~30 VALU instructions, and a 256 x 256 tile asked for them 16 times per thread.
//   x: the row's element offset ((group * NH) * N + token) * 64 without the head term, low / high word (-1 / -1: row past M);
//   z, w: 32 * (patch row + 1), 32 * (patch column + 1) - the rows of the cos / sin tables (0 for the special tokens)
__device__ __forceinline__ int4 vit_row_info(const GemmParams& p, int m) {
  const VitQkvEpi& e = p.vit;
  if (m >= p.M) return int4{-1, -1, 0, 0};
  const float invN = 1.f / (float)e.N, invP = 1.f / (float)(e.P > 0 ? e.P : 1), invW = 1.f / (float)(e.Wp > 0 ? e.Wp : 1);
  int gi, ti;
  divmod_f(m + e.m_off, e.N, invN, gi, ti);
  int py = 0, px = 0;
  if (e.use_rope) {
    int fr, tp = ti;
    if (e.P != e.N) divmod_f(ti, e.P, invP, fr, tp);
    if (tp >= e.patch_start) {
      divmod_f(tp - e.patch_start, e.Wp, invW, py, px);
      ++py; ++px;
    }
  }
  const long ob = ((long)gi * e.NH * e.N + ti) * 64;
  return int4{(int)(ob & 0xffffffffL), (int)(ob >> 32), py * 32, px * 32};
}

template <int BM, int BN>
__device__ __forceinline__ void vit_qkv_store(const GemmParams& p, const char* smem, int m0, int n0, int tid, int nthreads, const char* rowinfo,
                                              const char* ropetab) {        // ropetab: LDS copies of the cos (at 0) and sin (at 4096) tables, or null
  constexpr int CPR = BN / 8;
  const VitQkvEpi& e = p.vit;
  const int C = e.NH * 64;
  const int rpp = nthreads / CPR;
  const int c = tid % CPR;
  const int n = n0 + c * 8;
  if (n >= p.N) return;                                  // N = 3 C is a multiple of 64: a head is in or out as a whole
  const int which = n / C, rem = n - which * C, head = rem >> 6, d0 = rem & 63;
  const bool norm = which < 2 && e.use_norm, rope = which < 2 && e.use_rope;
  float w8[8], b8[8];
  if (norm) {
    const float* wp = (which == 0 ? e.qn_w : e.kn_w) + d0;
    const float* bp = (which == 0 ? e.qn_b : e.kn_b) + d0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { w8[j] = wp[j]; b8[j] = bp[j]; }
  }
  bf16_t* base = which == 0 ? e.Q : (which == 1 ? e.K : e.V);
  const bool neg = (d0 & 16) == 0;                       // (e & 16) == 0 -> rotate-half takes -x[e + 16]
  const float invN = 1.f / (float)e.N, invP = 1.f / (float)(e.P > 0 ? e.P : 1), invW = 1.f / (float)(e.Wp > 0 ? e.Wp : 1);
  for (int row0 = tid / CPR; row0 < BM; row0 += rpp * EPI_U) {
    // token coordinates and the cos / sin rows of EPI_U passes first (they depend on the row only), then the arithmetic: the table
    // reads' round trip is paid once per batch. Rows past M are computed on row M - 1 and not stored, so all lanes stay in step.
    long ob[EPI_U];                                      // element offset of the row's head 0, feature 0
    bool ok[EPI_U];
    u32x4 cr[EPI_U], sr[EPI_U];
#pragma unroll
    for (int u = 0; u < EPI_U; ++u) {
      const int row = row0 + u * rpp, m = m0 + row;
      ok[u] = row < BM && m < p.M;
      int pos32;
      if (rowinfo) {
        const int4 ri = *reinterpret_cast<const int4*>(rowinfo + 16 * (row < BM ? row : BM - 1));
        ob[u] = ((long)ri.y << 32) | (unsigned)ri.x;
        pos32 = d0 < 32 ? ri.z : ri.w;
      } else {
        int gi, ti;
        divmod_f((ok[u] ? m : p.M - 1) + e.m_off, e.N, invN, gi, ti);
        ob[u] = ((long)gi * e.NH * e.N + ti) * 64;
        int py = 0, px = 0;
        if (rope) {
          int fr, tp = ti;
          if (e.P != e.N) divmod_f(ti, e.P, invP, fr, tp);
          if (tp >= e.patch_start) {
            divmod_f(tp - e.patch_start, e.Wp, invW, py, px);
            ++py; ++px;
          }
        }
        pos32 = (d0 < 32 ? py : px) * 32;
      }
      if (rope) {
        if (ropetab) {
          cr[u] = *reinterpret_cast<const u32x4*>(ropetab + 2 * (pos32 + (d0 & 31)));
          sr[u] = *reinterpret_cast<const u32x4*>(ropetab + 4096 + 2 * (pos32 + (d0 & 31)));
        } else {
          cr[u] = *reinterpret_cast<const u32x4*>(e.cos + pos32 + (d0 & 31));
          sr[u] = *reinterpret_cast<const u32x4*>(e.sin + pos32 + (d0 & 31));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < EPI_U; ++u) {
      const int row = row0 + u * rpp;
      const u32x4 sv = *reinterpret_cast<const u32x4*>(smem + cstage_off<BN>(row < BM ? row : BM - 1, c));
      // Same arithmetic and rounding points as before, in fewer VALU issue slots: roundings through bf16 pair-wise (one conversion + two
      // unpacks per pair instead of a conversion + unpack per value), none before the final pack (the conversion is idempotent), the
      // rotate-half sign folded into the packed sine words once per row ((-p) s == p (-s) bit for bit), and the v columns - no norm, no
      // rotation - leave as the staged dwords they are.
      u32x4 o = sv;
      if (norm || rope) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = bf2f((bf16_t)(sv[k] & 0xffff)); x[2 * k + 1] = bf2f((bf16_t)(sv[k] >> 16)); }
        if (norm) {
          const float mean = sum8(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 64.f);
          float sq = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { x[j] -= mean; sq = fmaf(x[j], x[j], sq); }
          const float rs = rsqrtf(sum8(sq) * (1.f / 64.f) + e.eps);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float t0 = x[2 * k] * rs * w8[2 * k] + b8[2 * k], t1 = x[2 * k + 1] * rs * w8[2 * k + 1] + b8[2 * k + 1];
            if (rope) {
              const f32x2_t t = rbf2(f32x2_t{t0, t1});
              x[2 * k] = t[0]; x[2 * k + 1] = t[1];
            } else {
              o[k] = pack2bf(t0, t1);
            }
          }
        }
        if (rope) {
          const unsigned sflip = neg ? 0x80008000u : 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned cw = cr[u][k], sw = sr[u][k] ^ sflip;
            const float p0 = dpp_mov<0x4E>(x[2 * k]), p1 = dpp_mov<0x4E>(x[2 * k + 1]);              // feature e ^ 16: lane ^ 2
            const f32x2_t a = rbf2(f32x2_t{x[2 * k] * bf2f((bf16_t)(cw & 0xffff)), x[2 * k + 1] * bf2f((bf16_t)(cw >> 16))});
            const f32x2_t b = rbf2(f32x2_t{p0 * bf2f((bf16_t)(sw & 0xffff)), p1 * bf2f((bf16_t)(sw >> 16))});
            o[k] = pack2bf(a[0] + b[0], a[1] + b[1]);
          }
        }
      }
      if (!ok[u]) continue;
      *reinterpret_cast<u32x4*>(base + ob[u] + (long)head * e.N * 64 + d0) = o;
    }
  }
}

__device__ __forceinline__ bool staged_ok(const GemmParams& p, long coff, long roff) {
  return !p.out_f32 && p.nsplit == 1 && p.vec_ok && (p.ldc % 8 == 0) && (coff % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
         (!p.R || ((p.ldr % 8 == 0) && (roff % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.R) & 15) == 0)));
}

// host twin of staged_ok for every batch index (the v6 kernels compile ONLY the staged bf16 epilogue: gemm6.hip routes the rest elsewhere)
inline bool host_staged_ok(const GemmParams& p) {
  return !p.out_f32 && p.nsplit == 1 && p.vec_ok && (p.ldc % 8 == 0) && (p.sC1 % 8 == 0) && (p.sC2 % 8 == 0) &&
         ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
         (!p.R || ((p.ldr % 8 == 0) && (p.sR1 % 8 == 0) && (p.sR2 % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.R) & 15) == 0)));
}

// host: pick the XCD blocking (xm) and the band width (nbw) of the walk for a tile grid. Model of the bytes one XCD pulls through
// the fabric: its rectangle is mR x nR tiles, `conc` of them in flight together as a (conc / nbw) x nbw block; a row panel of A
// (BM x K) is fetched once per band it appears in, a panel of B (BN x K) once per block row of a band (nothing is assumed to survive
// a round in the 4 MiB L2: per round the block streams its A panels and writes its C tiles through it):
//   bytes = BM K 2 * mR * ceil(nR / nbw)  +  BN K 2 * nR * ceil(mR / (conc / nbw))
// VQ3_GEMM_XM / VQ3_GEMM_BAND override (A/B runs); VQ3_GEMM_BAND=0 restores round 2's m-fastest walk (nbw = 1 with xm by panel count).
inline int choose_xm(int mtiles, int ntiles) {
  int best = 1;
  double bc = 1e30;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const double c = (double)mtiles / xm + (double)ntiles / (8 / xm);
    if (c < bc - 1e-9) { bc = c; best = xm; }
  }
  return best;
}
inline void choose_tile_order(GemmParams& p, int BM, int BN, int wg_per_cu) {
  static int env_band = -2, env_xm = -2;
  if (env_band == -2) { const char* e = getenv("VQ3_GEMM_BAND"); env_band = e ? atoi(e) : -1; }
  if (env_xm == -2) { const char* e = getenv("VQ3_GEMM_XM"); env_xm = e ? atoi(e) : -1; }
  if (env_band == 0) { p.xm = choose_xm(p.mtiles, p.ntiles); p.nbw = 1; return; }
  if (env_band < 0 && env_xm < 0 && p.mtiles >= 64 && p.mtiles >= 6 * p.ntiles) {
    // Tall outputs (the tower at 49 392+ rows x 1024 .. 4096 columns): measured (tools/sweep_tile_order.sh, 4 XCD splits x 5 band widths) the
    // best walk gives every XCD its own strip of M - the activation panels, 100+ MB, cross the fabric once; the weight panels, 2-8 MB and
    // resident in the Infinity Cache, are what gets re-read - in bands of 2 n-tiles when N is wide: q|k|v 425 -> 412 us, fc1 519 -> 509 us
    // against the byte-count model below, which weighs A and B bytes alike and splits N across XCDs for these shapes
    p.xm = 8;
    p.nbw = p.ntiles >= 8 ? 2 : p.ntiles;
    return;
  }
  const int conc = 32 * (wg_per_cu < 1 ? 1 : wg_per_cu);
  double best = 1e300;
  int bxm = 1, bnb = 1;
  for (int xm = 1; xm <= 8; xm *= 2) {
    if (env_xm > 0 && xm != env_xm) continue;
    const int xn = 8 / xm;
    const int mR = (p.mtiles + xm - 1) / xm, nR = (p.ntiles + xn - 1) / xn;
    for (int nb = 1; nb <= 64; nb *= 2) {
      if (env_band > 0 && nb != env_band) continue;
      const int nbe = nb > nR ? nR : nb;
      int a = conc / nbe; a = a < 1 ? 1 : a;
      const double bytes = (double)BM * mR * ((nR + nbe - 1) / nbe) + (double)BN * nR * ((mR + a - 1) / a);
      // (regions that are not full - a grid smaller than the chip - cost what they cost: the model only ranks)
      if (bytes < best - 1e-9) { best = bytes; bxm = xm; bnb = nbe; }
      if (nb >= nR) break;
    }
  }
  p.xm = bxm; p.nbw = bnb;
}

// v2 (LDS-DMA pipelined) launcher, defined in gemm2.hip. cfg: 0 = 256x128 tile, 1 = 128x256, 2 = 128x128.
int launch_gemm_v2(GemmParams& p, int cfg, int nbatch, hipStream_t stream);
// v6 (256x256 tile, 8-phase schedule, NT, K % 64 == 0), defined in gemm6.hip.
int launch_gemm_v6(GemmParams& p, int shape, int nbatch, hipStream_t stream);   // shape: 0 = 256x256, 1 = 256x128, 2 = 128x256,
// 3 = 256x256 with the last round's tiles split along K (returns -1, nothing launched, where that does not apply)
// e4m3 operands on the 256 x 256 8-phase kernel (gemm6.hip; p.f8_rs / p.f8_cs, p.epi 0 / 2 / 3): -1 = outside its contract, nothing launched
int launch_gemm_v6_f8(GemmParams& p, hipStream_t stream);
// both operands k-major on the 256 x 256 8-phase kernel (gemm6.hip KM instantiations; cfg 106 / 107 = with the last round split): -1 = outside
int launch_gemm_v6_km(GemmParams& p, int nbatch, hipStream_t stream, bool split);
// host side of cfg 25 (gemm6.hip): the split plan, and the error word of the stream's workspace (-1: no split launch ever ran on it)
int gemm_split_plan(const GemmParams& p, int nbatch, int ncu, int* full, int* rem);
int gemm_split_gave_up(hipStream_t stream);
int gemm_split_poll(bool clear);
void gemm_split_set_spin_bound(unsigned polls);
void gemm_split_set_provider(vq3_ws_provider_t fn);
// v7 (256x128 tile, four waves, two workgroups per CU: a tile's epilogue runs under the co-resident workgroup's main loop), gemm7.hip.
// Returns -1 (nothing launched) when the problem is outside its contract: the caller picks another kernel.
int launch_gemm_v7(GemmParams& p, int nbatch, hipStream_t stream);
// v3 (any operand layout, K % 8 == 0), defined in gemm3.hip. nstage: 2 or 3.
int launch_gemm_v3(GemmParams& p, int transA, int transB, int nstage, int nbatch, hipStream_t stream);

}  // namespace vq3gemm
