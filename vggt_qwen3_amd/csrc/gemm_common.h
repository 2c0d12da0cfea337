// Shared by the GEMM kernels: launch parameters, the XCD-aware tile order and the fused epilogue.
#pragma once
#include <type_traits>

#include "common.h"

namespace vq3gemm {

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
  int kper;        // split-K: K range per blockIdx.y slice (multiple of 64); == K when not split
  int nsplit;      // number of K slices (gridDim.y); > 1 => f32 output combined with atomics
  int act;         // 0 none, 1 gelu(erf), 2 silu
  int out_f32;     // C / R dtype: 0 bf16, 1 f32
  int accumulate;  // C += result
  int vec_ok;      // C (and R) rows are 16B (f32) / 8B (bf16) aligned for 4-wide stores
  float alpha;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return gelu_erf(v);
  if (act == 2) return silu_f(v);
  return v;
}

// XCD-aware tile order. Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own 4 MiB
// L2), so workgroup b runs on XCD b % 8 as its (b / 8)-th tile. Each XCD is given one rectangle of an xm x (8/xm)
// blocking of the tile grid (walked m-fastest): the tiles sharing an L2 then re-read only mtiles/xm row panels of A
// and ntiles/xn panels of B. The host picks xm in {1,2,4,8} minimising mtiles/xm + ntiles/xn (xm = 1 is "n-tile
// major" order). Bijective for any grid size; placement only ever affects speed.
__device__ __forceinline__ void tile_coords_id(const GemmParams& p, int orig, int BM, int BN, int& m0, int& n0) {
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
      mt = xi * mc + s % mcnt;
      ntl = xj * nc + s / mcnt;
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
      v[r] = apply_act(v[r], p.act);
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
  if (p.act) {
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

// host: pick the XCD blocking for a tile grid
inline int choose_xm(int mtiles, int ntiles) {
  int best = 1;
  double bc = 1e30;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const double c = (double)mtiles / xm + (double)ntiles / (8 / xm);
    if (c < bc - 1e-9) { bc = c; best = xm; }
  }
  return best;
}

// v2 (LDS-DMA pipelined) launcher, defined in gemm2.hip. cfg: 0 = 256x128 tile, 1 = 128x256, 2 = 128x128.
int launch_gemm_v2(GemmParams& p, int cfg, int nbatch, hipStream_t stream);
// v6 (256x256 tile, 8-phase schedule, NT, K % 64 == 0), defined in gemm6.hip.
int launch_gemm_v6(GemmParams& p, int nbatch, hipStream_t stream);
// v3 (any operand layout, K % 8 == 0), defined in gemm3.hip. nstage: 2 or 3.
int launch_gemm_v3(GemmParams& p, int transA, int transB, int nstage, int nbatch, hipStream_t stream);

}  // namespace vq3gemm
