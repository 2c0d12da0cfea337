// Qwen3 attention pre-processing: split the fused q|k|v projection, per-head RMSNorm on q and k
// (modeling_qwen3.py:237-238,252-253), rotate-half RoPE (modeling_qwen3.py:104-148), and the layout change
// [B*L, heads*D] -> [B, heads, L, D] that the batched attention GEMMs consume. head_dim is 128: one wave owns one
// head vector, lane i holds elements i and i+64 - exactly the rotate_half pair, so RoPE needs no cross-lane
// traffic and the RMS reduction is a single wave_sum.
#include "common.h"
#include "vq3_hip.h"

namespace {

constexpr int D = 128;

__global__ __launch_bounds__(256) void qkprep_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ q_w,
                                                         const bf16_t* __restrict__ k_w, const bf16_t* __restrict__ cs,
                                                         const bf16_t* __restrict__ sn, bf16_t* __restrict__ Q,
                                                         bf16_t* __restrict__ K, bf16_t* __restrict__ V,
                                                         float* __restrict__ q_rstd, float* __restrict__ k_rstd, int L,
                                                         int Hq, int Hkv, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long t = blockIdx.x;  // b*L + l
  const int b = (int)(t / L), l = (int)(t - (long)b * L);
  const int HT = Hq + 2 * Hkv;
  const bf16_t* row = qkv + t * (long)HT * D;
  const float c1 = bf2f(cs[l * D + lane]), c2 = bf2f(cs[l * D + lane + 64]);
  const float s1 = bf2f(sn[l * D + lane]), s2 = bf2f(sn[l * D + lane + 64]);
  for (int h = wid; h < HT; h += 4) {
    const float x1 = bf2f(row[h * D + lane]), x2 = bf2f(row[h * D + lane + 64]);
    if (h >= Hq + Hkv) {  // value head: plain copy
      const int hv = h - Hq - Hkv;
      bf16_t* o = V + (((long)b * Hkv + hv) * L + l) * D;
      o[lane] = f2bf(x1);
      o[lane + 64] = f2bf(x2);
      continue;
    }
    const bool isq = h < Hq;
    const bf16_t* w = isq ? q_w : k_w;
    const float rs = rsqrtf(wave_sum(x1 * x1 + x2 * x2) / (float)D + eps);
    const float n1 = rbf(bf2f(w[lane]) * rbf(x1 * rs));
    const float n2 = rbf(bf2f(w[lane + 64]) * rbf(x2 * rs));
    // q*cos + rotate_half(q)*sin, each product and the sum rounded to bf16 as the PyTorch bf16 ops do
    const float o1 = rbf(rbf(n1 * c1) + rbf(-n2 * s1));
    const float o2 = rbf(rbf(n2 * c2) + rbf(n1 * s2));
    bf16_t* o;
    if (isq) {
      o = Q + (((long)b * Hq + h) * L + l) * D;
      if (q_rstd && lane == 0) q_rstd[t * Hq + h] = rs;
    } else {
      const int hk = h - Hq;
      o = K + (((long)b * Hkv + hk) * L + l) * D;
      if (k_rstd && lane == 0) k_rstd[t * Hkv + hk] = rs;
    }
    o[lane] = f2bf(o1);
    o[lane + 64] = f2bf(o2);
  }
}

__global__ __launch_bounds__(256) void qkprep_bwd_kernel(const bf16_t* __restrict__ dQ, const bf16_t* __restrict__ dK,
                                                         const bf16_t* __restrict__ dV, const bf16_t* __restrict__ qkv,
                                                         const bf16_t* __restrict__ q_w, const bf16_t* __restrict__ k_w,
                                                         const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                                         const float* __restrict__ q_rstd,
                                                         const float* __restrict__ k_rstd, bf16_t* __restrict__ dqkv,
                                                         float* __restrict__ dq_w, float* __restrict__ dk_w, int L,
                                                         int Hq, int Hkv) {
  __shared__ float dw_s[2][D];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x < 2 * D) (&dw_s[0][0])[threadIdx.x] = 0.f;
  __syncthreads();
  const long t = blockIdx.x;
  const int b = (int)(t / L), l = (int)(t - (long)b * L);
  const int HT = Hq + 2 * Hkv;
  const bf16_t* row = qkv + t * (long)HT * D;
  bf16_t* drow = dqkv + t * (long)HT * D;
  const float c1 = bf2f(cs[l * D + lane]), c2 = bf2f(cs[l * D + lane + 64]);
  const float s1 = bf2f(sn[l * D + lane]), s2 = bf2f(sn[l * D + lane + 64]);
  float aq1 = 0.f, aq2 = 0.f, ak1 = 0.f, ak2 = 0.f;
  for (int h = wid; h < HT; h += 4) {
    if (h >= Hq + Hkv) {
      const int hv = h - Hq - Hkv;
      const bf16_t* g = dV + (((long)b * Hkv + hv) * L + l) * D;
      drow[h * D + lane] = g[lane];
      drow[h * D + lane + 64] = g[lane + 64];
      continue;
    }
    const bool isq = h < Hq;
    const bf16_t* g = isq ? dQ + (((long)b * Hq + h) * L + l) * D : dK + (((long)b * Hkv + (h - Hq)) * L + l) * D;
    const bf16_t* w = isq ? q_w : k_w;
    const float rs = isq ? q_rstd[t * Hq + h] : k_rstd[t * Hkv + (h - Hq)];
    const float dy1 = bf2f(g[lane]), dy2 = bf2f(g[lane + 64]);
    // y1 = n1*c1 - n2*s1 ; y2 = n2*c2 + n1*s2
    const float dn1 = dy1 * c1 + dy2 * s2;
    const float dn2 = -dy1 * s1 + dy2 * c2;
    const float xh1 = bf2f(row[h * D + lane]) * rs, xh2 = bf2f(row[h * D + lane + 64]) * rs;
    const float g1 = dn1 * bf2f(w[lane]), g2 = dn2 * bf2f(w[lane + 64]);
    const float dot = wave_sum(g1 * xh1 + g2 * xh2) / (float)D;
    drow[h * D + lane] = f2bf(rs * (g1 - xh1 * dot));
    drow[h * D + lane + 64] = f2bf(rs * (g2 - xh2 * dot));
    if (isq) { aq1 += dn1 * xh1; aq2 += dn2 * xh2; } else { ak1 += dn1 * xh1; ak2 += dn2 * xh2; }
  }
  atomicAdd(&dw_s[0][lane], aq1);
  atomicAdd(&dw_s[0][lane + 64], aq2);
  atomicAdd(&dw_s[1][lane], ak1);
  atomicAdd(&dw_s[1][lane + 64], ak2);
  __syncthreads();
  // plain-store partial rows [token][D] (summed by vq3_colsum_f32_to_bf16): no contended global atomics
  if (threadIdx.x < D) {
    dq_w[t * D + threadIdx.x] = dw_s[0][threadIdx.x];
  } else {
    dk_w[t * D + threadIdx.x - D] = dw_s[1][threadIdx.x - D];
  }
}

}  // namespace

extern "C" int vq3_qwen_qkprep_fwd(const void* qkv, const void* q_w, const void* k_w, const void* cos, const void* sin,
                                   void* Q, void* K, void* V, float* q_rstd, float* k_rstd, int32_t B, int32_t L,
                                   int32_t Hq, int32_t Hkv, int32_t Dh, float eps, void* stream) {
  VQ3_CHECK_ARG(qkv && q_w && k_w && cos && sin && Q && K && V, "qkprep_fwd: null pointer");
  VQ3_CHECK_ARG(Dh == D, "qkprep_fwd: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && L > 0 && Hq > 0 && Hkv > 0, "qkprep_fwd: bad shape");
  hipLaunchKernelGGL(qkprep_fwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                     (const bf16_t*)q_w, (const bf16_t*)k_w, (const bf16_t*)cos, (const bf16_t*)sin, (bf16_t*)Q,
                     (bf16_t*)K, (bf16_t*)V, q_rstd, k_rstd, L, Hq, Hkv, eps);
  VQ3_CHECK_LAUNCH("qkprep_fwd");
  return 0;
}

extern "C" int vq3_qwen_qkprep_bwd(const void* dQ, const void* dK, const void* dV, const void* qkv, const void* q_w,
                                   const void* k_w, const void* cos, const void* sin, const float* q_rstd,
                                   const float* k_rstd, void* dqkv, float* dq_w_f32, float* dk_w_f32, int32_t B,
                                   int32_t L, int32_t Hq, int32_t Hkv, int32_t Dh, void* stream) {
  VQ3_CHECK_ARG(dQ && dK && dV && qkv && q_w && k_w && cos && sin && q_rstd && k_rstd && dqkv && dq_w_f32 && dk_w_f32,
                "qkprep_bwd: null pointer");
  VQ3_CHECK_ARG(Dh == D, "qkprep_bwd: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && L > 0 && Hq > 0 && Hkv > 0, "qkprep_bwd: bad shape");
  hipLaunchKernelGGL(qkprep_bwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dQ,
                     (const bf16_t*)dK, (const bf16_t*)dV, (const bf16_t*)qkv, (const bf16_t*)q_w, (const bf16_t*)k_w,
                     (const bf16_t*)cos, (const bf16_t*)sin, q_rstd, k_rstd, (bf16_t*)dqkv, dq_w_f32, dk_w_f32, L, Hq,
                     Hkv);
  VQ3_CHECK_LAUNCH("qkprep_bwd");
  return 0;
}
