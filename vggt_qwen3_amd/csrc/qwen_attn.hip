// Qwen3 attention pre-processing: split the fused q|k|v projection, per-head RMSNorm on q and k
// (modeling_qwen3.py:237-238,252-253), rotate-half RoPE (modeling_qwen3.py:104-148), and the layout change
// [B*L, heads*D] -> [B, heads, L, D] that the batched attention GEMMs consume. head_dim is 128: one wave owns one
// head vector, lane i holds elements i and i+64 - exactly the rotate_half pair, so RoPE needs no cross-lane
// traffic and the RMS reduction is one butterfly.
#include "common.h"
#include "vq3_hip.h"

namespace {

constexpr int D = 128;
constexpr int QK_IT = 6;     // heads per half-wave whose loads are in flight together (48 heads / 8 half-waves)

// Lane layout: a HALF wave (32 lanes) owns one head; lane j holds elements 2j, 2j+1 and their rotate-half partners
// 2j+64, 2j+65 (two 4-byte accesses), so a wave covers two heads per pass, RoPE needs no cross-lane traffic and the RMS
// reduction is a 32-lane butterfly.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void ld2(const bf16_t* p, float& a, float& b) {
  const uint32_t r = *reinterpret_cast<const uint32_t*>(p);
  a = bf2f((bf16_t)(r & 0xffff));
  b = bf2f((bf16_t)(r >> 16));
}

__global__ __launch_bounds__(256) void qkprep_fwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ q_w,
                                                         const bf16_t* __restrict__ k_w, const bf16_t* __restrict__ cs,
                                                         const bf16_t* __restrict__ sn, bf16_t* __restrict__ Q,
                                                         bf16_t* __restrict__ K, bf16_t* __restrict__ V,
                                                         float* __restrict__ q_rstd, float* __restrict__ k_rstd, int L,
                                                         int Hq, int Hkv, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int e = 2 * j;                       // elements e, e+1 and e+64, e+65
  const long t = blockIdx.x;  // b*L + l
  const int b = (int)(t / L), l = (int)(t - (long)b * L);
  const int HT = Hq + 2 * Hkv;
  const bf16_t* row = qkv + t * (long)HT * D;
  float c1a, c1b, c2a, c2b, s1a, s1b, s2a, s2b;
  ld2(cs + l * D + e, c1a, c1b); ld2(cs + l * D + e + 64, c2a, c2b);
  ld2(sn + l * D + e, s1a, s1b); ld2(sn + l * D + e + 64, s2a, s2b);
  // the rows of QK_IT heads are requested before the first is used (Qwen3-4B: 48 heads = 6 per half-wave = one batch): the loop used
  // to pay one memory round trip per head
  float wq1a, wq1b, wq2a, wq2b, wk1a, wk1b, wk2a, wk2b;
  ld2(q_w + e, wq1a, wq1b); ld2(q_w + e + 64, wq2a, wq2b);
  ld2(k_w + e, wk1a, wk1b); ld2(k_w + e + 64, wk2a, wk2b);
  for (int hb = 2 * wid + half; hb < HT; hb += 8 * QK_IT) {
  uint32_t xr1[QK_IT], xr2[QK_IT];
#pragma unroll
  for (int i = 0; i < QK_IT; ++i) {
    const int h = hb + 8 * i;
    const int hc = h < HT ? h : HT - 1;
    xr1[i] = *reinterpret_cast<const uint32_t*>(row + hc * D + e);
    xr2[i] = *reinterpret_cast<const uint32_t*>(row + hc * D + e + 64);
  }
#pragma unroll
  for (int i = 0; i < QK_IT; ++i) {
    const int h = hb + 8 * i;
    if (h >= HT) break;
    const float x1a = bf2f((bf16_t)(xr1[i] & 0xffff)), x1b = bf2f((bf16_t)(xr1[i] >> 16));
    const float x2a = bf2f((bf16_t)(xr2[i] & 0xffff)), x2b = bf2f((bf16_t)(xr2[i] >> 16));
    if (h >= Hq + Hkv) {  // value head: plain copy
      const int hv = h - Hq - Hkv;
      bf16_t* o = V + (((long)b * Hkv + hv) * L + l) * D;
      *reinterpret_cast<uint32_t*>(o + e) = pack2bf(x1a, x1b);
      *reinterpret_cast<uint32_t*>(o + e + 64) = pack2bf(x2a, x2b);
      continue;
    }
    const bool isq = h < Hq;
    const float w1a = isq ? wq1a : wk1a, w1b = isq ? wq1b : wk1b, w2a = isq ? wq2a : wk2a, w2b = isq ? wq2b : wk2b;
    const float rs = rsqrtf(half_sum(x1a * x1a + x1b * x1b + x2a * x2a + x2b * x2b) / (float)D + eps);
    const float n1a = rbf(w1a * rbf(x1a * rs)), n1b = rbf(w1b * rbf(x1b * rs));
    const float n2a = rbf(w2a * rbf(x2a * rs)), n2b = rbf(w2b * rbf(x2b * rs));
    // q*cos + rotate_half(q)*sin, each product and the sum rounded to bf16 as the PyTorch bf16 ops do
    const float o1a = rbf(rbf(n1a * c1a) + rbf(-n2a * s1a)), o1b = rbf(rbf(n1b * c1b) + rbf(-n2b * s1b));
    const float o2a = rbf(rbf(n2a * c2a) + rbf(n1a * s2a)), o2b = rbf(rbf(n2b * c2b) + rbf(n1b * s2b));
    bf16_t* o;
    if (isq) {
      o = Q + (((long)b * Hq + h) * L + l) * D;
      if (q_rstd && j == 0) q_rstd[t * Hq + h] = rs;
    } else {
      const int hk = h - Hq;
      o = K + (((long)b * Hkv + hk) * L + l) * D;
      if (k_rstd && j == 0) k_rstd[t * Hkv + hk] = rs;
    }
    *reinterpret_cast<uint32_t*>(o + e) = pack2bf(o1a, o1b);
    *reinterpret_cast<uint32_t*>(o + e + 64) = pack2bf(o2a, o2b);
  }
  }
}

__global__ __launch_bounds__(256) void qkprep_bwd_kernel(const bf16_t* __restrict__ dQ, const bf16_t* __restrict__ dK,
                                                         const bf16_t* __restrict__ dV, const bf16_t* __restrict__ qkv,
                                                         const bf16_t* __restrict__ q_w, const bf16_t* __restrict__ k_w,
                                                         const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                                         const float* __restrict__ q_rstd,
                                                         const float* __restrict__ k_rstd, bf16_t* __restrict__ dqkv,
                                                         float* __restrict__ dq_w, float* __restrict__ dk_w, int L,
                                                         int Hq, int Hkv, int kv_parts, long part_stride, long T) {
  // VQ3_QKPREP_BWD_TOKENS_PER_PART tokens per workgroup: the norm-weight partials of a workgroup's tokens are summed in registers and
  // meet in LDS once (one zeroing, one round of LDS atomics, two barriers and one partial row per 8 tokens instead of per token: kernel 150 -> 114-118 us cold at 9600 tokens, the two column sums 2 x 50 -> 2 x 5-14 us)
  // (round 5: the eight half-waves' partials meet in LDS in a FIXED order - slot per half-wave, summed 0..7 - instead of through LDS float
  // atomics: the q_norm / k_norm weight gradients were the last run-to-run difference of a training step's gradients)
  __shared__ float dw_s[8][2][D];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int e = 2 * j;
  const int HT = Hq + 2 * Hkv;
  float aq1[2] = {0.f, 0.f}, aq2[2] = {0.f, 0.f}, ak1[2] = {0.f, 0.f}, ak2[2] = {0.f, 0.f};
  float wq1[2], wq2[2], wk1[2], wk2[2];
  ld2(q_w + e, wq1[0], wq1[1]); ld2(q_w + e + 64, wq2[0], wq2[1]);
  ld2(k_w + e, wk1[0], wk1[1]); ld2(k_w + e + 64, wk2[0], wk2[1]);
  const int np = kv_parts < 2 ? 1 : 2;             // slabs fetched with the batch (further ones, kv_parts 3..4: in the arithmetic loop)
  for (int tt = 0; tt < VQ3_QKPREP_BWD_TOKENS_PER_PART; ++tt) {
  const long t = (long)blockIdx.x * VQ3_QKPREP_BWD_TOKENS_PER_PART + tt;
  if (t >= T) break;
  const int b = (int)(t / L), l = (int)(t - (long)b * L);
  const bf16_t* row = qkv + t * (long)HT * D;
  bf16_t* drow = dqkv + t * (long)HT * D;
  float c1[2], c2[2], s1[2], s2[2];
  ld2(cs + l * D + e, c1[0], c1[1]); ld2(cs + l * D + e + 64, c2[0], c2[1]);
  ld2(sn + l * D + e, s1[0], s1[1]); ld2(sn + l * D + e + 64, s2[0], s2[1]);
  // the gradient rows (first two partial slabs), the saved q|k|v rows and the row statistics of QK_IT heads are requested before the
  // first is used: one memory round trip per batch instead of one per head (Qwen3-4B: 48 heads = 6 per half-wave = one batch)
  for (int hb = 2 * wid + half; hb < HT; hb += 8 * QK_IT) {
    uint32_t ga1[QK_IT], ga2[QK_IT], gb1[QK_IT], gb2[QK_IT], xr1[QK_IT], xr2[QK_IT];
    float rsv[QK_IT];
#pragma unroll
    for (int i = 0; i < QK_IT; ++i) {
      const int h = hb + 8 * i;
      const int hc = h < HT ? h : HT - 1;
      const bool isq = hc < Hq, isv = hc >= Hq + Hkv;
      const bf16_t* g = isq ? dQ + (((long)b * Hq + hc) * L + l) * D
                            : (isv ? dV + (((long)b * Hkv + (hc - Hq - Hkv)) * L + l) * D : dK + (((long)b * Hkv + (hc - Hq)) * L + l) * D);
      ga1[i] = *reinterpret_cast<const uint32_t*>(g + e);
      ga2[i] = *reinterpret_cast<const uint32_t*>(g + e + 64);
      const bf16_t* g2 = (!isq && np == 2) ? g + part_stride : g;      // (q heads: re-read of the same line, never used)
      gb1[i] = *reinterpret_cast<const uint32_t*>(g2 + e);
      gb2[i] = *reinterpret_cast<const uint32_t*>(g2 + e + 64);
      xr1[i] = *reinterpret_cast<const uint32_t*>(row + hc * D + e);
      xr2[i] = *reinterpret_cast<const uint32_t*>(row + hc * D + e + 64);
      rsv[i] = isv ? 1.f : (isq ? q_rstd[t * Hq + hc] : k_rstd[t * Hkv + (hc - Hq)]);
    }
#pragma unroll
    for (int i = 0; i < QK_IT; ++i) {
      const int h = hb + 8 * i;
      if (h >= HT) break;
      const bool isq = h < Hq;
      float dy1[2] = {bf2f((bf16_t)(ga1[i] & 0xffff)), bf2f((bf16_t)(ga1[i] >> 16))};
      float dy2[2] = {bf2f((bf16_t)(ga2[i] & 0xffff)), bf2f((bf16_t)(ga2[i] >> 16))};
      if (!isq && np == 2) {                         // partial slabs of the split dK/dV pass: summed in f32 in slab order, rounded once
        dy1[0] += bf2f((bf16_t)(gb1[i] & 0xffff)); dy1[1] += bf2f((bf16_t)(gb1[i] >> 16));
        dy2[0] += bf2f((bf16_t)(gb2[i] & 0xffff)); dy2[1] += bf2f((bf16_t)(gb2[i] >> 16));
        const bf16_t* g = h >= Hq + Hkv ? dV + (((long)b * Hkv + (h - Hq - Hkv)) * L + l) * D : dK + (((long)b * Hkv + (h - Hq)) * L + l) * D;
        for (int sp = 2; sp < kv_parts; ++sp) {
          float t0, t1, u0, u1;
          ld2(g + sp * part_stride + e, t0, t1); ld2(g + sp * part_stride + e + 64, u0, u1);
          dy1[0] += t0; dy1[1] += t1; dy2[0] += u0; dy2[1] += u1;
        }
      }
      if (h >= Hq + Hkv) {
        if (kv_parts == 1) {
          *reinterpret_cast<uint32_t*>(drow + h * D + e) = ga1[i];
          *reinterpret_cast<uint32_t*>(drow + h * D + e + 64) = ga2[i];
        } else {
          *reinterpret_cast<uint32_t*>(drow + h * D + e) = pack2bf(dy1[0], dy1[1]);
          *reinterpret_cast<uint32_t*>(drow + h * D + e + 64) = pack2bf(dy2[0], dy2[1]);
        }
        continue;
      }
      const float rs = rsv[i];
      const float x1[2] = {bf2f((bf16_t)(xr1[i] & 0xffff)), bf2f((bf16_t)(xr1[i] >> 16))};
      const float x2[2] = {bf2f((bf16_t)(xr2[i] & 0xffff)), bf2f((bf16_t)(xr2[i] >> 16))};
      float dn1[2], dn2[2], xh1[2], xh2[2], g1[2], g2[2], part = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // y1 = n1*c1 - n2*s1 ; y2 = n2*c2 + n1*s2
        dn1[u] = dy1[u] * c1[u] + dy2[u] * s2[u];
        dn2[u] = -dy1[u] * s1[u] + dy2[u] * c2[u];
        xh1[u] = x1[u] * rs; xh2[u] = x2[u] * rs;
        g1[u] = dn1[u] * (isq ? wq1[u] : wk1[u]); g2[u] = dn2[u] * (isq ? wq2[u] : wk2[u]);
        part += g1[u] * xh1[u] + g2[u] * xh2[u];
      }
      const float dot = half_sum(part) / (float)D;
      *reinterpret_cast<uint32_t*>(drow + h * D + e) = pack2bf(rs * (g1[0] - xh1[0] * dot), rs * (g1[1] - xh1[1] * dot));
      *reinterpret_cast<uint32_t*>(drow + h * D + e + 64) = pack2bf(rs * (g2[0] - xh2[0] * dot), rs * (g2[1] - xh2[1] * dot));
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (isq) { aq1[u] += dn1[u] * xh1[u]; aq2[u] += dn2[u] * xh2[u]; } else { ak1[u] += dn1[u] * xh1[u]; ak2[u] += dn2[u] * xh2[u]; }
      }
    }
  }
  }   // tokens of this workgroup
  const int hw = 2 * wid + half;                   // this half-wave's slot: it covers all D features (lane j: e, e+1, e+64, e+65)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    dw_s[hw][0][e + u] = aq1[u];
    dw_s[hw][0][e + u + 64] = aq2[u];
    dw_s[hw][1][e + u] = ak1[u];
    dw_s[hw][1][e + u + 64] = ak2[u];
  }
  __syncthreads();
  // plain-store partial rows [workgroup][D] (summed by vq3_colsum_f32_to_bf16): no contended global atomics
  {
    const int which = threadIdx.x < D ? 0 : 1, x = threadIdx.x < D ? threadIdx.x : threadIdx.x - D;
    float v = dw_s[0][which][x];
#pragma unroll
    for (int i = 1; i < 8; ++i) v += dw_s[i][which][x];
    (which ? dk_w : dq_w)[(long)blockIdx.x * D + x] = v;
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
                                   const float* k_rstd, void* dqkv, float* dq_w_f32, float* dk_w_f32, int32_t kv_parts,
                                   int32_t B, int32_t L, int32_t Hq, int32_t Hkv, int32_t Dh, void* stream) {
  VQ3_CHECK_ARG(dQ && dK && dV && qkv && q_w && k_w && cos && sin && q_rstd && k_rstd && dqkv && dq_w_f32 && dk_w_f32,
                "qkprep_bwd: null pointer");
  VQ3_CHECK_ARG(Dh == D, "qkprep_bwd: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && L > 0 && Hq > 0 && Hkv > 0 && kv_parts >= 1 && kv_parts <= 4, "qkprep_bwd: bad shape");
  const long T = (long)B * L;
  const unsigned nblk = (unsigned)((T + VQ3_QKPREP_BWD_TOKENS_PER_PART - 1) / VQ3_QKPREP_BWD_TOKENS_PER_PART);
  hipLaunchKernelGGL(qkprep_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dQ,
                     (const bf16_t*)dK, (const bf16_t*)dV, (const bf16_t*)qkv, (const bf16_t*)q_w, (const bf16_t*)k_w,
                     (const bf16_t*)cos, (const bf16_t*)sin, q_rstd, k_rstd, (bf16_t*)dqkv, dq_w_f32, dk_w_f32, L, Hq,
                     Hkv, (int)kv_parts, (long)B * Hkv * L * D, T);
  VQ3_CHECK_LAUNCH("qkprep_bwd");
  return 0;
}
