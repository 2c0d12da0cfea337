// Greedy decoding for Qwen3 with a resident KV cache (the reference's inference callers:
// src/inference/qa_inference.py:207-216, arkit_inference.py:274-284 -> transformers generate(), greedy, with
// repetition_penalty and no_repeat_ngram_size). One new token per row and step, so every contraction is a
// matrix-vector product that streams the weights once: HBM-bound, not MFMA-bound.
//
// All step state lives in device memory (per-row cache lengths, step counter, generated ids, finished flags), so one
// decode step is a fixed sequence of launches with fixed arguments - capturable in a HIP graph and replayed per token.
//
//  * vq3_skinny_gemm_bf16   y[M<=8, N] = x W^T (+ residual): a wave owns 4 weight rows, lanes stride K in 16-byte
//                           pieces (coalesced 1 KiB per row per load), fp32 accumulation, wave reduction at the end.
//  * vq3_qwen_decode_qkprep q/k RMSNorm + RoPE at position lens[b], K and V written straight into the cache slot.
//  * vq3_qwen_decode_attn   softmax(q K^T / sqrt(D)) V over the row's lens[b]+1 cached positions, GQA by indexing.
//  * vq3_greedy_pick        repetition penalty -> n-gram ban -> argmax -> eos/pad bookkeeping, one workgroup per row.
//  * vq3_decode_advance     lens[b]++, step++ (single thread; runs last in the step).
#include "common.h"
#include "vq3_hip.h"

namespace {

constexpr int D = 128;

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
  f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, v.z << 16); f[5] = __builtin_bit_cast(float, v.z & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, v.w << 16); f[7] = __builtin_bit_cast(float, v.w & 0xffff0000u);
}

// ------------------------------------------------------------------------------------------------ skinny GEMM
// XMODE 0: x as given. 1: x = ln_w * bf16(x * rstd(x)) (Qwen3RMSNorm fused into the consumer; every wave recomputes the
// row statistics from the K-element rows, which sit in L2). 2: x = bf16(silu(gate) * up) with gate = xin[:, :K],
// up = xin[:, K:2K] (SwiGLU fused into down_proj).
template <int MT, int R, int XMODE>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                         void* __restrict__ y, const bf16_t* __restrict__ res,
                                                         const bf16_t* __restrict__ ln_w, float eps, int M, int N, int K,
                                                         long ldx, long ldw, long ldy, long ldr, int out_f32) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wid) * R;
  if (n0 >= N) return;
  float rstd[MT];
  if (XMODE == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float ss = 0.f;
      const bf16_t* xr = x + (long)min(m, M - 1) * ldx;
      for (int k = lane * 8; k < K; k += 512) {
        float f[8];
        unpack8(*(const u32x4*)(xr + k), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(f[j], f[j], ss);
      }
      rstd[m] = rsqrtf(wave_sum(ss) / (float)K + eps);
    }
  }
  float acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  const bf16_t* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = W + (long)min(n0 + r, N - 1) * ldw;
  constexpr int U = R >= 4 ? 1 : (R == 2 ? 2 : 4);   // K chunks in flight: R * U 16-byte weight loads per lane
  for (int k0 = lane * 8; k0 < K; k0 += 512 * U) {   // K % 8 == 0
    u32x4 wv[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + 512 * u, K - 8);        // clamped duplicate loads are masked below
#pragma unroll
      for (int r = 0; r < R; ++r) wv[u][r] = __builtin_nontemporal_load((const u32x4*)(wrow[r] + k));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 512 * u;
      if (k >= K) break;
      float xf[MT][8];
      float lw[8];
      if (XMODE == 1) unpack8(*(const u32x4*)(ln_w + k), lw);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bf16_t* xr = x + (long)min(m, M - 1) * ldx;
        unpack8(*(const u32x4*)(xr + k), xf[m]);
        if (XMODE == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[m][j] = rbf(lw[j] * rbf(xf[m][j] * rstd[m]));
        } else if (XMODE == 2) {
          float up[8];
          unpack8(*(const u32x4*)(xr + K + k), up);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float g = xf[m][j];
            xf[m][j] = rbf(rbf(silu_f(g)) * up[j]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float wf[8];
        unpack8(wv[u][r], wf);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r][m] = fmaf(wf[j], xf[m][j], acc[r][m]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float s = wave_sum(acc[r][m]);
      if (lane == 0 && m < M && n0 + r < N) {
        float v = s;
        if (res) v = rbf(v) + bf2f(res[(long)m * ldr + n0 + r]);   // GEMM output is bf16 before the residual add
        if (out_f32) ((float*)y)[(long)m * ldy + n0 + r] = v;
        else ((bf16_t*)y)[(long)m * ldy + n0 + r] = f2bf(v);
      }
    }
}

// Register-resident-x variant for 1-2 rows (the reference's callers decode one prompt at a time): a block of 4 waves is
// split into 4/KS row groups x KS K-slices (512-element chunks interleaved across the slices). Each wave loads its slice
// of x ONCE (<= CH chunks, transformed in registers: RMSNorm statistics are combined across the slices through LDS;
// SwiGLU is evaluated once per wave), then streams R weight rows with all R*CH 16-byte loads of a row group in
// flight before the first FMA. More waves per weight byte than the generic kernel -> enough loads in flight to
// approach HBM bandwidth at N = 2560.
template <int MT, int R, int KS, int XMODE>
__global__ __launch_bounds__(256) void skinny_regx_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                         void* __restrict__ y, const bf16_t* __restrict__ res,
                                                         const bf16_t* __restrict__ ln_w, float eps, int M, int N, int K,
                                                         long ldx, long ldw, long ldy, long ldr, int out_f32) {
  constexpr int CH = 5;
  constexpr int G = 4 / KS;
  __shared__ float red[4][MT];
  __shared__ float part[4][R][MT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = wid / KS, ks = wid % KS;
  const int n0 = (blockIdx.x * G + g) * R;
  float xf[MT][CH][8];
  int kof[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) kof[c] = (c * KS + ks) * 512 + lane * 8;
  // the weight stream does not depend on x: put all of this wave's weight loads in flight first, then prepare x under them
  const bool active = n0 < N;
  u32x4 wv[R][CH];
  if (active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bf16_t* wr = W + (long)min(n0 + r, N - 1) * ldw;
#pragma unroll
      for (int c = 0; c < CH; ++c) wv[r][c] = __builtin_nontemporal_load((const u32x4*)(wr + min(kof[c], K - 8)));
    }
  }
  u32x4 lwv[CH];
  if (XMODE == 1) {
#pragma unroll
    for (int c = 0; c < CH; ++c) lwv[c] = *(const u32x4*)(ln_w + min(kof[c], K - 8));
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bf16_t* xr = x + (long)min(m, M - 1) * ldx;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (kof[c] < K) {
        unpack8(*(const u32x4*)(xr + kof[c]), xf[m][c]);
        if (XMODE == 2) {
          float up[8];
          unpack8(*(const u32x4*)(xr + K + kof[c]), up);
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[m][c][j] = rbf(rbf(silu_f(xf[m][c][j])) * up[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[m][c][j] = 0.f;
      }
    }
  }
  if (XMODE == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(xf[m][c][j], xf[m][c][j], ss);
      ss = wave_sum(ss);
      if (lane == 0) red[wid][m] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < KS; ++q) tot += red[g * KS + q][m];
      const float rstd = rsqrtf(tot / (float)K + eps);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (kof[c] < K) {
          float lw[8];
          unpack8(lwv[c], lw);
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[m][c][j] = rbf(lw[j] * rbf(xf[m][c][j] * rstd));
        }
      }
    }
  }
  float acc[R][MT];
  if (active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float wf[8];
        unpack8(wv[r][c], wf);          // chunks past K meet x == 0
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r][m] = fmaf(wf[j], xf[m][c][j], acc[r][m]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[r][m] = wave_sum(acc[r][m]);
        if (KS > 1 && lane == 0) part[wid][r][m] = acc[r][m];
      }
  }
  if (KS > 1) __syncthreads();
  if (n0 >= N || ks != 0 || lane != 0) return;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m >= M || n0 + r >= N) continue;
      float v = acc[r][m];
#pragma unroll
      for (int q = 1; q < KS; ++q) v += part[wid + q][r][m];
      if (res) v = rbf(v) + bf2f(res[(long)m * ldr + n0 + r]);
      if (out_f32) ((float*)y)[(long)m * ldy + n0 + r] = v;
      else ((bf16_t*)y)[(long)m * ldy + n0 + r] = f2bf(v);
    }
}

// e4m3 weights (config C5): the same register-resident-x scheme with 16-byte pieces = 16 weights. The activation row is
// quantised per token exactly as the fp8 GEMM does (amax over the whole row -> e4m3 round trip in registers), so decode
// and prefill follow one numeric contract; the weight stream - the only thing that costs time here - is halved.
__device__ __forceinline__ void unpack16_fp8(const u32x4 v, float* f) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[w], false);
    const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[w], true);
    f[4 * w + 0] = lo[0]; f[4 * w + 1] = lo[1]; f[4 * w + 2] = hi[0]; f[4 * w + 3] = hi[1];
  }
}

template <int MT, int R, int KS, int XMODE>
__global__ __launch_bounds__(256) void skinny_fp8_kernel(const bf16_t* __restrict__ x, const uint8_t* __restrict__ W,
                                                        const float* __restrict__ wscale, bf16_t* __restrict__ y,
                                                        const bf16_t* __restrict__ res, const bf16_t* __restrict__ ln_w,
                                                        float eps, int M, int N, int K, long ldx, long ldw, long ldy, long ldr) {
  constexpr int CH = 3;
  constexpr int G = 4 / KS;
  __shared__ float red[4][MT];
  __shared__ float redm[4][MT];
  __shared__ float part[4][R][MT];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = wid / KS, ks = wid % KS;
  const int n0 = (blockIdx.x * G + g) * R;
  const bool active = n0 < N;
  int kof[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) kof[c] = (c * KS + ks) * 1024 + lane * 16;
  u32x4 wv[R][CH];
  if (active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint8_t* wr = W + (long)min(n0 + r, N - 1) * ldw;
#pragma unroll
      for (int c = 0; c < CH; ++c) wv[r][c] = __builtin_nontemporal_load((const u32x4*)(wr + min(kof[c], K - 16)));
    }
  }
  float xf[MT][CH][16];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const bf16_t* xr = x + (long)min(m, M - 1) * ldx;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (kof[c] < K) {
        unpack8(*(const u32x4*)(xr + kof[c]), xf[m][c]);
        unpack8(*(const u32x4*)(xr + kof[c] + 8), xf[m][c] + 8);
        if (XMODE == 2) {
          float up[16];
          unpack8(*(const u32x4*)(xr + K + kof[c]), up);
          unpack8(*(const u32x4*)(xr + K + kof[c] + 8), up + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) xf[m][c][j] = rbf(rbf(silu_f(xf[m][c][j])) * up[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) xf[m][c][j] = 0.f;
      }
    }
  }
  if (XMODE == 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) ss = fmaf(xf[m][c][j], xf[m][c][j], ss);
      ss = wave_sum(ss);
      if (lane == 0) red[wid][m] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < KS; ++q) tot += red[g * KS + q][m];
      const float rstd = rsqrtf(tot / (float)K + eps);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (kof[c] < K) {
          float lw[16];
          unpack8(*(const u32x4*)(ln_w + kof[c]), lw);
          unpack8(*(const u32x4*)(ln_w + kof[c] + 8), lw + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) xf[m][c][j] = rbf(lw[j] * rbf(xf[m][c][j] * rstd));
        }
      }
    }
  }
  // per-token e4m3 quantisation of the prepared row (vq3_quant_fp8_rows contract): q = e4m3(x * 448 / amax)
  float sx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) am = fmaxf(am, fabsf(xf[m][c][j]));
    am = wave_max(am);
    if (lane == 0) redm[wid][m] = am;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float am = 0.f;
#pragma unroll
    for (int q = 0; q < KS; ++q) am = fmaxf(am, redm[g * KS + q][m]);
    const float inv = am > 0.f ? 448.0f / am : 1.0f;
    sx[m] = am > 0.f ? am / 448.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(xf[m][c][j] * inv, xf[m][c][j + 1] * inv, 0, false);
        const auto back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
        xf[m][c][j] = back[0];
        xf[m][c][j + 1] = back[1];
      }
  }
  float acc[R][MT];
  if (active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float wf[16];
        unpack16_fp8(wv[r][c], wf);          // chunks past K meet x == 0
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[r][m] = fmaf(wf[j], xf[m][c][j], acc[r][m]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[r][m] = wave_sum(acc[r][m]);
        if (KS > 1 && lane == 0) part[wid][r][m] = acc[r][m];
      }
  }
  if (KS > 1) __syncthreads();
  if (!active || ks != 0 || lane != 0) return;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m >= M || n0 + r >= N) continue;
      float v = acc[r][m];
#pragma unroll
      for (int q = 1; q < KS; ++q) v += part[wid + q][r][m];
      v = rbf(v * (sx[m] * wscale[n0 + r]));
      if (res) v += bf2f(res[(long)m * ldr + n0 + r]);
      y[(long)m * ldy + n0 + r] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------ decode q/k prep
// grid (B, ceil(heads/4)): one head per wave; lane i holds elements i and i+64 (the rotate_half pair)
__global__ __launch_bounds__(256) void decode_qkprep_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ q_w,
                                                           const bf16_t* __restrict__ k_w, const bf16_t* __restrict__ cs,
                                                           const bf16_t* __restrict__ sn, const int32_t* __restrict__ lens,
                                                           bf16_t* __restrict__ Q, bf16_t* __restrict__ Kc,
                                                           bf16_t* __restrict__ Vc, int Hq, int Hkv, int Lmax, float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int pos = lens[b];
  if (pos >= Lmax) return;   // cache full: the host checks capacity before launching; never write out of bounds
  const int HT = Hq + 2 * Hkv;
  const bf16_t* row = qkv + (long)b * HT * D;
  const float c1 = bf2f(cs[(long)pos * D + lane]), c2 = bf2f(cs[(long)pos * D + lane + 64]);
  const float s1 = bf2f(sn[(long)pos * D + lane]), s2 = bf2f(sn[(long)pos * D + lane + 64]);
  {
    const int h = blockIdx.y * 4 + wid;
    if (h >= HT) return;
    const float x1 = bf2f(row[h * D + lane]), x2 = bf2f(row[h * D + lane + 64]);
    if (h >= Hq + Hkv) {
      bf16_t* o = Vc + (((long)b * Hkv + (h - Hq - Hkv)) * Lmax + pos) * D;
      o[lane] = f2bf(x1);
      o[lane + 64] = f2bf(x2);
      return;
    }
    const bool isq = h < Hq;
    const bf16_t* w = isq ? q_w : k_w;
    const float rs = rsqrtf(wave_sum(x1 * x1 + x2 * x2) / (float)D + eps);
    const float n1 = rbf(bf2f(w[lane]) * rbf(x1 * rs));
    const float n2 = rbf(bf2f(w[lane + 64]) * rbf(x2 * rs));
    const float o1 = rbf(rbf(n1 * c1) + rbf(-n2 * s1));
    const float o2 = rbf(rbf(n2 * c2) + rbf(n1 * s2));
    bf16_t* o = isq ? Q + ((long)b * Hq + h) * D : Kc + (((long)b * Hkv + (h - Hq)) * Lmax + pos) * D;
    o[lane] = f2bf(o1);
    o[lane + 64] = f2bf(o2);
  }
}

// ------------------------------------------------------------------------------------------------ decode attention
// one block (4 waves) per (b, q-head); T = lens[b] + 1 keys. 16 lanes x 16 bytes cover one 128-element K or V row, so a
// wave handles 4 cache rows per load instruction (fully coalesced 1 KiB) and the block 16. Scores and probabilities
// live in LDS (dynamic: Lmax floats).
__global__ __launch_bounds__(256) void decode_attn_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                         const bf16_t* __restrict__ Vc, const int32_t* __restrict__ lens,
                                                         bf16_t* __restrict__ O, int Hq, int Hkv, int Lmax, float scale) {
  extern __shared__ float sc[];      // [Lmax] scores, then probabilities
  __shared__ float red[4];
  __shared__ float opart[4][D];
  const int b = blockIdx.x / Hq, h = blockIdx.x % Hq;
  const int hk = h / (Hq / Hkv);
  const int T = min(lens[b] + 1, Lmax);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int sub = lane >> 4, dl = (lane & 15) * 8;     // which of the wave's 4 rows, which 8 elements of the row
  float qf[8];
  unpack8(*(const u32x4*)(Q + ((long)b * Hq + h) * D + dl), qf);
  const bf16_t* Kb = Kc + ((long)b * Hkv + hk) * Lmax * D;
  const bf16_t* Vb = Vc + ((long)b * Hkv + hk) * Lmax * D;
  float mx = -INFINITY;
  for (int l0 = wid * 4; l0 < T; l0 += 64) {          // 4 row-quads per trip: their loads are issued together
    u32x4 kr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kr[i] = *(const u32x4*)(Kb + (long)min(l0 + 16 * i + sub, T - 1) * D + dl);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = l0 + 16 * i + sub;
      float kf[8];
      unpack8(kr[i], kf);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qf[j], kf[j], s);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      s *= scale;                     // fp32 scores, like the prefill path's QK^T GEMM (alpha applied in fp32)
      if (l < T) {
        if ((lane & 15) == 0) sc[l] = s;
        mx = fmaxf(mx, s);
      }
    }
  }
  mx = block_max<4>(mx, red);
  float sum = 0.f;
  for (int l = tid; l < T; l += 256) {
    const float e = __expf(sc[l] - mx);
    sc[l] = e;
    sum += e;
  }
  sum = block_sum<4>(sum, red);
  const float inv = 1.f / sum;
  __syncthreads();
  float of[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) of[j] = 0.f;
  for (int l0 = wid * 4; l0 < T; l0 += 64) {
    u32x4 vr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vr[i] = *(const u32x4*)(Vb + (long)min(l0 + 16 * i + sub, T - 1) * D + dl);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = l0 + 16 * i + sub;
      const float p = l < T ? rbf(sc[l] * inv) : 0.f;   // probabilities are a bf16 tensor before the PV product
      float vf[8];
      unpack8(vr[i], vf);
#pragma unroll
      for (int j = 0; j < 8; ++j) of[j] = fmaf(p, vf[j], of[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    of[j] += __shfl_xor(of[j], 16, 64);
    of[j] += __shfl_xor(of[j], 32, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) opart[wid][dl + j] = of[j];
  }
  __syncthreads();
  if (tid < D) O[((long)b * Hq + h) * D + tid] = f2bf(opart[0][tid] + opart[1][tid] + opart[2][tid] + opart[3][tid]);
}

// ------------------------------------------------------------------------------------------------ greedy pick
// transformers: RepetitionPenaltyLogitsProcessor, NoRepeatNGramLogitsProcessor, argmax, eos/pad handling of
// GenerationMixin._sample (greedy). Phase 1, grid (NCH, B): a block owns one slice of the vocabulary - it applies the
// penalty and the bans that fall into its slice (kept in a small LDS overlay, so the logits are read once and never
// rewritten) and reduces the slice to (best value, smallest index). Phase 2, grid B: combines the NCH partials and
// does the bookkeeping.
constexpr int PICK_NCH = 64;
constexpr int PICK_OVL = 1024;  // overlay capacity per vocabulary slice: <= 512 penalised + <= 512 banned entries

__global__ __launch_bounds__(256) void greedy_pick_partial_kernel(const bf16_t* __restrict__ logits, long ldl, int V,
                                                                  const int64_t* __restrict__ gen, int max_new,
                                                                  const int32_t* __restrict__ step_p, float penalty,
                                                                  int ngram, float* __restrict__ pval,
                                                                  int32_t* __restrict__ pidx, int32_t* __restrict__ overflow) {
  __shared__ int ov_id[PICK_OVL];
  __shared__ float ov_val[PICK_OVL];
  __shared__ int ov_n;
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const int step = *step_p;
  const int per = (V + PICK_NCH - 1) / PICK_NCH;
  const int c0 = c * per, c1 = min(V, c0 + per);
  const bf16_t* lg = logits + (long)b * ldl;
  const int64_t* g = gen + (long)b * max_new;
  if (tid == 0) ov_n = 0;
  __syncthreads();
  // penalty: distinct generated ids in this slice (first occurrence only: gather - rescale - scatter semantics)
  if (penalty != 1.0f) {
    for (int i = tid; i < step; i += 256) {
      const long t = g[i];
      if (t < c0 || t >= c1) continue;
      bool first = true;
      for (int j = 0; j < i; ++j) first &= (g[j] != t);
      if (!first) continue;
      const float sv = bf2f(lg[t]);
      const int at = atomicAdd(&ov_n, 1);
      if (at < PICK_OVL) { ov_id[at] = (int)t; ov_val[at] = sv < 0.f ? sv * penalty : sv / penalty; }
    }
  }
  __syncthreads();
  const int n_pen = min(ov_n, PICK_OVL);
  // n-gram ban overrides the penalised value: reuse the entry if the id is already in the overlay
  if (ngram > 0 && step + 1 >= ngram) {
    const int pre = ngram - 1;
    for (int i = tid; i + ngram <= step; i += 256) {
      const long t = g[i + pre];
      if (t < c0 || t >= c1) continue;
      bool match = true;
      for (int j = 0; j < pre; ++j) match &= (g[i + j] == g[step - pre + j]);
      if (!match) continue;
      bool found = false;
      for (int k = 0; k < n_pen; ++k)
        if (ov_id[k] == (int)t) { ov_val[k] = -INFINITY; found = true; }
      if (!found) {
        const int at = atomicAdd(&ov_n, 1);
        if (at < PICK_OVL) { ov_id[at] = (int)t; ov_val[at] = -INFINITY; }
      }
    }
  }
  __syncthreads();
  const int n_ov = ov_n;
  if (n_ov > PICK_OVL) {            // cannot happen for max_new <= PICK_OVL; flagged, never silently wrong
    if (tid == 0) atomicExch(overflow, 1);
    return;
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = c0 + tid; i < c1; i += 256) {
    float v = bf2f(lg[i]);
    for (int k = 0; k < n_ov; ++k)
      if (ov_id[k] == i) v = ov_val[k];       // duplicates of a banned id all hold -inf; penalised ids are unique
    if (v > best) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ovv = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ovv > best || (ovv == best && oi < bi)) { best = ovv; bi = oi; }
  }
  if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 4; ++k)
      if (rv[k] > best || (rv[k] == best && ri[k] < bi)) { best = rv[k]; bi = ri[k]; }
    pval[b * PICK_NCH + c] = best;
    pidx[b * PICK_NCH + c] = bi;
  }
}

__global__ __launch_bounds__(64) void greedy_pick_final_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                              int64_t* __restrict__ gen, int max_new,
                                                              const int32_t* __restrict__ step_p, int32_t* __restrict__ finished,
                                                              const int64_t* __restrict__ eos_ids, int n_eos, long pad_id,
                                                              int32_t* __restrict__ next_ids) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float best = pval[b * PICK_NCH + lane];
  int bi = pidx[b * PICK_NCH + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    const int step = *step_p;
    long tok = bi == 0x7fffffff ? 0 : bi;
    const bool fin = finished[b] != 0;
    if (fin) tok = pad_id;
    if (step < max_new) gen[(long)b * max_new + step] = tok;
    next_ids[b] = (int32_t)tok;
    bool is_eos = false;
    for (int k = 0; k < n_eos; ++k) is_eos |= (tok == eos_ids[k]);
    if (!fin && is_eos) finished[b] = 1;
  }
}

__global__ void advance_kernel(int32_t* lens, int B, int32_t* step) {
  if (lens && threadIdx.x < B) lens[threadIdx.x] += 1;
  if (step && threadIdx.x == 0) *step += 1;
}

template <int MT, int R, int XMODE>
void launch_skinny3(const void* x, const void* W, void* y, const void* res, const void* ln_w, float eps, int M, int N, int K,
                    long ldx, long ldw, long ldy, long ldr, int out_f32, hipStream_t s) {
  const int blocks = (N + 4 * R - 1) / (4 * R);
  hipLaunchKernelGGL((skinny_gemm_kernel<MT, R, XMODE>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)W,
                     y, (const bf16_t*)res, (const bf16_t*)ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32);
}
template <int MT, int R>
void launch_skinny2(int xmode, const void* x, const void* W, void* y, const void* res, const void* ln_w, float eps, int M,
                    int N, int K, long ldx, long ldw, long ldy, long ldr, int out_f32, hipStream_t s) {
  if (xmode == 0) launch_skinny3<MT, R, 0>(x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (xmode == 1) launch_skinny3<MT, R, 1>(x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else launch_skinny3<MT, R, 2>(x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
}
template <int MT, int R, int KS>
void launch_regx(int xmode, const void* x, const void* W, void* y, const void* res, const void* ln_w, float eps, int M, int N,
                 int K, long ldx, long ldw, long ldy, long ldr, int out_f32, hipStream_t s) {
  const int rows_per_block = (4 / KS) * R;
  const dim3 grid((N + rows_per_block - 1) / rows_per_block), block(256);
#define VQ3_REGX(XM)                                                                                                      \
  hipLaunchKernelGGL((skinny_regx_kernel<MT, R, KS, XM>), grid, block, 0, s, (const bf16_t*)x, (const bf16_t*)W, y,        \
                     (const bf16_t*)res, (const bf16_t*)ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32)
  if (xmode == 0) VQ3_REGX(0);
  else if (xmode == 1) VQ3_REGX(1);
  else VQ3_REGX(2);
#undef VQ3_REGX
}
template <int MT, int KS>
void launch_regx_r(int xmode, const void* x, const void* W, void* y, const void* res, const void* ln_w, float eps, int M, int N,
                   int K, long ldx, long ldw, long ldy, long ldr, int out_f32, hipStream_t s) {
  // rows per wave: every wave keeps R * CH 16-byte loads in flight, so few waves per CU already saturate HBM; larger R
  // divides the x traffic from L2 and the per-wave x preparation (RMSNorm / SwiGLU) by R
  static int force = -1;
  if (force < 0) {
    const char* e = getenv("VQ3_SKINNY_R");
    force = e ? atoi(e) : 0;
  }
  const int r = force ? force : ((xmode != 0 || (long)N * KS >= 8192) ? (N >= 1024 ? 4 : (N >= 512 ? 2 : 1)) : 1);
  if (r >= 4) launch_regx<MT, 4, KS>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (r == 2) launch_regx<MT, 2, KS>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else launch_regx<MT, 1, KS>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
}
// generic kernel: rows per wave so that enough waves cover the chip (256 CUs x >= 8 waves) before growing R
template <int MT>
void launch_skinny(int xmode, const void* x, const void* W, void* y, const void* res, const void* ln_w, float eps, int M,
                   int N, int K, long ldx, long ldw, long ldy, long ldr, int out_f32, hipStream_t s) {
  if constexpr (MT <= 2) {
    const int chunks = (K + 511) / 512;
    if (chunks <= 5) return launch_regx_r<MT, 1>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
    if (chunks <= 10) return launch_regx_r<MT, 2>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
    if (chunks <= 20) return launch_regx_r<MT, 4>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
    const int r = N >= 16384 ? 4 : (N >= 4096 ? 2 : 1);
    if (r == 4) launch_skinny2<MT, 4>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
    else if (r == 2) launch_skinny2<MT, 2>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
    else launch_skinny2<MT, 1>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  } else {
    launch_skinny2<MT, 4>(xmode, x, W, y, res, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  }
}

template <int MT, int KS>
void launch_fp8_skinny(int xmode, const void* x, const void* W, const float* ws, void* y, const void* res, const void* ln_w,
                       float eps, int M, int N, int K, long ldx, long ldw, long ldy, long ldr, hipStream_t s) {
  constexpr int R = 4;
  const int rows_per_block = (4 / KS) * R;
  const dim3 grid((N + rows_per_block - 1) / rows_per_block), block(256);
#define VQ3_F8(XM)                                                                                                          \
  hipLaunchKernelGGL((skinny_fp8_kernel<MT, R, KS, XM>), grid, block, 0, s, (const bf16_t*)x, (const uint8_t*)W, ws,        \
                     (bf16_t*)y, (const bf16_t*)res, (const bf16_t*)ln_w, eps, M, N, K, ldx, ldw, ldy, ldr)
  if (xmode == 0) VQ3_F8(0);
  else if (xmode == 1) VQ3_F8(1);
  else VQ3_F8(2);
#undef VQ3_F8
}

}  // namespace

extern "C" int vq3_skinny_gemm_bf16(const void* x, const void* W, void* y, const void* residual, const void* ln_w,
                                    float eps, int32_t xmode, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldw,
                                    int64_t ldy, int64_t ldr, int32_t out_f32, void* stream) {
  VQ3_CHECK_ARG(x && W && y, "skinny_gemm: null pointer");
  VQ3_CHECK_ARG(M >= 1 && M <= 8, "skinny_gemm: M must be in [1, 8], got %d (use vq3_gemm_bf16_nt)", M);
  VQ3_CHECK_ARG(N > 0 && K > 0 && K % 8 == 0, "skinny_gemm: need N > 0 and K %% 8 == 0 (N=%d K=%d)", N, K);
  VQ3_CHECK_ARG(xmode >= 0 && xmode <= 2 && (xmode != 1 || ln_w), "skinny_gemm: xmode must be 0, 1 (needs ln_w) or 2");
  VQ3_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= (xmode == 2 ? 2l * K : (long)K) && ldw >= K,
                "skinny_gemm: row strides must cover the row and be multiples of 8");
  VQ3_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)W % 16 == 0) && (!ln_w || (uintptr_t)ln_w % 16 == 0),
                "skinny_gemm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (M == 1) launch_skinny<1>(xmode, x, W, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (M == 2) launch_skinny<2>(xmode, x, W, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (M <= 4) launch_skinny<4>(xmode, x, W, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else launch_skinny<8>(xmode, x, W, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  VQ3_CHECK_LAUNCH("skinny_gemm");
  return 0;
}

extern "C" int vq3_skinny_gemm_fp8(const void* x, const void* Wq, const float* w_scale, void* y, const void* residual,
                                   const void* ln_w, float eps, int32_t xmode, int32_t M, int32_t N, int32_t K, int64_t ldx,
                                   int64_t ldw, int64_t ldy, int64_t ldr, void* stream) {
  VQ3_CHECK_ARG(x && Wq && w_scale && y, "skinny_gemm_fp8: null pointer");
  VQ3_CHECK_ARG(M >= 1 && M <= 2, "skinny_gemm_fp8: M must be 1 or 2, got %d (prefill / larger batches use vq3_gemm_fp8_nt)", M);
  VQ3_CHECK_ARG(N > 0 && K > 0 && K % 16 == 0 && K <= 12288, "skinny_gemm_fp8: need K %% 16 == 0 and K <= 12288 (N=%d K=%d)", N, K);
  VQ3_CHECK_ARG(xmode >= 0 && xmode <= 2 && (xmode != 1 || ln_w), "skinny_gemm_fp8: xmode must be 0, 1 (needs ln_w) or 2");
  VQ3_CHECK_ARG(ldx % 8 == 0 && ldw % 16 == 0 && ldx >= (xmode == 2 ? 2l * K : (long)K) && ldw >= K,
                "skinny_gemm_fp8: row strides must cover the row (x: multiple of 8, W: multiple of 16)");
  VQ3_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)Wq % 16 == 0) && (!ln_w || (uintptr_t)ln_w % 16 == 0),
                "skinny_gemm_fp8: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (K + 1023) / 1024;
#define VQ3_GO(MT)                                                                                                                  \
  do {                                                                                                                              \
    if (chunks <= 3) launch_fp8_skinny<MT, 1>(xmode, x, Wq, w_scale, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, s);       \
    else if (chunks <= 6) launch_fp8_skinny<MT, 2>(xmode, x, Wq, w_scale, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, s);  \
    else launch_fp8_skinny<MT, 4>(xmode, x, Wq, w_scale, y, residual, ln_w, eps, M, N, K, ldx, ldw, ldy, ldr, s);                   \
  } while (0)
  if (M == 1) VQ3_GO(1);
  else VQ3_GO(2);
#undef VQ3_GO
  VQ3_CHECK_LAUNCH("skinny_gemm_fp8");
  return 0;
}

extern "C" int vq3_qwen_decode_qkprep(const void* qkv, const void* q_w, const void* k_w, const void* cos, const void* sin,
                                      const int32_t* lens, void* Q, void* Kcache, void* Vcache, int32_t B, int32_t Hq,
                                      int32_t Hkv, int32_t Dh, int32_t Lmax, float eps, void* stream) {
  VQ3_CHECK_ARG(qkv && q_w && k_w && cos && sin && lens && Q && Kcache && Vcache, "decode_qkprep: null pointer");
  VQ3_CHECK_ARG(Dh == D, "decode_qkprep: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Lmax > 0, "decode_qkprep: bad shape");
  hipLaunchKernelGGL(decode_qkprep_kernel, dim3(B, (Hq + 2 * Hkv + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
                     (const bf16_t*)q_w, (const bf16_t*)k_w, (const bf16_t*)cos, (const bf16_t*)sin, lens, (bf16_t*)Q,
                     (bf16_t*)Kcache, (bf16_t*)Vcache, Hq, Hkv, Lmax, eps);
  VQ3_CHECK_LAUNCH("decode_qkprep");
  return 0;
}

extern "C" int vq3_qwen_decode_attn(const void* Q, const void* Kcache, const void* Vcache, const int32_t* lens, void* O,
                                    int32_t B, int32_t Hq, int32_t Hkv, int32_t Dh, int32_t Lmax, float scale,
                                    void* stream) {
  VQ3_CHECK_ARG(Q && Kcache && Vcache && lens && O, "decode_attn: null pointer");
  VQ3_CHECK_ARG(Dh == D, "decode_attn: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "decode_attn: bad shape");
  VQ3_CHECK_ARG(Lmax >= 64 && Lmax <= 12288, "decode_attn: cache capacity %d outside [64, 12288]", Lmax);
  hipLaunchKernelGGL(decode_attn_kernel, dim3(B * Hq), dim3(256), (size_t)Lmax * sizeof(float), (hipStream_t)stream,
                     (const bf16_t*)Q, (const bf16_t*)Kcache, (const bf16_t*)Vcache, lens, (bf16_t*)O, Hq, Hkv, Lmax, scale);
  VQ3_CHECK_LAUNCH("decode_attn");
  return 0;
}

extern "C" int vq3_greedy_pick(const void* logits_bf16, int64_t ld_logits, float* work, int32_t B, int32_t V,
                               int64_t* generated, int32_t max_new, const int32_t* step, int32_t* finished,
                               float repetition_penalty, int32_t no_repeat_ngram, const int64_t* eos_ids, int32_t n_eos,
                               int64_t pad_id, int32_t* next_ids, void* stream) {
  VQ3_CHECK_ARG(logits_bf16 && work && generated && step && finished && next_ids, "greedy_pick: null pointer");
  VQ3_CHECK_ARG(B > 0 && V > 0 && max_new > 0 && ld_logits >= V, "greedy_pick: bad shape");
  VQ3_CHECK_ARG(max_new <= PICK_OVL / 2, "greedy_pick: at most %d ids per row (prompt + new), got %d", PICK_OVL / 2, max_new);
  VQ3_CHECK_ARG(repetition_penalty > 0.f, "greedy_pick: repetition_penalty must be > 0, got %f", (double)repetition_penalty);
  VQ3_CHECK_ARG(no_repeat_ngram >= 0 && n_eos >= 0 && (n_eos == 0 || eos_ids), "greedy_pick: bad ngram / eos arguments");
  // work: [B * 64] partial values, [B * 64] partial indices, 1 overflow flag (f32 / i32 slots)
  float* pval = work;
  int32_t* pidx = (int32_t*)(work + (long)B * PICK_NCH);
  int32_t* overflow = pidx + (long)B * PICK_NCH;
  hipLaunchKernelGGL(greedy_pick_partial_kernel, dim3(PICK_NCH, B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)logits_bf16, (long)ld_logits, V, generated, max_new, step, repetition_penalty,
                     no_repeat_ngram, pval, pidx, overflow);
  hipLaunchKernelGGL(greedy_pick_final_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pval, pidx, generated, max_new,
                     step, finished, eos_ids, n_eos, (long)pad_id, next_ids);
  VQ3_CHECK_LAUNCH("greedy_pick");
  return 0;
}

extern "C" int vq3_decode_advance(int32_t* lens, int32_t B, int32_t* step, void* stream) {
  VQ3_CHECK_ARG((lens || step) && B > 0 && B <= 64, "decode_advance: bad argument");
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lens, B, step);
  VQ3_CHECK_LAUNCH("decode_advance");
  return 0;
}
