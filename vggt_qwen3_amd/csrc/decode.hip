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
template <int MT, int R>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                         void* __restrict__ y, const bf16_t* __restrict__ res, int M, int N,
                                                         int K, long ldx, long ldw, long ldy, long ldr, int out_f32) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wid) * R;
  if (n0 >= N) return;
  float acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
  const bf16_t* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = W + (long)min(n0 + r, N - 1) * ldw;
  for (int k = lane * 8; k < K; k += 512) {  // K % 8 == 0
    u32x4 wv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wv[r] = *(const u32x4*)(wrow[r] + k);
    float xf[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const u32x4 xv = *(const u32x4*)(x + (long)min(m, M - 1) * ldx + k);
      unpack8(xv, xf[m]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float wf[8];
      unpack8(wv[r], wf);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][m] = fmaf(wf[j], xf[m][j], acc[r][m]);
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

// ------------------------------------------------------------------------------------------------ decode q/k prep
// one block per row b; wave w handles heads w, w+4, ...; lane i holds elements i and i+64 (the rotate_half pair)
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
  for (int h = wid; h < HT; h += 4) {
    const float x1 = bf2f(row[h * D + lane]), x2 = bf2f(row[h * D + lane + 64]);
    if (h >= Hq + Hkv) {
      bf16_t* o = Vc + (((long)b * Hkv + (h - Hq - Hkv)) * Lmax + pos) * D;
      o[lane] = f2bf(x1);
      o[lane + 64] = f2bf(x2);
      continue;
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
// one block (256 threads) per (b, q-head). T = lens[b] + 1 keys. scores in LDS (dynamic: Lmax floats).
__global__ __launch_bounds__(256) void decode_attn_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kc,
                                                         const bf16_t* __restrict__ Vc, const int32_t* __restrict__ lens,
                                                         bf16_t* __restrict__ O, int Hq, int Hkv, int Lmax, float scale) {
  extern __shared__ float sc[];      // [Lmax] scores / probabilities, then [2][D] partial outputs
  __shared__ float qs[D];
  __shared__ float red[4];
  const int b = blockIdx.x / Hq, h = blockIdx.x % Hq;
  const int hk = h / (Hq / Hkv);
  const int T = min(lens[b] + 1, Lmax);
  const int tid = threadIdx.x;
  if (tid < D) qs[tid] = bf2f(Q[((long)b * Hq + h) * D + tid]);
  __syncthreads();
  const bf16_t* Kb = Kc + ((long)b * Hkv + hk) * Lmax * D;
  const bf16_t* Vb = Vc + ((long)b * Hkv + hk) * Lmax * D;
  float mx = -INFINITY;
  for (int l = tid; l < T; l += 256) {
    const u32x4* kr = (const u32x4*)(Kb + (long)l * D);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
      float kf[8];
      unpack8(kr[c], kf);
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qs[c * 8 + j], kf[j], s);
    }
    s *= scale;                       // fp32 scores, like the prefill path's QK^T GEMM (alpha applied in fp32)
    sc[l] = s;
    mx = fmaxf(mx, s);
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
  // O[d] = sum_l p[l] V[l, d]: thread = (half, d); p rounded to bf16 like softmax(...).to(bf16)
  const int d = tid & (D - 1), half = tid >> 7;
  float o = 0.f;
  for (int l = half; l < T; l += 2) o = fmaf(rbf(sc[l] * inv), bf2f(Vb[(long)l * D + d]), o);
  __syncthreads();
  float* part = sc;  // reuse
  if (half == 1) part[d] = o;
  __syncthreads();
  if (half == 0) O[((long)b * Hq + h) * D + d] = f2bf(o + part[d]);
}

// ------------------------------------------------------------------------------------------------ greedy pick
// transformers: RepetitionPenaltyLogitsProcessor, NoRepeatNGramLogitsProcessor, argmax, eos/pad handling of
// GenerationMixin._sample (greedy). One block per row; `work` is an f32 scratch row [V].
__global__ __launch_bounds__(1024) void greedy_pick_kernel(const bf16_t* __restrict__ logits, long ldl, float* __restrict__ work,
                                                          int V, int64_t* __restrict__ gen, int max_new,
                                                          const int32_t* __restrict__ step_p, int32_t* __restrict__ finished,
                                                          float penalty, int ngram, const int64_t* __restrict__ eos_ids,
                                                          int n_eos, long pad_id, int32_t* __restrict__ next_ids) {
  __shared__ float rv[16];
  __shared__ int ri[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int step = *step_p;          // tokens generated so far for every row
  float* w = work + (long)b * V;
  const bf16_t* lg = logits + (long)b * ldl;
  int64_t* g = gen + (long)b * max_new;
  for (int i = tid; i < V; i += 1024) w[i] = bf2f(lg[i]);
  __syncthreads();
  // repetition penalty over the distinct generated ids (gather - rescale - scatter: duplicates count once)
  if (penalty != 1.0f) {
    for (int i = tid; i < step; i += 1024) {
      const long t = g[i];
      bool first = true;
      for (int j = 0; j < i; ++j) first &= (g[j] != t);
      if (first && t >= 0 && t < V) {
        const float s = bf2f(lg[t]);
        w[t] = s < 0.f ? s * penalty : s / penalty;
      }
    }
    __syncthreads();
  }
  // n-gram ban: any token that would complete an n-gram already present in the generated ids
  if (ngram > 0 && step + 1 >= ngram) {
    const int pre = ngram - 1;      // the last `pre` generated tokens form the prefix
    for (int i = tid; i + ngram <= step; i += 1024) {
      bool match = true;
      for (int j = 0; j < pre; ++j) match &= (g[i + j] == g[step - pre + j]);
      if (match) {
        const long t = g[i + pre];
        if (t >= 0 && t < V) w[t] = -INFINITY;
      }
    }
    __syncthreads();
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < V; i += 1024) {
    const float v = w[i];
    if (v > best) { best = v; bi = i; }          // strided scan keeps the smallest index within a thread on ties
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { rv[tid >> 6] = best; ri[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 16; ++k)
      if (rv[k] > best || (rv[k] == best && ri[k] < bi)) { best = rv[k]; bi = ri[k]; }
    long tok = bi == 0x7fffffff ? 0 : bi;
    const bool fin = finished[b] != 0;
    if (fin) tok = pad_id;
    if (step < max_new) g[step] = tok;
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

template <int MT>
int launch_skinny(const void* x, const void* W, void* y, const void* res, int M, int N, int K, long ldx, long ldw, long ldy,
                  long ldr, int out_f32, hipStream_t s) {
  constexpr int R = 4;
  const int blocks = (N + 4 * R - 1) / (4 * R);
  hipLaunchKernelGGL((skinny_gemm_kernel<MT, R>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)W, y,
                     (const bf16_t*)res, M, N, K, ldx, ldw, ldy, ldr, out_f32);
  return 0;
}

}  // namespace

extern "C" int vq3_skinny_gemm_bf16(const void* x, const void* W, void* y, const void* residual, int32_t M, int32_t N,
                                    int32_t K, int64_t ldx, int64_t ldw, int64_t ldy, int64_t ldr, int32_t out_f32,
                                    void* stream) {
  VQ3_CHECK_ARG(x && W && y, "skinny_gemm: null pointer");
  VQ3_CHECK_ARG(M >= 1 && M <= 8, "skinny_gemm: M must be in [1, 8], got %d (use vq3_gemm_bf16_nt)", M);
  VQ3_CHECK_ARG(N > 0 && K > 0 && K % 8 == 0, "skinny_gemm: need N > 0 and K %% 8 == 0 (N=%d K=%d)", N, K);
  VQ3_CHECK_ARG(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "skinny_gemm: row strides must be >= K and multiples of 8");
  VQ3_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)W % 16 == 0), "skinny_gemm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (M == 1) launch_skinny<1>(x, W, y, residual, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (M == 2) launch_skinny<2>(x, W, y, residual, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else if (M <= 4) launch_skinny<4>(x, W, y, residual, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  else launch_skinny<8>(x, W, y, residual, M, N, K, ldx, ldw, ldy, ldr, out_f32, s);
  VQ3_CHECK_LAUNCH("skinny_gemm");
  return 0;
}

extern "C" int vq3_qwen_decode_qkprep(const void* qkv, const void* q_w, const void* k_w, const void* cos, const void* sin,
                                      const int32_t* lens, void* Q, void* Kcache, void* Vcache, int32_t B, int32_t Hq,
                                      int32_t Hkv, int32_t Dh, int32_t Lmax, float eps, void* stream) {
  VQ3_CHECK_ARG(qkv && q_w && k_w && cos && sin && lens && Q && Kcache && Vcache, "decode_qkprep: null pointer");
  VQ3_CHECK_ARG(Dh == D, "decode_qkprep: head_dim must be %d, got %d", D, Dh);
  VQ3_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Lmax > 0, "decode_qkprep: bad shape");
  hipLaunchKernelGGL(decode_qkprep_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
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
  VQ3_CHECK_ARG(repetition_penalty > 0.f, "greedy_pick: repetition_penalty must be > 0, got %f", (double)repetition_penalty);
  VQ3_CHECK_ARG(no_repeat_ngram >= 0 && n_eos >= 0 && (n_eos == 0 || eos_ids), "greedy_pick: bad ngram / eos arguments");
  hipLaunchKernelGGL(greedy_pick_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits_bf16,
                     (long)ld_logits, work, V, generated, max_new, step, finished, repetition_penalty, no_repeat_ngram,
                     eos_ids, n_eos, (long)pad_id, next_ids);
  VQ3_CHECK_LAUNCH("greedy_pick");
  return 0;
}

extern "C" int vq3_decode_advance(int32_t* lens, int32_t B, int32_t* step, void* stream) {
  VQ3_CHECK_ARG((lens || step) && B > 0 && B <= 64, "decode_advance: bad argument");
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lens, B, step);
  VQ3_CHECK_LAUNCH("decode_advance");
  return 0;
}
