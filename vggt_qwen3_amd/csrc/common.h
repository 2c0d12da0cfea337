// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the VGGT -> Perceiver -> Qwen3 path.
// Wave = 64 lanes everywhere; bf16 is carried as raw uint16 and converted through the native __bf16 type
// (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VQ3_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, (uint32_t)v << 16);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// round a float through bf16 (emulates a PyTorch op that returns a bf16 tensor)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN). Written as
// two scalar casts + shift + or, hipcc 7.2 emits two conversions, a shift and an SDWA or - four instructions for every pair of every
// epilogue and element-wise kernel.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 vq3_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, vq3_bf16x2_t));
}
__device__ __forceinline__ f32x2_t unpack2bf(uint32_t u) {
  return f32x2_t{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
// a pair rounded through bf16 (three instructions: the conversion, a shift and a mask)
__device__ __forceinline__ f32x2_t rbf2(f32x2_t v) {
  const uint32_t u = pack2bf(v[0], v[1]);
  return f32x2_t{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of NW waves. `red` must hold >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Exact-form (erf) GELU, torch.nn.GELU()'s default, two values per VALU instruction: GELU(x) = x * Phi(x) with
//   Phi(-a) = 2^q(a),  a = min(|x|, 5.5),  q = degree-7 fit of log2 Phi(-a) on [0, 5.5]      (Phi(x) = 1 - Phi(-x) for x > 0)
// log2 Phi(-a) is smooth (it does not saturate as erf does), so 7 packed FMAs + one v_exp_f32 give |error| <= 7e-7 absolute and
// 1.2e-5 relative for |x| < 3 (Phi of the negative tail is the quantity computed directly, so it keeps its relative accuracy).
// This is the form for f32 results; values that are rounded to bf16 right after take gelu_erf2_b below.
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t a = {fminf(fabsf(x[0]), 5.5f), fminf(fabsf(x[1]), 5.5f)};
  f32x2_t q = {-1.921234625e-06f, -1.921234625e-06f};
  q = __builtin_elementwise_fma(q, a, f32x2_t{6.328849712e-05f, 6.328849712e-05f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-9.434208502e-04f, -9.434208502e-04f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{8.556272268e-03f, 8.556272268e-03f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-5.405228293e-02f, -5.405228293e-02f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-4.583817849e-01f, -4.583817849e-01f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-1.151278331e+00f, -1.151278331e+00f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-9.999938561e-01f, -9.999938561e-01f});
  const f32x2_t h = {0.5f - __builtin_amdgcn_exp2f(q[0]), 0.5f - __builtin_amdgcn_exp2f(q[1])};
  const f32x2_t phi = {0.5f + copysignf(h[0], x[0]), 0.5f + copysignf(h[1], x[1])};
  return x * phi;
}
// The same function for values that are ROUNDED TO BF16 right after (GEMM epilogues with a bf16 output, gelu_fwd): GELU(x) = relu(x) - a * Phi(-a) with
//   Phi(-a) = 2^q(a),  a = min(|x|, 5.5),  q = degree-5 fit of log2 Phi(-a) on [0, 5.5]      (x > 0: x (1 - Phi(-a)) = x - a Phi(-a))
// log2 Phi(-a) is smooth (it does not saturate as erf does), so 5 FMAs + one v_exp_f32 give |q error| <= 3.7e-4: GELU within 2.5e-4
// relative (Phi of the negative tail is the quantity computed directly, so it keeps its relative accuracy) = 0.13 ulp of the bf16 the
// result is rounded to right after; 3.4e-5 absolute. relu(x) is formed as (x + |x|) / 2 so that a NaN stays a NaN (v_max would drop it).
// No v_rcp, no log2(e) multiply, none of libm erff's range checks: on the VGGT fc1 GEMM (25 M outputs per launch, one workgroup per
// CU: nothing overlaps the epilogue) the activation is what the tile's tail costs, and the VALU's ISSUE slots are what it costs in
// (packed f32 and v_exp_f32 take two each): 11 slots per value here against 14 for the degree-7 form with phi = 0.5 + copysign(0.5 - e, x)
// (round 2-4; 1.2e-5 relative) and ~30 for the Abramowitz-Stegun 7.1.26 form (v_rcp + v_exp + 12 scalar-lane ops).
__device__ __forceinline__ f32x2_t gelu_erf2_b(f32x2_t x) {
  const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2_t a = {fminf(ax[0], 5.5f), fminf(ax[1], 5.5f)};
  f32x2_t q = {-2.3323995992541313e-04f, -2.3323995992541313e-04f};
  q = __builtin_elementwise_fma(q, a, f32x2_t{4.869441967457533e-03f, 4.869441967457533e-03f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-4.4596340507268906e-02f, -4.4596340507268906e-02f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-4.6968939900398254e-01f, -4.6968939900398254e-01f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-1.1462510824203491e+00f, -1.1462510824203491e+00f});
  q = __builtin_elementwise_fma(q, a, f32x2_t{-1.0003620386123657e+00f, -1.0003620386123657e+00f});
  const f32x2_t t = {a[0] * __builtin_amdgcn_exp2f(q[0]), a[1] * __builtin_amdgcn_exp2f(q[1])};
  return __builtin_elementwise_fma(x + ax, f32x2_t{0.5f, 0.5f}, -t);
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2_t{x, x})[0]; }
__device__ __forceinline__ float gelu_erf_b(float x) { return gelu_erf2_b(f32x2_t{x, x})[0]; }
// sigmoid through v_rcp_f32 (1 ulp) instead of the correctly rounded f32 division hipcc emits for `1.f / y` (v_div_scale x 2, v_rcp,
// four FMAs, v_div_fmas, v_div_fixup: ~10 VALU issue slots per value in epilogues whose results are rounded to bf16 right after - the
// SwiGLU forward / backward tails of the two widest text GEMMs). Every SiLU / sigmoid of the library goes through these two, so fused and
// unfused forms of an op stay bit-identical to each other.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// Counter-based dropout decision (vq3_dropout, the fused Perceiver cross-attention): 24 uniform bits from (seed, element index);
// an element is kept when they are >= (unsigned)(p * 2^24). Stateless, so a re-run with the same (seed, offset) repeats the mask.
__device__ __forceinline__ unsigned drop_bits(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 40);      // 24 uniform bits
}

// ---- host side error plumbing (C ABI returns int, message kept per thread) ----
#ifdef __cplusplus
extern "C" const char* vq3_last_error(void);
#endif
void vq3_set_error(const char* fmt, ...);

#define VQ3_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      vq3_set_error(__VA_ARGS__);           \
      return 1;                             \
    }                                       \
  } while (0)

#define VQ3_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      vq3_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
