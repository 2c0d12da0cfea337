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

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
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

// erf to 1.5e-7 absolute (Abramowitz-Stegun 7.1.26): one v_exp, one v_rcp, 5 FMAs - libm's erff costs ~10x more and
// the result is rounded to bf16 (eps 4e-3) right after, so nothing observable changes.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
// exact-form (erf) GELU as torch.nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

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
