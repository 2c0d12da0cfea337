// Device-side batch builder for the step right before the model (reference: src/dataio/collate_multiview.py:12-19
// image transform, :56-79 id/label/mask layout). Integer / byte work, HBM-bound, results bit-exact:
//
//  * vq3_resample_plan (host, no GPU): Pillow's bicubic resampling plan - per output index the first source index,
//    the tap count and the taps as 22-bit fixed point - computed exactly as Pillow's Resample.c does (double
//    arithmetic, antialiasing support scaled by the shrink factor, round-half-away fixed-point conversion).
//  * vq3_resize_crop_u8: Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor() for a whole batch of uint8 HWC images
//    in ONE launch. A workgroup owns a TY x 64 tile of one output image: the horizontal pass runs over only the
//    source rows that tile's vertical taps touch and lands in LDS as uint8 (Pillow rounds the intermediate image to
//    uint8, so must we), the vertical pass reads LDS and writes the three fp32 planes (value / 255, IEEE division
//    like torch's). The intermediate image never exists in HBM and source pixels outside the crop are never read.
//  * vq3_pack_tokens: ragged prompt / answer id lists -> padded input_ids, labels (-100 on prompt and padding),
//    attention_mask (ids != pad), all int64 [B, L].
#include <cmath>
#include <vector>

#include "common.h"
#include "vq3_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int TX = 64;

#pragma STDC FP_CONTRACT OFF
inline double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid: (ceil(S / 64), ceil(S / TY), n_images); block 256; dynamic LDS = max_rows * 64 * 4 bytes
__global__ __launch_bounds__(256) void resize_crop_kernel(const vq3_image_desc* __restrict__ descs,
                                                         const int32_t* __restrict__ coefs,
                                                         const int32_t* __restrict__ bounds, float* __restrict__ out,
                                                         int S, int TY, int max_rows) {
  extern __shared__ uint32_t tile[];  // [rows][64] packed r | g<<8 | b<<16
  const vq3_image_desc d = descs[blockIdx.z];
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int ny = min(TY, S - y0);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int32_t* bv = bounds + d.bv_off;
  const int32_t* bh = bounds + d.bh_off;
  // source rows this tile's vertical taps touch (bounds are monotone in the output index)
  const int oy_first = d.crop_y + y0, oy_last = d.crop_y + y0 + ny - 1;
  const int rmin = bv[2 * oy_first];
  const int rmax = bv[2 * oy_last] + bv[2 * oy_last + 1];
  const int nrows = rmax - rmin;  // <= max_rows by construction of the launch
  const int ox = d.crop_x + x0 + tx;  // output column in the resized (uncropped) image
  const bool xin = x0 + tx < S;
  if (xin) {
    const int xmin = bh[2 * ox], xn = bh[2 * ox + 1];
    const int32_t* k = coefs + d.kh_off + (long)ox * d.ksize_h;
    for (int r = ty; r < nrows; r += 4) {
      const uint8_t* row = d.src + (long)(rmin + r) * d.pitch + (long)xmin * 3;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int t = 0; t < xn; ++t) {
        const int kk = k[t];
        s0 += (int)row[3 * t + 0] * kk;
        s1 += (int)row[3 * t + 1] * kk;
        s2 += (int)row[3 * t + 2] * kk;
      }
      tile[r * TX + tx] = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16);
    }
  }
  __syncthreads();
  if (!xin) return;
  float* o = out + (long)blockIdx.z * 3 * S * S;
  for (int y = ty; y < ny; y += 4) {
    const int oy = d.crop_y + y0 + y;
    const int ymin = bv[2 * oy] - rmin, yn = bv[2 * oy + 1];
    const int32_t* k = coefs + d.kv_off + (long)oy * d.ksize_v;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < yn; ++t) {
      const uint32_t p = tile[(ymin + t) * TX + tx];
      const int kk = k[t];
      s0 += (int)(p & 255u) * kk;
      s1 += (int)((p >> 8) & 255u) * kk;
      s2 += (int)((p >> 16) & 255u) * kk;
    }
    const long at = (long)(y0 + y) * S + x0 + tx;
    o[at] = (float)clip8(s0) / 255.0f;
    o[at + (long)S * S] = (float)clip8(s1) / 255.0f;
    o[at + 2l * S * S] = (float)clip8(s2) / 255.0f;
  }
}

__global__ __launch_bounds__(256) void pack_tokens_kernel(const int32_t* __restrict__ prompt, const int32_t* __restrict__ poff,
                                                         const int32_t* __restrict__ answer, const int32_t* __restrict__ aoff,
                                                         int L, int max_length, long pad_id, int64_t* __restrict__ ids,
                                                         int64_t* __restrict__ labels, int64_t* __restrict__ mask) {
  const int b = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const int np = poff[b + 1] - poff[b], na = aoff[b + 1] - aoff[b];
  const int n = min(np + na, max_length);
  long id = pad_id, lab = -100;
  if (l < n) {
    if (l < np) {
      id = prompt[poff[b] + l];
    } else {
      id = answer[aoff[b] + l - np];
      lab = id;
    }
  }
  const long at = (long)b * L + l;
  ids[at] = id;
  labels[at] = lab;
  mask[at] = id != pad_id ? 1 : 0;
}

}  // namespace

extern "C" int vq3_resample_ksize(int32_t in_size, int32_t out_size) {
  if (in_size <= 0 || out_size <= 0) return -1;
  double filterscale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  return (int)std::ceil(support) * 2 + 1;
}

extern "C" int vq3_resample_plan(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs) {
  VQ3_CHECK_ARG(in_size > 0 && out_size > 0 && bounds && coefs, "resample_plan: bad argument");
  const float in0 = 0.0f, in1 = (float)in_size;
  double scale, filterscale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
    }
    for (; x < ksize; ++x) k[x] = 0;
    for (x = 0; x < ksize; ++x) {
      const double v = k[x];
      coefs[(long)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return 0;
}

extern "C" int vq3_resize_crop_u8(const vq3_image_desc* descs_dev, int32_t n_images, const int32_t* coefs_dev,
                                  const int32_t* bounds_dev, float* out, int32_t S, int32_t tile_rows,
                                  int32_t max_src_rows, void* stream) {
  VQ3_CHECK_ARG(descs_dev && coefs_dev && bounds_dev && out, "resize_crop_u8: null pointer");
  VQ3_CHECK_ARG(n_images > 0 && n_images <= 65535 && S > 0, "resize_crop_u8: bad batch/size (%d images, S=%d)", n_images, S);
  VQ3_CHECK_ARG(tile_rows >= 1 && tile_rows <= 64, "resize_crop_u8: tile_rows must be in [1, 64], got %d", tile_rows);
  const long lds = (long)max_src_rows * TX * 4;
  VQ3_CHECK_ARG(max_src_rows > 0 && lds <= 64 * 1024,
                "resize_crop_u8: %d source rows per tile need %ld B of LDS (> 64 KiB): use fewer tile_rows", max_src_rows, lds);
  dim3 grid((S + TX - 1) / TX, (S + tile_rows - 1) / tile_rows, n_images);
  hipLaunchKernelGGL(resize_crop_kernel, grid, dim3(256), (size_t)lds, (hipStream_t)stream, descs_dev, coefs_dev,
                     bounds_dev, out, S, tile_rows, max_src_rows);
  VQ3_CHECK_LAUNCH("resize_crop_u8");
  return 0;
}

extern "C" int vq3_pack_tokens(const int32_t* prompt_ids, const int32_t* prompt_off, const int32_t* answer_ids,
                               const int32_t* answer_off, int32_t B, int32_t L, int32_t max_length, int64_t pad_id,
                               int64_t* input_ids, int64_t* labels, int64_t* attention_mask, void* stream) {
  VQ3_CHECK_ARG(prompt_ids && prompt_off && answer_ids && answer_off && input_ids && labels && attention_mask,
                "pack_tokens: null pointer");
  VQ3_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && max_length > 0, "pack_tokens: bad shape B=%d L=%d max_length=%d", B, L, max_length);
  hipLaunchKernelGGL(pack_tokens_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, prompt_ids,
                     prompt_off, answer_ids, answer_off, L, max_length, (long)pad_id, input_ids, labels, attention_mask);
  VQ3_CHECK_LAUNCH("pack_tokens");
  return 0;
}
