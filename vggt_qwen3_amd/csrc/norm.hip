// RMSNorm (fwd/bwd) and LayerNorm (fwd) for gfx950. HBM-bound row kernels: one 64-lane wave per row,
// 16-byte vector loads, wave shuffles for the row reductions (no LDS on the forward path).
#include <map>
#include <mutex>

#include "common.h"
#include "vq3_hip.h"

namespace {

// ---------------------------------------------------------------- RMSNorm forward
// y = w * bf16(x * rsqrt(mean(x^2) + eps))   (modeling_qwen3.py:59-64: the cast to bf16 happens BEFORE the
// multiply by the weight)
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd,
                                                          long rows, int cols, long ldx, long ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  float ss = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf2f((bf16_t)v[j]);
      ss += f * f;
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)cols + eps);
  if (rstd && lane == 0) rstd[row] = rs;
  bf16_t* yr = y + row * ldy;
  for (int c = lane * 8; c < cols; c += 512) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + c);
    const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float n = rbf(bf2f((bf16_t)v[j]) * rs);
      o[j] = (short)f2bf(bf2f((bf16_t)wv[j]) * n);
    }
    *reinterpret_cast<bf16x8*>(yr + c) = o;
  }
}

// ---------------------------------------------------------------- RMSNorm backward
// 4 rows per block (one per wave; 300 blocks at M = 1200 fill the chip). A wave keeps its row's x / dy / w chunks in
// registers (one HBM read); the four waves' dw contributions meet in LDS and leave as ONE plain-store row of the
// partial slab dw_part[blockIdx][cols] - no global atomics (300 blocks hammering one 10-KB row ran ~14x below the
// atomic rate); vq3_colsum_f32_to_bf16 sums the slab straight into the bf16 gradient.
template <int NCH, int RB>  // cols <= NCH * 512; RB = rows per workgroup as a compile-time constant (0: the run-time argument)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16_t* dres, bf16_t* dx, float* __restrict__ dw,
                                                          long rows, int cols, int rb_rows) {
  extern __shared__ __attribute__((aligned(16))) float dw_s[];   // [4 waves][NCH * 512]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float dwacc[NCH][8];
  bf16x8 wv[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c = lane * 8 + ch * 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[ch][j] = 0.f;
    if (c < cols) wv[ch] = *reinterpret_cast<const bf16x8*>(w + c);
  }
  const int nrb = RB > 0 ? RB : rb_rows;              // (4, the default, as a constant: one pass, every load of the row at the top - as a
                                                      // run-time trip count the same kernel measured 73 us against 51)
  for (int rr = 0; rr < nrb / 4; ++rr) {              // nrb rows per workgroup (= per partial dw row), one row per wave and pass
    const long row = (long)blockIdx.x * nrb + rr * 4 + wid;
    if (row >= rows) break;
    const float rs = rstd[row];
    const bf16_t* xr = x + row * (long)cols;
    const bf16_t* gr = dy + row * (long)cols;
    bf16x8 xv[NCH], gv[NCH], rv[NCH];
    const bf16_t* rr_ = dres ? dres + row * (long)cols : nullptr;
    float dot = 0.f;
    // all three row operands are requested up front: the residual-gradient read otherwise waits behind the row reduction
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 8 + ch * 512;
      if (c < cols) {
        xv[ch] = *reinterpret_cast<const bf16x8*>(xr + c);
        gv[ch] = *reinterpret_cast<const bf16x8*>(gr + c);
        if (rr_) rv[ch] = *reinterpret_cast<const bf16x8*>(rr_ + c);
      }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 8 + ch * 512;
      if (c < cols) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = bf2f((bf16_t)xv[ch][j]) * rs;
          const float d = bf2f((bf16_t)gv[ch][j]);
          dot += d * bf2f((bf16_t)wv[ch][j]) * xh;
          dwacc[ch][j] += d * xh;
        }
      }
    }
    dot = wave_sum(dot) / (float)cols;
    bf16_t* dxr = dx + row * (long)cols;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = lane * 8 + ch * 512;
      if (c < cols) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = bf2f((bf16_t)xv[ch][j]) * rs;
          const float g = bf2f((bf16_t)gv[ch][j]) * bf2f((bf16_t)wv[ch][j]);
          float d = rs * (g - xh * dot);
          if (rr_) d += bf2f((bf16_t)rv[ch][j]);
          o[j] = (short)f2bf(d);
        }
        *reinterpret_cast<bf16x8*>(dxr + c) = o;
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c = lane * 8 + ch * 512;
    if (c < cols) {
      // LDS image [wave][chunk][half][lane][4]: a wave instruction writes 64 x 16 contiguous bytes (lane-major rows of 8 floats put
      // lanes l and l + 4 on the same banks: conflict share 0.75 in profiles/r3_pmc_mfma_lds.csv)
      float* base = dw_s + wid * (NCH * 512) + ch * 512 + lane * 4;
      *reinterpret_cast<f32x4*>(base) = f32x4{dwacc[ch][0], dwacc[ch][1], dwacc[ch][2], dwacc[ch][3]};
      *reinterpret_cast<f32x4*>(base + 256) = f32x4{dwacc[ch][4], dwacc[ch][5], dwacc[ch][6], dwacc[ch][7]};
    }
  }
  __syncthreads();
  float* out = dw + (long)blockIdx.x * cols;
  for (int i = threadIdx.x; i < NCH * 512; i += 256) {          // image index -> column: chunk i / 512, half (i / 256) & 1, lane (i & 255) / 4
    const int c = (i >> 9) * 512 + ((i & 255) >> 2) * 8 + ((i >> 8) & 1) * 4 + (i & 3);
    if (c < cols) out[c] = dw_s[i] + dw_s[NCH * 512 + i] + dw_s[2 * NCH * 512 + i] + dw_s[3 * NCH * 512 + i];
  }
}

// out_bf16[c] (+)= sum_r part[r][c]: 64 columns per block, 16 waves stride over the rows (4 independent loads in
// flight per lane), LDS combine.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ part, int nrows, int cols,
                                                      bf16_t* __restrict__ out, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 48 < nrows; r += 64) {
      a0 += part[(long)r * cols + c];
      a1 += part[(long)(r + 16) * cols + c];
      a2 += part[(long)(r + 32) * cols + c];
      a3 += part[(long)(r + 48) * cols + c];
    }
    for (; r < nrows; r += 16) a0 += part[(long)r * cols + c];
  }
  red[rg][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg == 0 && c < cols) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][lane];
    if (accumulate) v += bf2f(out[c]);
    out[c] = f2bf(v);
  }
}

// Several column sums in ONE launch (blockIdx.y = job): a decoder layer's backward leaves four partial slabs (two RMSNorm
// weights, q_norm, k_norm) - four 6 us launches and their boundaries become one.
// With merged micro-batches the slabs are tall (9600 partial rows of 128 columns from the q/k-prep backward: two workgroups walking
// 600 iterations each took 84 us): the rows are split over gridDim.z workgroups; each leaves its partial sums in a scratch slab
// (sc1 stores, drained before the ticket) and the last one to arrive adds them in split order - deterministic, one launch.
constexpr int CS_MAX_SPLIT = 32, CS_MAX_COLS = 4096;
struct ColsumJobs {
  const float* part[8];
  bf16_t* out[8];
  int nrows[8], cols[8], accumulate[8];
  float* scratch;     // [8 jobs][CS_MAX_SPLIT][CS_MAX_COLS] (gridDim.z > 1)
  int* tickets;       // [8 jobs][CS_MAX_COLS / 64], zero between launches
};
__global__ __launch_bounds__(1024) void colsum_multi_kernel(ColsumJobs jobs) {
  __shared__ float red[16][64];
  __shared__ int ticket_s;
  const int job = blockIdx.y;
  const float* __restrict__ part = jobs.part[job];
  const int cols = jobs.cols[job];
  if ((int)blockIdx.x * 64 >= cols) return;
  const int nsplit = gridDim.z, sp = blockIdx.z;
  const int chunk = ((jobs.nrows[job] + nsplit - 1) / nsplit + 15) / 16 * 16;
  const int r0 = sp * chunk;
  const int nrows = jobs.nrows[job] < r0 + chunk ? jobs.nrows[job] : r0 + chunk;       // this workgroup walks rows r0 .. nrows - 1
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int r = r0 + rg;
    for (; r + 48 < nrows; r += 64) {
      a0 += part[(long)r * cols + c];
      a1 += part[(long)(r + 16) * cols + c];
      a2 += part[(long)(r + 32) * cols + c];
      a3 += part[(long)(r + 48) * cols + c];
    }
    for (; r < nrows; r += 16) a0 += part[(long)r * cols + c];
  }
  red[rg][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float v = 0.f;
  if (rg == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][lane];
  }
  if (nsplit > 1) {
    float* sc = jobs.scratch + ((long)job * CS_MAX_SPLIT) * CS_MAX_COLS;
    if (rg == 0 && c < cols) __hip_atomic_store(sc + (long)sp * CS_MAX_COLS + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      int* tk = jobs.tickets + job * (CS_MAX_COLS / 64) + blockIdx.x;
      const int t = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == nsplit - 1) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket_s = t;
    }
    __syncthreads();
    if (ticket_s != nsplit - 1) return;
    v = 0.f;
    if (rg == 0 && c < cols)
      for (int s2 = 0; s2 < nsplit; ++s2) v += __hip_atomic_load(sc + (long)s2 * CS_MAX_COLS + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (rg == 0 && c < cols) {
    bf16_t* out = jobs.out[job];
    if (jobs.accumulate[job]) v += bf2f(out[c]);
    out[c] = f2bf(v);
  }
}

// ---------------------------------------------------------------- LayerNorm forward
template <bool X_F32, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* __restrict__ x_, const void* __restrict__ res_,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            bf16_t* y_bf16, float* y_f32, long rows, int cols,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  auto ld4 = [&](const void* base, int c, float (&o)[4]) {
    if (X_F32) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + row * (long)cols + c);
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    } else {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(base) + row * (long)cols + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bf2f((bf16_t)v[j]);
    }
  };
  auto ldrow = [&](int c, float (&o)[4]) {
    ld4(x_, c, o);
    if (res_) {
      float r[4];
      ld4(res_, c, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] += r[j];
        if (!X_F32) o[j] = rbf(o[j]);  // bf16 tensors: the sum is a bf16 tensor in PyTorch
      }
    }
  };
  // rows of up to 256 * NCH * ... elements are read ONCE and kept in registers (NCH chunks of 4 per lane); NCH == 0 is the
  // generic three-pass form for other widths
  float cache[NCH > 0 ? NCH : 1][4];
  float s = 0.f;
  if (NCH > 0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < cols) {
        ldrow(c, cache[i]);
        s += cache[i][0] + cache[i][1] + cache[i][2] + cache[i][3];
      }
    }
  } else {
    for (int c = lane * 4; c < cols; c += 256) {
      float v[4];
      ldrow(c, v);
      s += v[0] + v[1] + v[2] + v[3];
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
  if (NCH > 0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (lane * 4 + 256 * i < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = cache[i][j] - mean;
          q += d * d;
        }
      }
    }
  } else {
    for (int c = lane * 4; c < cols; c += 256) {
      float v[4];
      ldrow(c, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[j] - mean;
        q += d * d;
      }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
  auto emit = [&](int c, const float (&v)[4]) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rs * wv[j] + bv[j];
    if (y_f32) *reinterpret_cast<f32x4*>(y_f32 + row * (long)cols + c) = f32x4{o[0], o[1], o[2], o[3]};
    if (y_bf16) {
      u32x2 p;
      p[0] = pack2bf(o[0], o[1]);
      p[1] = pack2bf(o[2], o[3]);
      *reinterpret_cast<u32x2*>(y_bf16 + row * (long)cols + c) = p;
    }
  };
  if constexpr (NCH > 0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < cols) emit(c, cache[i]);
    }
  } else {
    for (int c = lane * 4; c < cols; c += 256) {
      float v[4];
      ldrow(c, v);
      emit(c, v);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm backward (Perceiver "train_projector" mode)
// y = xhat * w + b, xhat = (x - mean) * rstd over a row of `cols` f32 values (torch.nn.LayerNorm under autograd). One wave per row,
// the row in registers (NCH chunks of 4 values per lane); statistics recomputed from x (two-pass, as the forward kernel does):
//   g = dy * w;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dw_part[blk] = sum_rows dy * xhat;  db_part[blk] = sum_rows dy
// A workgroup (4 waves) walks LNB_ROWS rows and leaves ONE partial row of dw / db (plain stores, every slot written): reduce them
// with vq3_colsum_f32.
constexpr int LNB_ROWS = 16;
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res, const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ dw_part, float* __restrict__ db_part, long rows,
                                                            int cols, float eps) {
  extern __shared__ float lnb_red[];               // [2][cols]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float aw[NCH][4], ab[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { aw[c][j] = 0.f; ab[c][j] = 0.f; }
  const float inv = 1.f / (float)cols;
  for (int rr = wid; rr < LNB_ROWS; rr += 4) {
    const long row = (long)blockIdx.x * LNB_ROWS + rr;
    if (row >= rows) break;
    float xv[NCH][4], gv[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < cols) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + row * (long)cols + col);
        if (res) v += *reinterpret_cast<const f32x4*>(res + row * (long)cols + col);     // the forward normalised x + res
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[c][j] = v[j]; s += v[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[c][j] = 0.f;
      }
    }
    const float mean = wave_sum(s) * inv;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[c][j] -= mean; sq = fmaf(xv[c][j], xv[c][j], sq); }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < cols) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * (long)cols + col);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xv[c][j] *= rstd;                         // xhat
          gv[c][j] = d[j] * wv[j];
          sg += gv[c][j];
          sgx = fmaf(gv[c][j], xv[c][j], sgx);
          aw[c][j] = fmaf(d[j], xv[c][j], aw[c][j]);
          ab[c][j] += d[j];
        }
      }
    }
    const float mg = wave_sum(sg) * inv, mgx = wave_sum(sgx) * inv;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 4;
      if (col < cols) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (gv[c][j] - mg - xv[c][j] * mgx);
        *reinterpret_cast<f32x4*>(dx + row * (long)cols + col) = o;
      }
    }
  }
  // combine the four waves' partial column sums (the waves add into one [2][cols] LDS image in turn), one partial row per workgroup
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (wid == k) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 4;
        if (col < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lnb_red[col + j] = (k ? lnb_red[col + j] : 0.f) + aw[c][j];
            lnb_red[cols + col + j] = (k ? lnb_red[cols + col + j] : 0.f) + ab[c][j];
          }
        }
      }
    }
    __syncthreads();
  }
  for (int col = threadIdx.x; col < cols; col += 256) {
    dw_part[(long)blockIdx.x * cols + col] = lnb_red[col];
    db_part[(long)blockIdx.x * cols + col] = lnb_red[cols + col];
  }
}

// out_f32[c] (+)= sum_r part[r][c]  (the f32 twin of colsum_kernel: the Perceiver's parameters and gradients are fp32)
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* __restrict__ part, int nrows, int cols, float* __restrict__ out,
                                                          int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 48 < nrows; r += 64) {
      a0 += part[(long)r * cols + c];
      a1 += part[(long)(r + 16) * cols + c];
      a2 += part[(long)(r + 32) * cols + c];
      a3 += part[(long)(r + 48) * cols + c];
    }
    for (; r < nrows; r += 16) a0 += part[(long)r * cols + c];
  }
  red[rg][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg == 0 && c < cols) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += red[i][lane];
    if (accumulate) v += out[c];
    out[c] = v;
  }
}

}  // namespace

extern "C" int vq3_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int32_t cols,
                               int64_t ldx, int64_t ldy, float eps, void* stream) {
  VQ3_CHECK_ARG(x && w && y, "rmsnorm_fwd: null pointer");
  VQ3_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "rmsnorm_fwd: cols=%d must be a positive multiple of 8", cols);
  VQ3_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= cols && ldy >= cols, "rmsnorm_fwd: bad row strides");
  VQ3_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)w % 16 == 0),
                "rmsnorm_fwd: pointers must be 16-byte aligned");
  const long nblk = (rows + 3) / 4;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)w, (bf16_t*)y, rstd, (long)rows, cols, (long)ldx, (long)ldy, eps);
  VQ3_CHECK_LAUNCH("rmsnorm_fwd");
  return 0;
}

static int rmsnorm_bwd_impl(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                            float* dw_f32, int64_t rows, int32_t cols, int32_t rows_per_part, void* stream) {
  VQ3_CHECK_ARG(dy && x && w && rstd && dx && dw_f32, "rmsnorm_bwd: null pointer (dw_part must hold ceil(rows/rows_per_part)*cols floats)");
  VQ3_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 4096, "rmsnorm_bwd: bad cols=%d (<= 4096)", cols);
  VQ3_CHECK_ARG(rows_per_part >= 4 && rows_per_part <= 256 && rows_per_part % 4 == 0, "rmsnorm_bwd: rows_per_part=%d must be a multiple of 4 in 4..256", rows_per_part);
  const long nblk = (rows + rows_per_part - 1) / rows_per_part;
#define VQ3_RB_LAUNCH(NCH)                                                                                           \
  do {                                                                                                              \
    if (rows_per_part == 4)                                                                                         \
      hipLaunchKernelGGL((rmsnorm_bwd_kernel<NCH, 4>), dim3((unsigned)nblk), dim3(256), 4 * NCH * 512 * sizeof(float),     \
                         (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd,           \
                         (const bf16_t*)dres, (bf16_t*)dx, dw_f32, (long)rows, cols, 4);                            \
    else                                                                                                            \
      hipLaunchKernelGGL((rmsnorm_bwd_kernel<NCH, 0>), dim3((unsigned)nblk), dim3(256), 4 * NCH * 512 * sizeof(float),     \
                         (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, rstd,           \
                         (const bf16_t*)dres, (bf16_t*)dx, dw_f32, (long)rows, cols, (int)rows_per_part);           \
  } while (0)
  if (cols <= 512) VQ3_RB_LAUNCH(1);
  else if (cols <= 1024) VQ3_RB_LAUNCH(2);
  else if (cols <= 2560) VQ3_RB_LAUNCH(5);
  else VQ3_RB_LAUNCH(8);
#undef VQ3_RB_LAUNCH
  VQ3_CHECK_LAUNCH("rmsnorm_bwd");
  return 0;
}

extern "C" int vq3_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                               float* dw_f32, int64_t rows, int32_t cols, float eps, void* stream) {
  (void)eps;
  return rmsnorm_bwd_impl(dy, x, w, rstd, dres, dx, dw_f32, rows, cols, 4, stream);
}

// The same with rows_per_part (a multiple of 4) rows per workgroup: dw_part is [ceil(rows / rows_per_part), cols]. At 9 600 rows (a
// merged pass) 4 rows per partial make the slab a tenth of the kernel's traffic and its column sum a launch of half the kernel's length.
extern "C" int vq3_rmsnorm_bwd_rows(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                                    float* dw_f32, int64_t rows, int32_t cols, int32_t rows_per_part, void* stream) {
  return rmsnorm_bwd_impl(dy, x, w, rstd, dres, dx, dw_f32, rows, cols, rows_per_part, stream);
}

namespace {
// (sum, sum of squares) per row and 128-column group of a bf16 matrix: the input of a LayerNorm folded into the GEMM that consumes it
// (gemm_common.h: ln_row), for rows that no GEMM epilogue has produced statistics for. One thread per 8 columns, 16 lanes per group.
__global__ __launch_bounds__(256) void rowstats128_kernel(const bf16_t* __restrict__ x, float* __restrict__ st, long rows, int cols) {
  const int tpr = cols >> 3;                                   // threads per row (multiple of 16)
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = gid / tpr;
  const int c = (int)(gid - row * tpr);
  float sm = 0.f, sq = 0.f;
  if (row < rows) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * cols + c * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float lo = bf2f((bf16_t)(v[k] & 0xffff)), hi = bf2f((bf16_t)(v[k] >> 16));
      sm += lo + hi;
      sq = fmaf(lo, lo, fmaf(hi, hi, sq));
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }
  if (row < rows && (c & 15) == 0) *reinterpret_cast<float2*>(st + (row * (cols >> 7) + (c >> 4)) * 2) = float2{sm, sq};
}
}  // namespace

extern "C" int vq3_rowstats128(const void* x, float* stats, int64_t rows, int32_t cols, void* stream) {
  VQ3_CHECK_ARG(x && stats && rows > 0 && cols > 0 && cols % 128 == 0, "rowstats128: bad arguments (cols must be a multiple of 128)");
  VQ3_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)stats % 8 == 0, "rowstats128: x must be 16-byte aligned");
  const long threads = rows * (cols / 8);
  hipLaunchKernelGGL(rowstats128_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, stats,
                     (long)rows, cols);
  VQ3_CHECK_LAUNCH("rowstats128");
  return 0;
}

extern "C" int vq3_layernorm_fwd(const void* x, const void* res, int32_t x_f32, const float* w, const float* b,
                                 void* y_bf16, float* y_f32, int64_t rows, int32_t cols, float eps, void* stream) {
  VQ3_CHECK_ARG(x && w && b && (y_bf16 || y_f32), "layernorm_fwd: null pointer");
  VQ3_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "layernorm_fwd: cols=%d must be a positive multiple of 4", cols);
  const long nblk = (rows + 3) / 4;
  const int nch = cols <= 1024 ? 4 : (cols <= 2048 ? 8 : (cols <= 4096 ? 16 : 0));
#define VQ3_LN(F32, N)                                                                                                   \
  hipLaunchKernelGGL((layernorm_fwd_kernel<F32, N>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, res, w, b, \
                     (bf16_t*)y_bf16, y_f32, (long)rows, cols, eps)
  if (x_f32) {
    if (nch == 4) VQ3_LN(true, 4); else if (nch == 8) VQ3_LN(true, 8); else if (nch == 16) VQ3_LN(true, 16); else VQ3_LN(true, 0);
  } else {
    if (nch == 4) VQ3_LN(false, 4); else if (nch == 8) VQ3_LN(false, 8); else if (nch == 16) VQ3_LN(false, 16); else VQ3_LN(false, 0);
  }
#undef VQ3_LN
  VQ3_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

namespace {
struct CsWs { float* scratch = nullptr; int* tickets = nullptr; };
std::mutex g_cs_mutex;
std::map<hipStream_t, CsWs> g_cs;     // one scratch per stream: launches of a stream run in order
bool colsum_scratch(hipStream_t s, float** scratch, int** tickets) {
  std::lock_guard<std::mutex> lock(g_cs_mutex);
  CsWs& w = g_cs[s];
  if (!w.scratch) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
    const size_t nt = 8 * (CS_MAX_COLS / 64);
    if (hipMalloc(&w.scratch, (size_t)8 * CS_MAX_SPLIT * CS_MAX_COLS * sizeof(float)) != hipSuccess ||
        hipMalloc(&w.tickets, nt * sizeof(int)) != hipSuccess) {
      (void)hipGetLastError();
      w.scratch = nullptr;
      return false;
    }
  }
  // The arrival tickets are zeroed on the CALLER's stream before every launch that uses them (a memset node when captured): ordered
  // with the kernel by the stream itself, and a launch that died half-way cannot poison the next one (cdna guide, Guideline 16
  // "Re-initialise every call"; the block starts at its allocation's start and is a multiple of 16 bytes).
  static_assert((8 * (CS_MAX_COLS / 64) * sizeof(int)) % 16 == 0, "ticket block: multiple of 16 bytes");
  if (hipMemsetAsync(w.tickets, 0, 8 * (CS_MAX_COLS / 64) * sizeof(int), s) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  *scratch = w.scratch; *tickets = w.tickets;
  return true;
}
}  // namespace

extern "C" int vq3_colsum_multi(const vq3_colsum_job* jobs, int32_t njobs, void* stream) {
  VQ3_CHECK_ARG(jobs && njobs >= 1 && njobs <= 8, "colsum_multi: 1..8 jobs");
  ColsumJobs j;
  int maxcols = 0, maxrows = 0;
  for (int i = 0; i < njobs; ++i) {
    VQ3_CHECK_ARG(jobs[i].part && jobs[i].out_bf16 && jobs[i].nrows > 0 && jobs[i].cols > 0, "colsum_multi: bad job %d", i);
    j.part[i] = jobs[i].part; j.out[i] = (bf16_t*)jobs[i].out_bf16;
    j.nrows[i] = jobs[i].nrows; j.cols[i] = jobs[i].cols; j.accumulate[i] = jobs[i].accumulate;
    maxcols = jobs[i].cols > maxcols ? jobs[i].cols : maxcols;
    maxrows = jobs[i].nrows > maxrows ? jobs[i].nrows : maxrows;
  }
  int nsplit = maxrows / 512;                            // ~512 partial rows per workgroup
  nsplit = nsplit < 1 ? 1 : (nsplit > CS_MAX_SPLIT ? CS_MAX_SPLIT : nsplit);
  j.scratch = nullptr; j.tickets = nullptr;
  if (nsplit > 1 && (maxcols > CS_MAX_COLS || !colsum_scratch((hipStream_t)stream, &j.scratch, &j.tickets))) nsplit = 1;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3((maxcols + 63) / 64, njobs, nsplit), dim3(1024), 0, (hipStream_t)stream, j);
  VQ3_CHECK_LAUNCH("colsum_multi");
  return 0;
}

extern "C" int vq3_layernorm_bwd(const float* dy, const float* x, const float* res, const float* w, float* dx, float* dw_part,
                                 float* db_part, int64_t rows, int32_t cols, float eps, void* stream) {
  VQ3_CHECK_ARG(dy && x && w && dx && dw_part && db_part, "layernorm_bwd: null pointer (dw_part / db_part: ceil(rows / 16) * cols floats each)");
  VQ3_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 4096, "layernorm_bwd: cols=%d must be a multiple of 4, <= 4096", cols);
  VQ3_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)res | (uintptr_t)w | (uintptr_t)dx) % 16) == 0, "layernorm_bwd: pointers must be 16-byte aligned");
  const long nblk = (rows + LNB_ROWS - 1) / LNB_ROWS;
  const size_t smem = (size_t)2 * cols * sizeof(float);
#define VQ3_LNB(N)                                                                                                          \
  hipLaunchKernelGGL((layernorm_bwd_kernel<N>), dim3((unsigned)nblk), dim3(256), smem, (hipStream_t)stream, dy, x, res, w, dx, dw_part, \
                     db_part, (long)rows, cols, eps)
  if (cols <= 1024) VQ3_LNB(4); else if (cols <= 2048) VQ3_LNB(8); else VQ3_LNB(16);
#undef VQ3_LNB
  VQ3_CHECK_LAUNCH("layernorm_bwd");
  return 0;
}

extern "C" int vq3_colsum_f32(const float* part, int32_t nrows, int32_t cols, float* out_f32, int32_t accumulate, void* stream) {
  VQ3_CHECK_ARG(part && out_f32 && nrows > 0 && cols > 0, "colsum_f32: bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, nrows, cols, out_f32, accumulate);
  VQ3_CHECK_LAUNCH("colsum_f32");
  return 0;
}

extern "C" int vq3_colsum_f32_to_bf16(const float* part, int32_t nrows, int32_t cols, void* out_bf16, int32_t accumulate,
                                      void* stream) {
  VQ3_CHECK_ARG(part && out_bf16 && nrows > 0 && cols > 0, "colsum: bad args");
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, nrows, cols,
                     (bf16_t*)out_bf16, accumulate);
  VQ3_CHECK_LAUNCH("colsum_f32_to_bf16");
  return 0;
}
