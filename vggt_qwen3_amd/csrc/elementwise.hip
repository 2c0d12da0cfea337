// HBM-bound element-wise / data-movement kernels: SwiGLU gate, transposes, casts, row gather/scatter,
// embedding + visual-span splice, masked softmax, cross entropy, fused AdamW. All 16-byte vectorised where the
// layout allows; grid-stride over at most 2048 blocks for the flat ones (256 CUs x 8).
#include "common.h"
#include "vq3_hip.h"

namespace {

inline unsigned flat_grid(long n_items, int per_block) {
  long b = (n_items + per_block - 1) / per_block;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------- SwiGLU
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                           long rows, int inter) {
  const int cpr = inter / 8;  // 16B chunks per row
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + r * 2L * inter + c);
    const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + r * 2L * inter + inter + c);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // act_fn(gate) is a bf16 tensor, then * up -> bf16 (modeling_qwen3.py:82)
      const float s = rbf(silu_f(bf2f((bf16_t)g[j])));
      o[j] = (short)f2bf(s * bf2f((bf16_t)u[j]));
    }
    *reinterpret_cast<bf16x8*>(act + r * (long)inter + c) = o;
  }
}

__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const bf16_t* __restrict__ dact,
                                                           const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                                           long rows, int inter) {
  const int cpr = inter / 8;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gu + r * 2L * inter + c);
    const bf16x8 u = *reinterpret_cast<const bf16x8*>(gu + r * 2L * inter + inter + c);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dact + r * (long)inter + c);
    bf16x8 og, ou;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gf = bf2f((bf16_t)g[j]), uf = bf2f((bf16_t)u[j]), df = bf2f((bf16_t)d[j]);
      const float sg = sigmoid_f(gf);
      const float sl = gf * sg;
      og[j] = (short)f2bf(df * uf * (sg * (1.f + gf * (1.f - sg))));
      ou[j] = (short)f2bf(df * sl);
    }
    *reinterpret_cast<bf16x8*>(dgu + r * 2L * inter + c) = og;
    *reinterpret_cast<bf16x8*>(dgu + r * 2L * inter + inter + c) = ou;
  }
}

// ---------------------------------------------------------------- GELU (erf form) forward on a materialised pre-activation
// h = bf16(gelu(z)): what the GEMM epilogue (act = 1) applies to its rounded output - the trained-projector path keeps z for the backward
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ h, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 zz = *reinterpret_cast<const bf16x8*>(z + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2_t g = gelu_erf2_b(f32x2_t{bf2f((bf16_t)zz[2 * j]), bf2f((bf16_t)zz[2 * j + 1])});
      o[2 * j] = (short)f2bf(g[0]); o[2 * j + 1] = (short)f2bf(g[1]);
    }
    *reinterpret_cast<bf16x8*>(h + i * 8) = o;
  }
}

// ---------------------------------------------------------------- GELU (erf form) backward: dz = dh * (Phi(z) + z * phi(z))
// z = the bf16 pre-activation the forward GEMM rounded before applying GELU (projector_perceiver.py:35-37 under autograd).
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ z, bf16_t* __restrict__ dz,
                                                       long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dh + i * 8);
    const bf16x8 zz = *reinterpret_cast<const bf16x8*>(z + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float zf = bf2f((bf16_t)zz[j]), df = bf2f((bf16_t)d[j]);
      const float cdf = 0.5f * (1.f + erff(zf * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * __expf(-0.5f * zf * zf);
      o[j] = (short)f2bf(df * (cdf + zf * pdf));
    }
    *reinterpret_cast<bf16x8*>(dz + i * 8) = o;
  }
}

// ---------------------------------------------------------------- transpose (64x64 tiles through LDS)
// 16-byte global loads and stores on both sides when the leading dimensions allow (vec != 0), scalar at the edges.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                                        int C, int Rpad, long lds_, long ldd, int n1, int n2, long s0,
                                                        long s1, long s2, long d0, long d1, long d2, int vec) {
  // row stride 66 elements = 33 dwords (odd): the store phase reads element (ch * 8 + j, cc) with ch = lane & 7, i.e. rows 8 apart - at the
  // old stride of 72 elements (36 dwords, 8 rows = 288 dwords = 0 mod 32) all eight landed on ONE bank (8-way; LDS-conflict share 0.82 in
  // profiles/r3_pmc_mfma_lds.csv); with the column swizzle of rows >= 32 (below) both phases are conflict-free
  __shared__ bf16_t tile[64][66];
  const int bz = blockIdx.z;
  const int i2 = bz % n2, i1 = (bz / n2) % n1, i0 = bz / (n2 * n1);
  src += i0 * s0 + i1 * s1 + i2 * s2;
  dst += i0 * d0 + i1 * d1 + i2 * d2;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  // load: 64 rows x 8 chunks of 8 elements; thread -> (row = id>>3, chunk = id&7), two passes
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int id = threadIdx.x + pass * 256;
    const int rr = id >> 3, ch = id & 7;
    const int r = r0 + rr, c = c0 + ch * 8;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < R) {
      if (vec && c + 8 <= C) {
        v = *reinterpret_cast<const bf16x8*>(src + (long)r * lds_ + c);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c + j < C) v[j] = (short)src[(long)r * lds_ + c + j];
      }
    }
    {
      const u32x4 u = __builtin_bit_cast(u32x4, v);
      unsigned* tp = reinterpret_cast<unsigned*>(&tile[rr][ch * 8]);       // (4-byte aligned: 132 rr + 16 ch)
      const int sw = (rr >> 4) & 2;            // rows 32-63 keep the two 4-column halves of a chunk swapped (below)
      tp[0 ^ sw] = u[0]; tp[1 ^ sw] = u[1]; tp[2 ^ sw] = u[2]; tp[3 ^ sw] = u[3];
    }
  }
  __syncthreads();
  // store: dst row = c (64 of them), 8 chunks of 8 consecutive r
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int id = threadIdx.x + pass * 256;
    const int cc = id >> 3, ch = id & 7;
    const int c = c0 + cc, r = r0 + ch * 8;
    if (c >= C || r >= Rpad) continue;
    bf16x8 v;
#pragma unroll
    // bank of element (row, col) = (row + col / 2) mod 32 at the 33-dword row stride: the eight ch of a half-wave read rows 8 ch + j, i.e.
    // banks 8 ch + j + cc / 2 - ch and ch + 4 would meet; columns of rows >= 32 are stored with bit 2 flipped, which moves them two banks on
    for (int j = 0; j < 8; ++j) v[j] = (short)tile[ch * 8 + j][cc ^ ((ch >> 2) << 2)];
    if (vec && r + 8 <= Rpad) {
      *reinterpret_cast<bf16x8*>(dst + (long)c * ldd + r) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r + j < Rpad) dst[(long)c * ldd + r + j] = (bf16_t)v[j];
    }
  }
}

// ---------------------------------------------------------------- casts
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    u32x2 o;
    o[0] = pack2bf(v[0], v[1]);
    o[1] = pack2bf(v[2], v[3]);
    *reinterpret_cast<u32x2*>(y + i * 4) = o;
  }
  if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) y[i] = f2bf(x[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) =
        f32x4{bf2f((bf16_t)v[0]), bf2f((bf16_t)v[1]), bf2f((bf16_t)v[2]), bf2f((bf16_t)v[3])};
  }
  if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) y[i] = bf2f(x[i]);
}
__global__ __launch_bounds__(256) void f32_to_bf16_acc_kernel(const float* __restrict__ src, bf16_t* __restrict__ acc,
                                                              long n, int accumulate) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = src[i];
    if (accumulate) v += bf2f(acc[i]);
    acc[i] = f2bf(v);
  }
}

// ---------------------------------------------------------------- row gather / scatter (bf16 rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                          bf16_t* __restrict__ out, int n, int cols) {
  const int i = blockIdx.x;
  bf16_t* o = out + (long)i * cols;
  if (i < n) {
    const bf16_t* s = src + (long)idx[i] * cols;
    for (int c = threadIdx.x * 8; c < cols; c += 2048) *reinterpret_cast<bf16x8*>(o + c) = *reinterpret_cast<const bf16x8*>(s + c);
  } else {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x * 8; c < cols; c += 2048) *reinterpret_cast<bf16x8*>(o + c) = z;
  }
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                           bf16_t* __restrict__ dst, int cols, int accumulate) {
  const int i = blockIdx.x;
  const bf16_t* s = src + (long)i * cols;
  bf16_t* d = dst + (long)idx[i] * cols;
  for (int c = threadIdx.x * 8; c < cols; c += 2048) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(s + c);
    if (accumulate) {
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(d + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)v[j]) + bf2f((bf16_t)o[j]));
    }
    *reinterpret_cast<bf16x8*>(d + c) = v;
  }
}

// ---------------------------------------------------------------- embedding + splice
// srcmap[b*L+l] < 0: row comes from table[ids]; otherwise from feat[b, srcmap, :] (vggt_qwen3_vlm.py:190-195).
__global__ __launch_bounds__(256) void embed_splice_fwd_kernel(const int64_t* __restrict__ ids,
                                                               const bf16_t* __restrict__ table,
                                                               const bf16_t* __restrict__ feat,
                                                               const int32_t* __restrict__ srcmap,
                                                               bf16_t* __restrict__ out, int L, int H, int S) {
  const long t = blockIdx.x;  // b*L + l
  const int b = (int)(t / L);
  const int sm = srcmap[t];
  const bf16_t* s = sm < 0 ? table + ids[t] * (long)H : feat + ((long)b * S + sm) * H;
  bf16_t* o = out + t * (long)H;
  for (int c = threadIdx.x * 8; c < H; c += 2048) *reinterpret_cast<bf16x8*>(o + c) = *reinterpret_cast<const bf16x8*>(s + c);
}

// Backward into the (tied) embedding table gradient. `order` is the argsort of ids over all B*L positions and
// sorted_ids the sorted values: the blocks whose position starts a run of equal ids sum the run and add it to the
// bf16 gradient row once (no atomics, deterministic). Positions that were overwritten by the splice are skipped.
// grid (T, ceil(H / 512)): a block owns 512 columns of one run; its four waves take the run's rows in interleaved
// batches of 16 (all 16 row loads in flight; the padding id's run is ~1000 rows long) and meet in LDS.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ sorted_ids,
                                                        const int64_t* __restrict__ order,
                                                        const int32_t* __restrict__ srcmap,
                                                        const bf16_t* __restrict__ dout, bf16_t* __restrict__ dtable,
                                                        int T, int H) {
  __shared__ float part[4][8][64];      // [wave][j][lane]: consecutive lanes on consecutive banks (lane-major rows of 8 were 8-way)
  __shared__ int any_s[4];
  const int i = blockIdx.x;
  const int64_t id = sorted_ids[i];
  if (i > 0 && sorted_ids[i - 1] == id) return;
  int lo = i + 1, hi = T;                       // first index > i whose id differs (sorted: binary search)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted_ids[mid] == id) lo = mid + 1; else hi = mid;
  }
  const int end = lo;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.y * 512 + lane * 8;
  const bool cin = c < H;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool any = false;
  for (int k0 = i + wid * 16; k0 < end; k0 += 64) {
    long t[16];
    bool use[16];
    bf16x8 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = order[k0 + u < end ? k0 + u : end - 1];
#pragma unroll
    for (int u = 0; u < 16; ++u) use[u] = (k0 + u < end) && srcmap[t[u]] < 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (cin && use[u]) ? *reinterpret_cast<const bf16x8*>(dout + t[u] * (long)H + c) : bf16x8{};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (use[u]) {
        any = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += bf2f((bf16_t)v[u][j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[wid][j][lane] = a[j];
  if (lane == 0) any_s[wid] = any ? 1 : 0;      // `any` is wave-uniform (it depends on the rows, not on the lane)
  __syncthreads();
  if (wid != 0 || !cin || !(any_s[0] | any_s[1] | any_s[2] | any_s[3])) return;
  bf16_t* d = dtable + id * (long)H + c;
  const bf16x8 o = *reinterpret_cast<const bf16x8*>(d);
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    r[j] = (short)f2bf(((part[0][j][lane] + part[1][j][lane]) + (part[2][j][lane] + part[3][j][lane])) + bf2f((bf16_t)o[j]));
  *reinterpret_cast<bf16x8*>(d) = r;
}
// dfeat_f32[b, s, :] += dout[b, l, :] where srcmap[b,l] == s
__global__ __launch_bounds__(256) void splice_bwd_kernel(const int32_t* __restrict__ srcmap,
                                                         const bf16_t* __restrict__ dout, float* __restrict__ dfeat,
                                                         int L, int H, int S) {
  const long t = blockIdx.x;
  const int sm = srcmap[t];
  if (sm < 0) return;
  const int b = (int)(t / L);
  float* d = dfeat + ((long)b * S + sm) * H;
  for (int c = threadIdx.x; c < H; c += 256) atomicAdd(&d[c], bf2f(dout[t * (long)H + c]));
}

// ---------------------------------------------------------------- masked softmax (one wave per row)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, bf16_t* __restrict__ P,
                                                          const uint8_t* __restrict__ keymask, int heads_per_mask,
                                                          int Lq, int Lk, int ldS, int ldP, int causal, long nrows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int nbi = (int)(row / Lq), i = (int)(row - (long)nbi * Lq);
  const float* s = S + row * (long)ldS;
  bf16_t* p = P + row * (long)ldP;
  const uint8_t* km = keymask ? keymask + (long)(nbi / heads_per_mask) * Lk : nullptr;
  const int lim = causal ? (i + 1 < Lk ? i + 1 : Lk) : Lk;
  float mx = -INFINITY;
  for (int j = lane; j < lim; j += 64)
    if (!km || km[j]) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < lim; j += 64)
    if (!km || km[j]) sum += __expf(s[j] - mx);
  sum = wave_sum(sum);
  const float inv = (sum > 0.f) ? 1.f / sum : 0.f;
  for (int j = lane; j < ldP; j += 64) {
    float v = 0.f;
    if (j < lim && (!km || km[j])) v = __expf(s[j] - mx) * inv;
    p[j] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP,
                                                          bf16_t* __restrict__ dS, int Lk, int ldS, int ldP, float scale,
                                                          long nrows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const bf16_t* p = P + row * (long)ldP;
  const float* dp = dP + row * (long)ldS;
  bf16_t* ds = dS + row * (long)ldP;
  float t = 0.f;
  for (int j = lane; j < Lk; j += 64) t += bf2f(p[j]) * dp[j];
  t = wave_sum(t);
  for (int j = lane; j < ldP; j += 64) {
    float v = 0.f;
    if (j < Lk) v = scale * bf2f(p[j]) * (dp[j] - t);
    ds[j] = f2bf(v);
  }
}

// ---------------------------------------------------------------- cross entropy (one block per row)
// 1024 threads, 16-byte accesses: a 152k-entry row is 19 vector loads per thread and pass (three passes: max, sum of
// exponentials, gradient written in place). Rows start 16-byte aligned (ldl % 8 == 0, checked on the host).
// row_scale / row_loss (both optional): a per-row gradient scale instead of the uniform one, and the row's loss written out instead
// of added to loss_sum - several micro-batches share one launch, each normalised by its own number of labelled rows.
__global__ __launch_bounds__(1024) void cross_entropy_kernel(bf16_t* __restrict__ logits, const int32_t* __restrict__ tgt,
                                                             float* __restrict__ loss_sum, int V, int ldl, float gscale,
                                                             const float* __restrict__ row_scale, float* __restrict__ row_loss) {
  __shared__ float red[16];
  bf16_t* lr = logits + (long)blockIdx.x * ldl;
  const int t = tgt[blockIdx.x];
  const int nv = ldl / 8;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < nv; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c * 8 + j < V) mx = fmaxf(mx, bf2f((bf16_t)v[j]));
  }
  mx = block_max<16>(mx, red);
  float sum = 0.f;
  for (int c = threadIdx.x; c < nv; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c * 8 + j < V) sum += __expf(bf2f((bf16_t)v[j]) - mx);
  }
  sum = block_sum<16>(sum, red);
  const float lse = mx + __logf(sum);
  if (row_scale) gscale = row_scale[blockIdx.x];
  if (threadIdx.x == 0) {
    if (row_loss) row_loss[blockIdx.x] = lse - bf2f(lr[t]);
    else atomicAdd(loss_sum, lse - bf2f(lr[t]));
  }
  __syncthreads();  // the target logit has been read before any thread rewrites the row
  for (int c = threadIdx.x; c < nv; c += 1024) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(lr + c * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = c * 8 + j;
      float g = 0.f;
      if (col < V) g = (__expf(bf2f((bf16_t)v[j]) - lse) - (col == t ? 1.f : 0.f)) * gscale;
      o[j] = (short)f2bf(g);
    }
    *reinterpret_cast<bf16x8*>(lr + c * 8) = o;
  }
}

// ---------------------------------------------------------------- AdamW (f32 master, bf16 compute copy)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, const bf16_t* __restrict__ grad,
                                                    bf16_t* __restrict__ w, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                                    const float* __restrict__ clip_sumsq, float max_norm) {
  if (clip_sumsq) {   // global-norm clipping: coefficient from the device-side squared norm (same value in every thread)
    const float total = sqrtf(*clip_sumsq) * gscale;
    gscale *= fminf(1.f, max_norm / (total + 1e-6f));
  }
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 p = *reinterpret_cast<const f32x4*>(master + i * 4);
    f32x4 mm = *reinterpret_cast<const f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + i * 4);
    const bf16x4 gg = *reinterpret_cast<const bf16x4*>(grad + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = bf2f((bf16_t)gg[j]) * gscale;
      mm[j] = b1 * mm[j] + (1.f - b1) * g;
      vv[j] = b2 * vv[j] + (1.f - b2) * g * g;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      p[j] = p[j] * (1.f - lr * wd) - (lr / bc1) * (mm[j] / denom);
    }
    *reinterpret_cast<f32x4*>(master + i * 4) = p;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
    u32x2 o;
    o[0] = pack2bf(p[0], p[1]);
    o[1] = pack2bf(p[2], p[3]);
    *reinterpret_cast<u32x2*>(w + i * 4) = o;
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float g = bf2f(grad[i]) * gscale;
      const float mi = b1 * m[i] + (1.f - b1) * g;
      const float vi = b2 * v[i] + (1.f - b2) * g * g;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      const float p = master[i] * (1.f - lr * wd) - (lr / bc1) * (mi / denom);
      master[i] = p; m[i] = mi; v[i] = vi; w[i] = f2bf(p);
    }
  }
}

// inverted dropout in place: element i is kept with probability 1 - p (and scaled by 1 / (1 - p)), decided by a counter-based
// hash of (seed, offset + i) - stateless, so a graph replay or a re-run with the same (seed, offset) repeats the mask
// (drop_bits: common.h - the fused Perceiver cross-attention regenerates the same mask)
// 16 bytes per lane (8 bf16 / 4 f32 elements, one decision each - the mask of element i does not depend on how the kernel walks the tensor):
// with one 2-byte element per lane a wave instruction moved 128 B and the [6144, 16384] bf16 activation of the Perceiver's MLP took 180 us
template <bool F32IO>
__global__ __launch_bounds__(256) void dropout_kernel(void* __restrict__ xv, long n, unsigned thresh, float scale,
                                                      unsigned long long seed, unsigned long long offset) {
  constexpr int V = F32IO ? 4 : 8;
  const bool vec = (reinterpret_cast<uintptr_t>(xv) & 15) == 0;
  const long nv = vec ? n / V : 0;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nv; c += (long)gridDim.x * 256) {
    const long i0 = c * V;
    u32x4 w = reinterpret_cast<u32x4*>(xv)[c];
    if (F32IO) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool keep = drop_bits(seed, offset + (unsigned long long)(i0 + k)) >= thresh;
        w[k] = keep ? __float_as_uint(__uint_as_float(w[k]) * scale) : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool k0 = drop_bits(seed, offset + (unsigned long long)(i0 + 2 * k)) >= thresh;
        const bool k1 = drop_bits(seed, offset + (unsigned long long)(i0 + 2 * k + 1)) >= thresh;
        const bf16_t lo = k0 ? f2bf(bf2f((bf16_t)(w[k] & 0xffff)) * scale) : (bf16_t)0;
        const bf16_t hi = k1 ? f2bf(bf2f((bf16_t)(w[k] >> 16)) * scale) : (bf16_t)0;
        w[k] = (uint32_t)lo | ((uint32_t)hi << 16);
      }
    }
    reinterpret_cast<u32x4*>(xv)[c] = w;
  }
  for (long i = nv * V + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {      // tail / unaligned tensors
    const bool keep = drop_bits(seed, offset + (unsigned long long)i) >= thresh;
    if (F32IO) {
      float* x = reinterpret_cast<float*>(xv);
      x[i] = keep ? x[i] * scale : 0.f;
    } else {
      bf16_t* x = reinterpret_cast<bf16_t*>(xv);
      x[i] = keep ? f2bf(bf2f(x[i]) * scale) : (bf16_t)0;
    }
  }
}

// sum of squares, stage 1: one partial per block (fixed grid => fixed summation order)
template <bool F32IN>
__global__ __launch_bounds__(256) void sumsq_kernel(const void* __restrict__ xv, long n, float* __restrict__ partials) {
  __shared__ float red[4];
  float acc = 0.f;
  if (F32IN) {
    const float* x = reinterpret_cast<const float*>(xv);
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
      acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
      for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
  } else {
    const bf16_t* x = reinterpret_cast<const bf16_t*>(xv);
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = bf2f((bf16_t)v[j]); acc += f * f; }
    }
    if (blockIdx.x == 0)
      for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) { const float f = bf2f(x[i]); acc += f * f; }
  }
  const float t = block_sum<4>(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// stage 2: one block adds the partials in a fixed tree order
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partials, int nb, float* __restrict__ accum) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partials[i];
  const float t = block_sum<4>(acc, red);
  if (threadIdx.x == 0) accum[0] += t;
}

}  // namespace

// ==================================================================== C ABI
extern "C" int vq3_silu_mul_fwd(const void* gu, void* act, int64_t rows, int32_t inter, void* stream) {
  VQ3_CHECK_ARG(gu && act && rows > 0 && inter > 0 && inter % 8 == 0, "silu_mul_fwd: bad args (inter %% 8)");
  hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(flat_grid(rows * (inter / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gu, (bf16_t*)act, (long)rows, inter);
  VQ3_CHECK_LAUNCH("silu_mul_fwd");
  return 0;
}
extern "C" int vq3_silu_mul_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int32_t inter, void* stream) {
  VQ3_CHECK_ARG(dact && gu && dgu && rows > 0 && inter > 0 && inter % 8 == 0, "silu_mul_bwd: bad args");
  hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(flat_grid(rows * (inter / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dact, (const bf16_t*)gu, (bf16_t*)dgu, (long)rows, inter);
  VQ3_CHECK_LAUNCH("silu_mul_bwd");
  return 0;
}

extern "C" int vq3_gelu_fwd(const void* z, void* h, int64_t n, void* stream) {
  VQ3_CHECK_ARG(z && h && n > 0 && n % 8 == 0, "gelu_fwd: n must be a positive multiple of 8");
  VQ3_CHECK_ARG((((uintptr_t)z | (uintptr_t)h) % 16) == 0, "gelu_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(flat_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (bf16_t*)h, (long)(n / 8));
  VQ3_CHECK_LAUNCH("gelu_fwd");
  return 0;
}

extern "C" int vq3_gelu_bwd(const void* dh, const void* z, void* dz, int64_t n, void* stream) {
  VQ3_CHECK_ARG(dh && z && dz && n > 0 && n % 8 == 0, "gelu_bwd: n must be a positive multiple of 8");
  VQ3_CHECK_ARG((((uintptr_t)dh | (uintptr_t)z | (uintptr_t)dz) % 16) == 0, "gelu_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(flat_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, (const bf16_t*)z,
                     (bf16_t*)dz, (long)(n / 8));
  VQ3_CHECK_LAUNCH("gelu_bwd");
  return 0;
}

extern "C" int vq3_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int32_t Rpad, int64_t lds,
                                  int64_t ldd, int32_t n0, int32_t n1, int32_t n2, int64_t s0, int64_t s1, int64_t s2,
                                  int64_t d0, int64_t d1, int64_t d2, void* stream) {
  VQ3_CHECK_ARG(src && dst && R > 0 && C > 0 && Rpad >= R, "transpose: bad shape");
  VQ3_CHECK_ARG(lds >= C && ldd >= Rpad, "transpose: leading dims too small");
  VQ3_CHECK_ARG(n0 >= 1 && n1 >= 1 && n2 >= 1 && (long)n0 * n1 * n2 <= 65535, "transpose: bad batch dims");
  dim3 grid((Rpad + 63) / 64, (C + 63) / 64, n0 * n1 * n2);
  const bool al = (lds % 8 == 0) && (ldd % 8 == 0) && (s0 % 8 == 0) && (s1 % 8 == 0) && (s2 % 8 == 0) && (d0 % 8 == 0) &&
                  (d1 % 8 == 0) && (d2 % 8 == 0) && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, R, C,
                     Rpad, (long)lds, (long)ldd, n1, n2, (long)s0, (long)s1, (long)s2, (long)d0, (long)d1, (long)d2,
                     al ? 1 : 0);
  VQ3_CHECK_LAUNCH("transpose_bf16");
  return 0;
}

extern "C" int vq3_cast(const void* x, void* y, int64_t n, int32_t dir, void* stream) {
  VQ3_CHECK_ARG(x && y && n > 0 && (dir == 0 || dir == 1), "cast: bad args");
  VQ3_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "cast: pointers must be 16-byte aligned");
  if (dir == 0)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(flat_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, (bf16_t*)y, (long)n);
  else
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(flat_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (float*)y, (long)n);
  VQ3_CHECK_LAUNCH("cast");
  return 0;
}

extern "C" int vq3_f32_to_bf16_acc(const float* src, void* acc_bf16, int64_t n, int32_t accumulate, void* stream) {
  VQ3_CHECK_ARG(src && acc_bf16 && n > 0, "f32_to_bf16_acc: bad args");
  hipLaunchKernelGGL(f32_to_bf16_acc_kernel, dim3(flat_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)acc_bf16, (long)n, accumulate);
  VQ3_CHECK_LAUNCH("f32_to_bf16_acc");
  return 0;
}

extern "C" int vq3_gather_rows(const void* src, const int32_t* idx, void* out, int32_t n, int32_t n_pad, int32_t cols,
                               void* stream) {
  VQ3_CHECK_ARG(src && idx && out && n >= 0 && n_pad >= n && n_pad > 0 && cols > 0 && cols % 8 == 0,
                "gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_pad), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx,
                     (bf16_t*)out, n, cols);
  VQ3_CHECK_LAUNCH("gather_rows");
  return 0;
}
extern "C" int vq3_scatter_rows(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t cols,
                                int32_t accumulate, void* stream) {
  VQ3_CHECK_ARG(src && idx && dst && n > 0 && cols > 0 && cols % 8 == 0, "scatter_rows: bad args");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx,
                     (bf16_t*)dst, cols, accumulate);
  VQ3_CHECK_LAUNCH("scatter_rows");
  return 0;
}

extern "C" int vq3_embed_splice_fwd(const int64_t* ids, const void* table, const void* feat, const int32_t* srcmap,
                                    void* out, int32_t B, int32_t L, int32_t H, int32_t S, void* stream) {
  VQ3_CHECK_ARG(ids && table && srcmap && out, "embed_splice_fwd: null pointer");
  VQ3_CHECK_ARG(B > 0 && L > 0 && H > 0 && H % 8 == 0 && S >= 0, "embed_splice_fwd: bad shape");
  VQ3_CHECK_ARG(S == 0 || feat, "embed_splice_fwd: feat is null but S > 0");
  hipLaunchKernelGGL(embed_splice_fwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, ids,
                     (const bf16_t*)table, (const bf16_t*)feat, srcmap, (bf16_t*)out, L, H, S);
  VQ3_CHECK_LAUNCH("embed_splice_fwd");
  return 0;
}

extern "C" int vq3_embed_splice_bwd(const int64_t* sorted_ids, const int64_t* order, const int32_t* srcmap,
                                    const void* dout, void* dtable_bf16, float* dfeat_f32, int32_t B, int32_t L,
                                    int32_t H, int32_t S, void* stream) {
  VQ3_CHECK_ARG(sorted_ids && order && srcmap && dout, "embed_splice_bwd: null pointer");
  VQ3_CHECK_ARG(B > 0 && L > 0 && H > 0 && H % 8 == 0, "embed_splice_bwd: bad shape");
  if (dtable_bf16) {
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * L, (H + 511) / 512), dim3(256), 0, (hipStream_t)stream, sorted_ids, order, srcmap,
                       (const bf16_t*)dout, (bf16_t*)dtable_bf16, B * L, H);
    VQ3_CHECK_LAUNCH("embed_bwd");
  }
  if (dfeat_f32 && S > 0) {
    hipLaunchKernelGGL(splice_bwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, srcmap, (const bf16_t*)dout,
                       dfeat_f32, L, H, S);
    VQ3_CHECK_LAUNCH("splice_bwd");
  }
  return 0;
}

extern "C" int vq3_softmax_fwd(const float* S, void* P, const uint8_t* keymask, int32_t nb, int32_t heads_per_mask,
                               int32_t Lq, int32_t Lk, int32_t ldS, int32_t ldP, int32_t causal, void* stream) {
  VQ3_CHECK_ARG(S && P && nb > 0 && Lq > 0 && Lk > 0 && ldS >= Lk && ldP >= Lk && heads_per_mask >= 1,
                "softmax_fwd: bad args");
  const long nrows = (long)nb * Lq;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S,
                     (bf16_t*)P, keymask, heads_per_mask, Lq, Lk, ldS, ldP, causal, nrows);
  VQ3_CHECK_LAUNCH("softmax_fwd");
  return 0;
}
extern "C" int vq3_softmax_bwd(const void* P, const float* dP, void* dS, int32_t nb, int32_t Lq, int32_t Lk,
                               int32_t ldS, int32_t ldP, float scale, void* stream) {
  VQ3_CHECK_ARG(P && dP && dS && nb > 0 && Lq > 0 && Lk > 0 && ldS >= Lk && ldP >= Lk, "softmax_bwd: bad args");
  const long nrows = (long)nb * Lq;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)P, dP, (bf16_t*)dS, Lk, ldS, ldP, scale, nrows);
  VQ3_CHECK_LAUNCH("softmax_bwd");
  return 0;
}

extern "C" int vq3_cross_entropy_fwd_bwd(void* logits, const int32_t* targets, float* loss_sum_f32, int32_t n,
                                         int32_t V, int32_t ldl, float gscale, void* stream) {
  VQ3_CHECK_ARG(logits && targets && loss_sum_f32 && n > 0 && V > 0 && ldl >= V, "cross_entropy: bad args");
  VQ3_CHECK_ARG(ldl % 8 == 0 && (uintptr_t)logits % 16 == 0, "cross_entropy: rows must be 16-byte aligned (ldl %% 8 == 0)");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, (bf16_t*)logits, targets,
                     loss_sum_f32, V, ldl, gscale, (const float*)nullptr, (float*)nullptr);
  VQ3_CHECK_LAUNCH("cross_entropy");
  return 0;
}

extern "C" int vq3_cross_entropy_rows(void* logits, const int32_t* targets, const float* row_scale, float* row_loss, int32_t n,
                                      int32_t V, int32_t ldl, void* stream) {
  VQ3_CHECK_ARG(logits && targets && row_scale && row_loss && n > 0 && V > 0 && ldl >= V, "cross_entropy_rows: bad args");
  VQ3_CHECK_ARG(ldl % 8 == 0 && (uintptr_t)logits % 16 == 0, "cross_entropy_rows: rows must be 16-byte aligned (ldl %% 8 == 0)");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, (bf16_t*)logits, targets,
                     (float*)nullptr, V, ldl, 0.f, row_scale, row_loss);
  VQ3_CHECK_LAUNCH("cross_entropy_rows");
  return 0;
}

extern "C" int vq3_adamw_step(float* master, float* m, float* v, const void* grad_bf16, void* w_bf16, int64_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                              float gscale, const float* clip_sumsq, float max_norm, void* stream) {
  VQ3_CHECK_ARG(master && m && v && grad_bf16 && w_bf16 && n > 0 && step >= 1, "adamw: bad args");
  VQ3_CHECK_ARG(((uintptr_t)master % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                    ((uintptr_t)grad_bf16 % 8 == 0) && ((uintptr_t)w_bf16 % 8 == 0),
                "adamw: buffers must be 16-byte (f32) / 8-byte (bf16) aligned");
  VQ3_CHECK_ARG(!clip_sumsq || max_norm > 0.f, "adamw: clipping needs max_norm > 0");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, master, m, v,
                     (const bf16_t*)grad_bf16, (bf16_t*)w_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                     gscale, clip_sumsq, max_norm);
  VQ3_CHECK_LAUNCH("adamw");
  return 0;
}

extern "C" int vq3_sumsq(const void* x, int32_t is_f32, int64_t n, float* partials, float* accum, void* stream) {
  VQ3_CHECK_ARG(x && partials && accum && n > 0, "sumsq: bad args");
  VQ3_CHECK_ARG((uintptr_t)x % 16 == 0, "sumsq: x must be 16-byte aligned");
  const long vec = is_f32 ? n / 4 : n / 8;
  long nb = (vec + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
  if (is_f32)
    hipLaunchKernelGGL(sumsq_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long)n, partials);
  else
    hipLaunchKernelGGL(sumsq_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long)n, partials);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, (int)nb, accum);
  VQ3_CHECK_LAUNCH("sumsq");
  return 0;
}

extern "C" int vq3_dropout(void* x, int32_t is_f32, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  VQ3_CHECK_ARG(x && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args (0 <= p < 1)");
  if (p == 0.f) return 0;
  const unsigned thresh = (unsigned)(p * 16777216.0f);
  const float scale = 1.f / (1.f - p);
  if (is_f32)
    hipLaunchKernelGGL(dropout_kernel<true>, dim3(flat_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n, thresh,
                       scale, (unsigned long long)seed, (unsigned long long)offset);
  else
    hipLaunchKernelGGL(dropout_kernel<false>, dim3(flat_grid(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n, thresh,
                       scale, (unsigned long long)seed, (unsigned long long)offset);
  VQ3_CHECK_LAUNCH("dropout");
  return 0;
}
