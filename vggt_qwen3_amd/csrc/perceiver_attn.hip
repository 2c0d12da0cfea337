// Fused cross-attention of a PerceiverLayer for gfx950: nn.MultiheadAttention(latents, context, context)'s
// softmax(alpha q k^T) -> dropout -> . v for one (sample, head) and 64 latent rows per workgroup (SURVEY.md 7.1-4;
// /root/reference/src/models/projector_perceiver.py:33,44). The f32 score tensor [B*H, N, T], the softmax launch and the two batched
// GEMMs either side of it are gone; P / dropout(P) are written only when the caller keeps them for a backward pass.
//
// 4 waves x 16 latent rows. A wave keeps its Q rows in registers as MFMA B fragments (HD / 32 x 16 bytes per lane: lane = (row fr,
// depth 8 g ..)), K / V pass through LDS in chunks of KC = 32 context rows shared by the four waves, staged through registers with
// the next chunk's rows requested before this chunk is multiplied. Two passes over the chunks:
//   1  S^T = K . Q^T (v_mfma_f32_16x16x32_bf16, A = K rows, B = Q rows: a lane ends with 4 keys 4 g + r of its query fr per 16-key
//      tile) -> running (max, sum of exponentials) per lane, merged over the four lane groups at the end;
//   2  S^T again, P = bf16(exp(alpha s - max) / sum) - the formula AND the rounding point of vq3_softmax_fwd, so a caller that keeps
//      P gets what the three-launch route produced - then the dropout decision of vq3_dropout for element ((b H + h) N + n) Tp + t
//      of the P tensor, and O^T += V^T . Pd^T with the two tiles' quads as they stand as the B fragment (contraction slot 8 g + j =
//      key 4 g + j of tile 0 for j < 4, of tile 1 for j >= 4) and V^T read from the ROW-MAJOR V chunk through ds_read_b64_tr_b16
//      (4 keys x 16 depth columns per 16-lane group) in the same key order.
// Up to 128 context rows (the reference: 128) the scores of pass 1 stay in registers (32 per lane) and pass 2 only streams V
// (KEEP_S); beyond that S is recomputed, a third more MFMA work than an online softmax would do and no rescaling of a 16 x HD
// accumulator. The whole product is 13 GFLOP per layer at the reference's sizes (48 samples x 8 heads, 128 x 128 x 512): what the
// launch costs is its 200 MB of q / k / v / o traffic and the latency of 8 chunk hand-overs per workgroup, not the matrix pipe.
// LDS: two chunk regions; K rows at a pitch of 2 HD + 16 bytes (the 16 rows a b128 fragment read touches land in 16 different
// 16-byte bank groups), V rows at 2 HD + 32 (the 4 rows x 32 bytes of a transposed read do not overlap): 66 KiB at HD = 512, two
// workgroups per CU.
#include "common.h"
#include "vq3_hip.h"

namespace {

struct XattnParams {
  const bf16_t* q; const bf16_t* kv; bf16_t* o; bf16_t* P; bf16_t* Pd;
  int Hh, N, T, Tp;
  long ldq, ldkv, ldo, voff;
  float alpha, scale;
  unsigned thresh;
  unsigned long long seed, offset;
};

constexpr int KC = 32;
typedef short s16x4 __attribute__((ext_vector_type(4)));

// (the any-T variant carries K and V prefetch registers beside the 16 x HD accumulator: one workgroup per CU from HD = 256 on)
template <int HD, bool KEEP_S>
__global__ __launch_bounds__(256, (KEEP_S || HD <= 128) ? 2 : 1) void perceiver_xattn_kernel(XattnParams p) {
  constexpr int KP = HD * 2 + 16, VP = HD * 2 + 32;       // row pitches in bytes
  constexpr int REG = KC * VP;                            // one chunk region (a K chunk fits: KP < VP); two regions
  constexpr int CPR = HD / 8;                             // 16-byte pieces per row
  constexpr int NPRE = KC * CPR / 256;                    // pieces per thread and chunk (HD / 64)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qrow = blockIdx.x * 64 + wid * 16 + fr;
  const bool qvalid = qrow < p.N;
  const int qr = qvalid ? qrow : p.N - 1;
  const long bh = (long)b * p.Hh + h;
  const long prow = (bh * p.N + qr) * (long)p.Tp;         // element index of P[b, h, query, 0]

  bf16x8 qf[HD / 32];
  {
    const bf16_t* qp = p.q + ((long)b * p.N + qr) * p.ldq + (long)h * HD + 8 * g;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s);
  }
  const bf16_t* kvb = p.kv + (long)b * p.T * p.ldkv + (long)h * HD;

  // chunk staging through registers: the next chunk's rows are requested before this chunk is multiplied
  auto fetch = [&](int c0, long coloff, u32x4 (&buf)[NPRE]) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + 256 * i, r = idx / CPR, c = idx % CPR;
      buf[i] = u32x4{0u, 0u, 0u, 0u};                     // rows past T: zeros (0 * finite in the products)
      if (c0 + r < p.T) buf[i] = *reinterpret_cast<const u32x4*>(kvb + (long)(c0 + r) * p.ldkv + coloff + c * 8);
    }
  };
  auto put = [&](int base, int pitch, const u32x4 (&buf)[NPRE]) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + 256 * i, r = idx / CPR, c = idx % CPR;
      *reinterpret_cast<u32x4*>(smem + base + r * pitch + c * 16) = buf[i];
    }
  };
  // S^T tile of the K chunk at `base`: keys 16 tile + 4 g + r (r = 0..3) of query fr
  auto s_tile = [&](int base, int tile) -> f32x4 {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const char* kb = smem + base + (16 * tile + fr) * KP + 16 * g;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + 64 * s);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc, 0, 0, 0);
    }
    return acc;
  };
  // bf16 P quad of keys k0 .. k0 + 3 from scores, its dropped-out copy as the return value; both stored when kept
  auto p_quad = [&](const f32x4& s, int k0, float M, float inv) -> u32x2 {
    bf16_t pb[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = (k0 + r < p.T) ? __expf(p.alpha * s[r] - M) * inv : 0.f;
      pb[r] = f2bf(e);
      pd[r] = pb[r];
      if (p.thresh) {
        const bool keep = drop_bits(p.seed, p.offset + (unsigned long long)(prow + k0 + r)) >= p.thresh;
        pd[r] = keep ? f2bf(bf2f(pb[r]) * p.scale) : (bf16_t)0;
      }
    }
    const u32x2 wpb = {(uint32_t)pb[0] | ((uint32_t)pb[1] << 16), (uint32_t)pb[2] | ((uint32_t)pb[3] << 16)};
    const u32x2 wpd = {(uint32_t)pd[0] | ((uint32_t)pd[1] << 16), (uint32_t)pd[2] | ((uint32_t)pd[3] << 16)};
    if (qvalid && k0 < p.Tp) {
      if (p.P) *reinterpret_cast<u32x2*>(p.P + prow + k0) = wpb;
      if (p.Pd) *reinterpret_cast<u32x2*>(p.Pd + prow + k0) = wpd;
    }
    return wpd;
  };
  f32x4 oacc[HD / 16];
#pragma unroll
  for (int d = 0; d < HD / 16; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  // O^T += V^T . Pd^T over the V chunk at `base`. V^T fragments: lane fr of group g addresses row 4 g + (fr >> 2), 8-byte piece
  // fr & 3 of a 4-row x 32-byte block and receives column fr of it (keys 4 g .. 4 g + 3 of depth column 16 d + fr); the second
  // read takes tile 1's rows - the key order of the B fragment (tile 0's quad, tile 1's quad)
  auto pv = [&](int base, const u32x2& w0, const u32x2& w1) {
    const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{w0[0], w0[1], w1[0], w1[1]});
    const char* vb = smem + base + (4 * g + (fr >> 2)) * VP + (fr & 3) * 8;
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + d * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 16 * VP + d * 32));
      const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[d], 0, 0, 0);
    }
  };

  u32x4 pre[NPRE];
  if constexpr (KEEP_S) {
    // ---- T <= 128: the scores of all (<= 4) chunks stay in registers; chunks alternate between the two LDS regions, so one
    // barrier per chunk orders "everyone has finished reading region x two chunks ago" before it is overwritten
    f32x4 sreg[8];
    fetch(0, 0, pre);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      sreg[2 * ci] = sreg[2 * ci + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ci * KC < p.T) {
        put((ci & 1) * REG, KP, pre);
        __syncthreads();
        if ((ci + 1) * KC < p.T) fetch((ci + 1) * KC, 0, pre);
        else fetch(0, p.voff, pre);                        // the first V chunk rides behind the last K chunk
        sreg[2 * ci] = s_tile((ci & 1) * REG, 0);
        sreg[2 * ci + 1] = s_tile((ci & 1) * REG, 1);
      }
    }
    float M = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * j + 4 * g + r < p.T) M = fmaxf(M, p.alpha * sreg[j][r]);
    M = fmaxf(M, __shfl_xor(M, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    float lt = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * j + 4 * g + r < p.T) lt += __expf(p.alpha * sreg[j][r] - M);
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = (lt > 0.f) ? 1.f / lt : 0.f;
    const int nck = (p.T + KC - 1) / KC;
    u32x2 pw[8];                                           // (both tiles of every visited chunk: a kept P is written up to 32 nck)
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j] = (j < 2 * nck) ? p_quad(sreg[j], 16 * j + 4 * g, M, inv) : u32x2{0u, 0u};
    __syncthreads();                                       // (the last K chunk's readers, before region parity restarts at 0)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      if (ci < nck) {
        put((ci & 1) * REG, VP, pre);
        __syncthreads();
        if (ci + 1 < nck) fetch((ci + 1) * KC, p.voff, pre);
        pv((ci & 1) * REG, pw[2 * ci], pw[2 * ci + 1]);
      }
    }
  } else {
    // ---- any T: pass 1 keeps a running (max, sum of exponentials) per lane, pass 2 multiplies K . Q^T again
    float m = -INFINITY, l = 0.f;
    fetch(0, 0, pre);
    int par = 0;
    for (int c0 = 0; c0 < p.T; c0 += KC, par ^= 1) {
      put(par * REG, KP, pre);
      __syncthreads();
      if (c0 + KC < p.T) fetch(c0 + KC, 0, pre);
      float v[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 s = s_tile(par * REG, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * t + r] = (c0 + 16 * t + 4 * g + r < p.T) ? p.alpha * s[r] : -INFINITY;
      }
      float cm = v[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
      const float mn = fmaxf(m, cm);
      if (mn > -INFINITY) {
        float e = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) e += __expf(v[i] - mn);
        l = l * __expf(m - mn) + e;
        m = mn;
      }
    }
    float M = fmaxf(m, __shfl_xor(m, 16, 64));
    M = fmaxf(M, __shfl_xor(M, 32, 64));
    float lt = (m > -INFINITY) ? l * __expf(m - M) : 0.f;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = (lt > 0.f) ? 1.f / lt : 0.f;
    u32x4 prev[NPRE];
    fetch(0, 0, pre);
    fetch(0, p.voff, prev);
    for (int c0 = 0; c0 < p.T; c0 += KC) {
      __syncthreads();                                     // the previous chunk's readers (K in region 0, V in region 1)
      put(0, KP, pre);
      put(REG, VP, prev);
      __syncthreads();
      if (c0 + KC < p.T) { fetch(c0 + KC, 0, pre); fetch(c0 + KC, p.voff, prev); }
      const u32x2 w0 = p_quad(s_tile(0, 0), c0 + 4 * g, M, inv);
      const u32x2 w1 = p_quad(s_tile(0, 1), c0 + 16 + 4 * g, M, inv);
      pv(REG, w0, w1);
    }
  }
  if (qvalid) {
    // pad columns [32-multiple past T, Tp) of a kept P row are never visited by the chunk loop: zero them (vq3_softmax_fwd does)
    const int tcov = (p.T + KC - 1) / KC * KC;
    for (int k0 = tcov + 4 * g; k0 < p.Tp; k0 += 16) {
      if (p.P) *reinterpret_cast<u32x2*>(p.P + prow + k0) = u32x2{0u, 0u};
      if (p.Pd) *reinterpret_cast<u32x2*>(p.Pd + prow + k0) = u32x2{0u, 0u};
    }
    // O^T[depth 16 d + 4 g + r][query fr] -> O[b N + query][h HD + depth]: 4 consecutive depth columns per lane
    bf16_t* op = p.o + ((long)b * p.N + qrow) * p.ldo + (long)h * HD + 4 * g;
#pragma unroll
    for (int d = 0; d < HD / 16; ++d)
      *reinterpret_cast<u32x2*>(op + 16 * d) = u32x2{pack2bf(oacc[d][0], oacc[d][1]), pack2bf(oacc[d][2], oacc[d][3])};
  }
}

template <int HD, bool KEEP_S>
int launch(const XattnParams& p, int B, hipStream_t stream) {
  constexpr int bytes = 2 * KC * (HD * 2 + 32);
  static bool attr_done = false;
  if (bytes > 48 * 1024 && !attr_done) {
    if (hipFuncSetAttribute((const void*)perceiver_xattn_kernel<HD, KEEP_S>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
      vq3_set_error("perceiver_xattn: cannot reserve %d bytes of LDS", bytes);
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((perceiver_xattn_kernel<HD, KEEP_S>), dim3((unsigned)((p.N + 63) / 64), (unsigned)p.Hh, (unsigned)B), dim3(256), bytes,
                     stream, p);
  VQ3_CHECK_LAUNCH("perceiver_xattn");
  return 0;
}
template <int HD>
int launch_hd(const XattnParams& p, int B, hipStream_t stream) {
  return p.T <= 4 * KC ? launch<HD, true>(p, B, stream) : launch<HD, false>(p, B, stream);
}

}  // namespace

extern "C" int vq3_perceiver_xattn_fwd(const void* q, const void* kv, void* o, void* P, void* Pd, int32_t B, int32_t H, int32_t N,
                                       int32_t T, int32_t head_dim, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t v_off, int32_t Tp,
                                       float alpha, float p_drop, uint64_t seed, uint64_t offset, void* stream) {
  VQ3_CHECK_ARG(q && kv && o, "perceiver_xattn: null pointer");
  VQ3_CHECK_ARG(B > 0 && H > 0 && N > 0 && T > 0 && B <= 65535 && H <= 65535, "perceiver_xattn: bad sizes");
  VQ3_CHECK_ARG(head_dim == 64 || head_dim == 128 || head_dim == 256 || head_dim == 512,
                "perceiver_xattn: head_dim must be 64, 128, 256 or 512 (got %d)", head_dim);
  VQ3_CHECK_ARG(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0 && v_off % 8 == 0 && ldq >= (int64_t)H * head_dim && ldo >= (int64_t)H * head_dim &&
                    ldkv >= v_off + (int64_t)H * head_dim && v_off >= 0,
                "perceiver_xattn: leading dimensions must cover H * head_dim and keep 16-byte rows (ldq %% 8, ldkv %% 8, v_off %% 8, ldo %% 4)");
  VQ3_CHECK_ARG((uintptr_t)q % 16 == 0 && (uintptr_t)kv % 16 == 0 && (uintptr_t)o % 8 == 0, "perceiver_xattn: q / kv must be 16-byte, o 8-byte aligned");
  VQ3_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "perceiver_xattn: 0 <= p_drop < 1");
  VQ3_CHECK_ARG((!P && !Pd) || (Tp >= T && Tp % 4 == 0 && (!P || (uintptr_t)P % 8 == 0) && (!Pd || (uintptr_t)Pd % 8 == 0)),
                "perceiver_xattn: a kept P needs Tp >= T, Tp %% 4 == 0 and 8-byte aligned buffers");
  VQ3_CHECK_ARG(Tp >= T, "perceiver_xattn: Tp (row length of the P tensor the dropout mask is indexed by) must be >= T");
  XattnParams p;
  p.q = (const bf16_t*)q; p.kv = (const bf16_t*)kv; p.o = (bf16_t*)o; p.P = (bf16_t*)P; p.Pd = (bf16_t*)Pd;
  p.Hh = H; p.N = N; p.T = T; p.Tp = Tp;
  p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo; p.voff = v_off;
  p.alpha = alpha;
  p.thresh = (unsigned)(p_drop * 16777216.0f);
  p.scale = 1.f / (1.f - p_drop);
  p.seed = seed; p.offset = offset;
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 64: return launch_hd<64>(p, B, st);
    case 128: return launch_hd<128>(p, B, st);
    case 256: return launch_hd<256>(p, B, st);
    default: return launch_hd<512>(p, B, st);
  }
}
