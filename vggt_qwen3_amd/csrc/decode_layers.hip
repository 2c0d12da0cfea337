// All decoder layers of one greedy-decoding step (one token, B = 1) in ONE persistent launch: vq3_qwen_decode_layers.
// (the reference's inference callers decode one prompt at a time: src/inference/qa_inference.py:207-216, arkit_inference.py:274-284;
// the layer being computed is transformers' Qwen3DecoderLayer, modeling_qwen3.py:49-83, 185-207, 237-330.)
//
// Why: with one launch per projection (decode.hip) a layer is six 8-29 us kernels whose 21-100 MB of weights stream in 3-17 us - every
// kernel pays its launch, first-byte latency, reduction and drain, and the HBM pipe is idle in between (2.99 ms / token = 0.34 of the
// 8 TB/s peak). Here the weight stream never stops:
//
//  * 256 workgroups (one per CU, made exclusive by their LDS request) x 12 waves. Waves 0-7 STREAM: a workgroup owns a fixed slab of
//    rows of every projection (24 of q|k|v, 10 of o, 38 + 38 of gate | up, 10 of down - contiguous bytes, 16 per lane and step), and a
//    thread's 97 16-byte pieces per layer form one sequence with RING = 20 of them in flight at any time (a ring of registers, static
//    indices) - ACROSS the phases of a layer and across layers, because weight addresses depend on nothing the token computes. The
//    phases' synchronisation therefore gates the FMAs, not the loads.
//  * Waves 8-11 HELP: they wait at the grid barrier, bring the phase's activation row into LDS (bf16; RMSNorm in registers / the
//    attention output / the SwiGLU product), and after the streamers' partial sums are in LDS add them per row in a fixed order, apply
//    the epilogue (residual - kept in a register of the thread that produced it - / SwiGLU) and publish the rows two bf16 at a time.
//    They hold no weight loads. Attention for the new token (q/k RMSNorm + RoPE, cache append, softmax over the cached positions) is
//    helper work of workgroups 0 .. Hq-1, one head each, in decode.hip's arithmetic order; the first 128 cached key rows are requested
//    while q|k|v is still being multiplied.
//  * Five grid barriers per layer (q|k|v -> attention -> o -> gate|up -> down): 16 counter words 256 bytes apart, arrival = relaxed
//    agent-scope add once the workgroup's stores are acknowledged, wait = one wave per workgroup reading the 16 words. Rows that cross
//    workgroups are stored and loaded with agent-scope accesses (no cache-wide write-back / invalidate).
//    Every wait is BOUNDED: after 2^21 polls the waiter raises the status word, every later wait returns at once, the grid drains and
//    the host reports the failure (the kernel cannot hang the device when a workgroup was not co-resident).
//  * No scratch: a private segment makes the runtime set up scratch for every wave slot at each launch (0.3 ms per token, measured) -
//    hence the two role instantiations of the layer loop, the laundered indices and the norm weights staged through LDS below.
//
// Arithmetic is that of decode.hip (bf16 rounding points: normed row, projection outputs before the residual, SwiGLU product,
// probabilities; fp32 sums), only the ORDER of a dot product's fp32 terms differs (v_dot2c pairs, 16-lane groups per 1 KiB of a row
// instead of a wave per row), as it does between any two GEMV kernels. Measured: DESIGN.md section 7 (2.99 -> 2.33-2.38 ms / token).
#include <type_traits>
#include <utility>

#include "common.h"
#include "vq3_hip.h"

// (diagnostic build: make EXTRA=-DVQ3_DL_STAMPS, then tools/diag/decode_layers_stamps.py prints where a layer's microseconds go)
#ifdef VQ3_DL_STAMPS
__device__ unsigned long long dl_stamps[2][32];
__device__ unsigned long long dl_layer_end[64];     // [0] kernel start, [1 + l] end of layer l (workgroup 0)
#define DL_STAMP(i)                                                                        \
  do {                                                                                     \
    if (l == 1 && (wg == 0 || wg == 37) && lead) dl_stamps[wg == 0 ? 0 : 1][i] = wall_clock64(); \
  } while (0)
extern "C" int vq3_debug_decode_stamps(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dl_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return 1;
  return (int)hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(dl_layer_end), sizeof(unsigned long long) * 64);
}
#else
#define DL_STAMP(i)
#endif

namespace {

// f(integral_constant<int, B>{}), ..., f(integral_constant<int, E - 1>{})
template <int B, int E, class F>
__device__ __forceinline__ void dl_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    dl_static_for<B + 1, E>(f);
  }
}

constexpr int DL_G = 256;        // workgroups (= CUs of an MI355X)
constexpr int DL_TS = 512;       // streaming threads per workgroup
constexpr int DL_TH = 256;       // helper threads
#ifndef VQ3_DL_RING
#define VQ3_DL_RING 20
#endif
constexpr int DL_RING = VQ3_DL_RING;      // 16-byte weight loads in flight per streaming thread
constexpr int DL_D = 128;        // head_dim
constexpr int DL_LMAX = 2048;    // cache positions the score buffer holds
constexpr unsigned DL_POLLS = 1u << 21;

// (pointers read from device memory are generic to the compiler: typed as global here, so the weight stream is global_load with a
// scalar base + one per-thread offset, not flat_load - which would also count against the LDS counter)
typedef const __attribute__((address_space(1))) bf16_t* dl_gbf;
typedef const __attribute__((address_space(1))) char* dl_gchar;
typedef const __attribute__((address_space(1))) u32x4* dl_gvec;
struct LayerW { dl_gbf qkv, o, gu, down, ln1, ln2, qn, kn; };

struct Args {
  const LayerW* w;               // [nl] (device)
  bf16_t* h;                     // [H]: the layer stack's input row, overwritten by its output
  bf16_t *qkv, *ao, *hmid, *act, *hx;  // workspace rows (bf16): (Hq + 2 Hkv) 128, Hq 128, H, I, H
  const bf16_t *cs, *sn;         // RoPE tables [>= Lmax, 128]
  const int32_t* lens;           // [1]: cached positions = position of the new token
  bf16_t *Kc, *Vc;               // layer 0's cache [Hkv, Lmax, 128]
  long cache_stride;             // elements between two layers' caches
  unsigned* bar;                 // DL_NBAR arrival counters, DL_BAR_STRIDE words apart (zero at launch)
  unsigned* status;              // sticky: 1 = a barrier wait ran out, 2 = cache full
  int nl, Lmax;
  float eps, scale;
};

__device__ __forceinline__ void dl_unpack8(const u32x4 v, float* f) {
  f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, v.z << 16); f[5] = __builtin_bit_cast(float, v.z & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, v.w << 16); f[7] = __builtin_bit_cast(float, v.w & 0xffff0000u);
}

typedef __bf16 dl_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dl_dot2(unsigned w, unsigned x, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dl_bf2, w), __builtin_bit_cast(dl_bf2, x), c, false);
}

// sum over the 16 lanes of a DPP row, valid in the row's lane 15 (VALU only: no LDS round trip per step)
__device__ __forceinline__ float dl_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
  return v;
}

// The arrival counter is DL_NBAR words 256 bytes apart (workgroup g adds to word g % DL_NBAR; the waiting wave reads all of them in one
// round trip and adds): 256 agent-scope adds on ONE address serialise at the memory side - 6-8 us per barrier measured, against ~1 us so.
#ifndef VQ3_DL_NBAR
#define VQ3_DL_NBAR 16
#endif
constexpr int DL_NBAR = VQ3_DL_NBAR, DL_BAR_STRIDE = 64;
// called by a whole wave (helper wave 0); returns when the sum of the counters has reached target (or the status word is raised)
__device__ __forceinline__ void dl_grid_wait(unsigned* bar, unsigned* status, unsigned target, int lane) {
  unsigned n = 0;
  for (;;) {
    unsigned v = lane < DL_NBAR ? __hip_atomic_load(bar + lane * DL_BAR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
#ifndef VQ3_DL_SLEEP
#define VQ3_DL_SLEEP 1
#endif
    if (VQ3_DL_SLEEP > 0) __builtin_amdgcn_s_sleep(VQ3_DL_SLEEP);
    ++n;
    if ((n & 255u) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    if (n > DL_POLLS) {
      if (lane == 0) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}
__device__ __forceinline__ void dl_grid_arrive(unsigned* bar, int wg) {
  __hip_atomic_fetch_add(bar + (wg % DL_NBAR) * DL_BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Rows that cross workgroups (and XCDs, whose L2s are not coherent with each other) move through agent-scope accesses: the store is
// written through, the load bypasses the non-coherent levels. The first version published with a release fence and read behind an
// acquire fence instead - buffer_wbl2 / buffer_inv sc1 from 1024 helper waves five times per layer: 8-10 us per phase
// (tools/diag/decode_layers_stamps.py), whatever the depth of the weight ring.
__device__ __forceinline__ unsigned dl_ld32(const bf16_t* p) {      // two bf16
  return __hip_atomic_load(reinterpret_cast<unsigned*>(const_cast<bf16_t*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x2 dl_ld64(const bf16_t* p) {         // four bf16
  const unsigned long long v =
      __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<bf16_t*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x2{(unsigned)v, (unsigned)(v >> 32)};
}
// rows leave their producers two values at a time: thread 2 i stores (value of thread 2 i, value of thread 2 i + 1) as one 32-bit word
// (called by WHOLE waves; idx < n decides who stores, n even)
__device__ __forceinline__ void dl_st_pair(bf16_t* row, int idx, int n, float v) {
  const float nb = __shfl_down(v, 1, 64);
  if ((idx & 1) == 0 && idx < n)
    __hip_atomic_store(reinterpret_cast<unsigned*>(row + idx), pack2bf(v, nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the wave's stores have been acknowledged (CDNA4 counts stores in vmcnt): what a workgroup's arrival at the grid barrier promises.
// (Tried instead: a 16-bit epoch in the low half of every workspace word, checked and re-read by the consumer, so that a producer arrives
// without waiting for the acknowledgement - the store's visibility latency is on the critical path either way, the readers' retries add
// traffic, and the 36-layer step got slower: 2.85 -> 4.65 ms / token.)
__device__ __forceinline__ void dl_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// H hidden, I intermediate, HQ / HKV heads (head_dim 128). Every projection's rows divide over the 256 workgroups and every K is a
// multiple of 512 (a wave's 64 x 16 bytes stay inside one weight row).
template <int H, int I, int HQ, int HKV>
struct Geo {
  static constexpr int NQKV = (HQ + 2 * HKV) * DL_D, KO = HQ * DL_D;
  static_assert(NQKV % DL_G == 0 && H % DL_G == 0 && I % DL_G == 0, "rows divide over the workgroups");
  static_assert(H % 512 == 0 && KO % 512 == 0 && I % 512 == 0, "a wave's 1 KiB stays inside one weight row");
  static constexpr int R0 = NQKV / DL_G, R2 = H / DL_G, RG = I / DL_G, R4 = H / DL_G;
  static constexpr int C0 = R0 * (H / 8), C2 = R2 * (KO / 8), C3H = RG * (H / 8), C3 = 2 * C3H, C4 = R4 * (I / 8);   // 16-byte pieces per workgroup
  static_assert(C3H % 64 == 0, "the gate | up slab boundary is wave-aligned");
  static constexpr int S0 = (C0 + DL_TS - 1) / DL_TS, S2 = (C2 + DL_TS - 1) / DL_TS, S3 = (C3 + DL_TS - 1) / DL_TS, S4 = (C4 + DL_TS - 1) / DL_TS;
  static constexpr int B0 = 0, B2 = S0, B3 = S0 + S2, B4 = S0 + S2 + S3, NS = S0 + S2 + S3 + S4;
  static constexpr int NSLOT = (NS + DL_RING - 1) / DL_RING * DL_RING;      // padded so that a slot's ring register is the same in every layer
  static constexpr int KMAX = (I > KO ? (I > H ? I : H) : (KO > H ? KO : H));
  static constexpr int UMAX = C3 / 64 > C4 / 64 ? C3 / 64 : C4 / 64;       // wave-sized pieces per phase and workgroup
  static constexpr int PER_LAYER = 4 * DL_G + HQ;                            // barrier arrivals per layer
  // LDS (bytes)
  static constexpr int XS = 0, PART = KMAX / 8 * 16, SC = PART + UMAX * 16 + 64, QS = SC + DL_LMAX * 4,
                       RED = QS + 3 * DL_D * 4, OPART = RED + 64, LNW = OPART + 4 * DL_D * 4, LDS_USED = LNW + 2 * H * 2;
};

template <int H, int I, int HQ, int HKV>
__global__ __launch_bounds__(DL_TS + DL_TH) void decode_layers_kernel(const Args a) {
  using Gm = Geo<H, I, HQ, HKV>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  u32x4* const xs = reinterpret_cast<u32x4*>(lds + Gm::XS);       // the phase's activation row, bf16: piece c = x[8 c .. 8 c + 7]
  float* const part = reinterpret_cast<float*>(lds + Gm::PART);   // [wave piece u][16-lane group]
  float* const sc = reinterpret_cast<float*>(lds + Gm::SC);
  float* const qs = reinterpret_cast<float*>(lds + Gm::QS);       // q', k', v of the new token (fp32 values on the bf16 grid)
  float* const red = reinterpret_cast<float*>(lds + Gm::RED);
  float* const opart = reinterpret_cast<float*>(lds + Gm::OPART);
  u32x4* const lnw1 = reinterpret_cast<u32x4*>(lds + Gm::LNW);   // input_layernorm / post_attention_layernorm weights of the layer, staged by
  u32x4* const lnw2 = lnw1 + H / 8;                               // helper wave 3 while the others work (no registers held across phases)

  const int tid = threadIdx.x;
  int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const bool streamer = wave < DL_TS / 64;
  int ht = tid - DL_TS;                                           // helper thread / wave index (negative for streamers)
  const int hw = wave - DL_TS / 64;
  const bool lead = tid == DL_TS;

  u32x4 ring[DL_RING];

  // ---- weight piece of slot S (compile time) for this thread: uniform base (scalar registers) + per-thread byte offset
#ifdef VQ3_DL_NOHBM
  const int wga = 0;     // diagnostic: every workgroup streams workgroup 0's slabs (L2 hits): what is left is the latency chain
#else
  const int wga = wg;
#endif
  const unsigned voff = (unsigned)tid * 16u;
  auto issue = [&](const LayerW& w, auto tag) {
    constexpr int S = decltype(tag)::value;
    if constexpr (S < Gm::NS) {
      dl_gchar base;
      unsigned off = voff;
      if constexpr (S < Gm::B2) {
        constexpr int j = S - Gm::B0;
        base = (dl_gchar)w.qkv + ((size_t)wga * Gm::R0 * H * 2 + (size_t)j * DL_TS * 16);
      } else if constexpr (S < Gm::B3) {
        constexpr int j = S - Gm::B2;
        base = (dl_gchar)w.o + ((size_t)wga * Gm::R2 * Gm::KO * 2 + (size_t)j * DL_TS * 16);
      } else if constexpr (S < Gm::B4) {
        constexpr int j = S - Gm::B3;
        constexpr int CB = DL_TS * j;                                // first piece of the step
        const dl_gchar gate = (dl_gchar)w.gu + (size_t)wga * Gm::RG * H * 2;
        const dl_gchar up = (dl_gchar)w.gu + ((size_t)I + (size_t)wga * Gm::RG) * H * 2;
        if constexpr (CB + DL_TS <= Gm::C3H) base = gate + (size_t)CB * 16;
        else if constexpr (CB >= Gm::C3H) {
          base = up + (size_t)(CB - Gm::C3H) * 16;
          if (j == Gm::S3 - 1 && CB + tid >= Gm::C3) off = 0;               // ragged last step: lanes past the slab re-read its first piece of the step (never summed)
        } else {
          base = (CB + tid < Gm::C3H) ? gate + (long)CB * 16 : up + ((long)CB - Gm::C3H) * 16;   // the slab boundary falls inside this step (wave-aligned)
        }
      } else {
        constexpr int j = S - Gm::B4;
        base = (dl_gchar)w.down + ((size_t)wga * Gm::R4 * I * 2 + (size_t)j * DL_TS * 16);
        if (j == Gm::S4 - 1 && DL_TS * j + tid >= Gm::C4) off = 0;
      }
      ring[S % DL_RING] = __builtin_nontemporal_load((dl_gvec)(base + off));
    }
  };
  // ---- one piece: 8 weights x 8 activations, 16-lane sums into part[u][lane / 16]
  auto consume = [&](auto tag) {
    constexpr int S = decltype(tag)::value;
    if constexpr (S < Gm::NS) {
      constexpr int PH = S < Gm::B2 ? 0 : (S < Gm::B3 ? 2 : (S < Gm::B4 ? 3 : 4));
      constexpr int j = S - (PH == 0 ? Gm::B0 : PH == 2 ? Gm::B2 : PH == 3 ? Gm::B3 : Gm::B4);
      constexpr int K = PH == 2 ? Gm::KO : (PH == 4 ? I : H);
      constexpr int WPR = K / 512;                                 // wave pieces per weight row
      constexpr int UTOT = (PH == 0 ? Gm::C0 : PH == 2 ? Gm::C2 : PH == 3 ? Gm::C3 : Gm::C4) / 64;
      // (the wave index is laundered per piece: left visible, the compiler hoists all 97 pieces' loop-invariant LDS addresses - two
      // registers each - out of the layer loop and spills the ring)
      int wv = wave;
      asm volatile("" : "+s"(wv));
      const int u = wv + 8 * j;
      const int kc = (u % WPR) * 64 + lane;
      // v_dot2c_f32_bf16: two products of bf16 pairs per instruction, straight from the packed words (no unpacking of weights or x)
      const u32x4 wq = ring[S % DL_RING], xq = xs[kc];
      float acc = dl_dot2(wq.x, xq.x, 0.f);
      acc = dl_dot2(wq.y, xq.y, acc);
      acc = dl_dot2(wq.z, xq.z, acc);
      acc = dl_dot2(wq.w, xq.w, acc);
      acc = dl_row16_sum(acc);
      if ((lane & 15) == 15 && u < UTOT) part[u * 4 + (lane >> 4)] = acc;
    }
  };
  // sum of a row's pieces in a fixed order
  auto row_sum = [&](int r, auto wpr_tag) {
    constexpr int WPR = decltype(wpr_tag)::value;
    f32x4 p[WPR];
#pragma unroll
    for (int u = 0; u < WPR; ++u) p[u] = *reinterpret_cast<const f32x4*>(part + (r * WPR + u) * 4);      // all reads in flight, then the adds
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < WPR; ++u) s += (p[u][0] + p[u][1]) + (p[u][2] + p[u][3]);
    return s;
  };
  // Qwen3RMSNorm of an H-element row into the x planes (helper wave 0: H / 512 pieces per lane, statistics by one wave_sum). The row is
  // the bf16 input of the stack (layer 0) or a workspace row.
  auto xprep_rms = [&](const bf16_t* row16, const bf16_t* rowws, const u32x4* lnv) {
    constexpr int NP = H / 512;
    float f[NP][8];
    float ss = 0.f;
    if (row16) {
#pragma unroll
      for (int i = 0; i < NP; ++i) dl_unpack8(*reinterpret_cast<const u32x4*>(row16 + (lane + 64 * i) * 8), f[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const u32x2 lo = dl_ld64(rowws + (lane + 64 * i) * 8), hi = dl_ld64(rowws + (lane + 64 * i) * 8 + 4);
        dl_unpack8(u32x4{lo[0], lo[1], hi[0], hi[1]}, f[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf(f[i][e], f[i][e], ss);
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + a.eps);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float lw[8];
      dl_unpack8(lnv[lane + 64 * i], lw);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[i][e] = rbf(lw[e] * rbf(f[i][e] * rstd));
      xs[lane + 64 * i] = u32x4{pack2bf(f[i][0], f[i][1]), pack2bf(f[i][2], f[i][3]), pack2bf(f[i][4], f[i][5]), pack2bf(f[i][6], f[i][7])};
    }
  };
  // a workspace row as it is (all helper threads)
  auto xprep_copy = [&](const bf16_t* row, int K) {
    for (int c = ht; c < K / 8; c += DL_TH) {
      const u32x2 lo = dl_ld64(row + c * 8), hi = dl_ld64(row + c * 8 + 4);
      xs[c] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  };

  // the first RING pieces of layer 0 are requested before anything else
  if (streamer) {
    const LayerW w0 = a.w[0];
    dl_static_for<0, DL_RING>([&](auto tag) { issue(w0, tag); });
  }
#ifdef VQ3_DL_STAMPS
  if (wg == 0 && lead) dl_layer_end[0] = wall_clock64();
#endif
  if (!streamer && hw == 3) {                                       // layer 0's first norm weight (later ones are staged a layer ahead)
    const LayerW w0 = a.w[0];
#pragma unroll
    for (int i = 0; i < H / 512; ++i) lnw1[lane + 64 * i] = *(dl_gvec)(w0.ln1 + (lane + 64 * i) * 8);
  }
  const int pos = a.lens[0];
  if (pos >= a.Lmax && lead) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool cache_ok = pos < a.Lmax;
  const bool attn = !streamer && wg < HQ && cache_ok;
  const int T = pos + 1;
  // the residual stream's rows this helper thread owns (h[n], h_mid[n] for n = wg * H / 256 + ht): produced and consumed by the same
  // thread, so they never travel
  float h_own = 0.f, hmid_own = 0.f;
  if (!streamer && ht < Gm::R2) h_own = bf2f(a.h[wg * Gm::R2 + ht]);

  // Two instantiations of the layer loop, one per role, with the same sequence of workgroup barriers: in one body the register allocator
  // sees a helper's prefetched cache rows live across the streamers' ring (it cannot know the branch is never taken by that wave) and
  // spills both.
  auto run = [&](auto role) {
    constexpr bool STREAM = decltype(role)::value;
    for (int l = 0; l < a.nl; ++l) {
      const LayerW w = a.w[l];
      const LayerW wn = a.w[l + 1 < a.nl ? l + 1 : l];               // past the last layer: harmless re-reads of its first pieces
      const unsigned base = (unsigned)l * Gm::PER_LAYER;
      // (helpers: the thread indices are laundered once per layer - visible as loop invariants they make the compiler hoist every
      // per-thread address of the layer out of the loop and spill them around the attention's row buffers)
      if constexpr (!STREAM) asm volatile("" : "+v"(lane), "+v"(ht));
      // attention operands that do not depend on this token: the head's cache, RoPE row and q / k norm weights
      bf16_t* const Kl = a.Kc + (long)l * a.cache_stride + (long)(wg / (HQ / HKV)) * a.Lmax * DL_D;
      bf16_t* const Vl = a.Vc + (long)l * a.cache_stride + (long)(wg / (HQ / HKV)) * a.Lmax * DL_D;
      const int sub = lane >> 4, dl = (lane & 15) * 8;
      const int last = pos > 0 ? pos - 1 : 0;                         // cached rows only; the new row comes from LDS (its store is another workgroup's)
      // rows c0 + 4 hw + 64 (it / 4) + 16 (it % 4) + sub, 8 elements from dl: decode.hip's key -> lane map, 128 keys per call
      auto load_rows = [&](const bf16_t* base_rows, int c0, u32x4 (&r)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
          r[it] = *reinterpret_cast<const u32x4*>(base_rows + (long)min(c0 + hw * 4 + 64 * (it >> 2) + 16 * (it & 3) + sub, last) * DL_D + dl);
      };
      u32x4 kpre[2][8];                                               // two chunks of key rows, then of value rows, in flight (registers: 64)
      float rc1 = 0.f, rc2 = 0.f, rs1 = 0.f, rs2 = 0.f, nw1 = 0.f, nw2 = 0.f;
      if constexpr (!STREAM) if (attn) {
        load_rows(Kl, 0, kpre[0]);
        if (hw < 2) {
          const dl_gbf nw = hw == 0 ? w.qn : w.kn;
          rc1 = bf2f(a.cs[(long)pos * DL_D + lane]); rc2 = bf2f(a.cs[(long)pos * DL_D + lane + 64]);
          rs1 = bf2f(a.sn[(long)pos * DL_D + lane]); rs2 = bf2f(a.sn[(long)pos * DL_D + lane + 64]);
          nw1 = bf2f(nw[lane]); nw2 = bf2f(nw[lane + 64]);
        }
      }
      // one streaming phase: pieces [SB, SE) are multiplied, the pieces RING ahead are requested into the registers just freed
      auto stream_phase = [&](auto sb, auto se) {
        constexpr int SB = decltype(sb)::value, SE = decltype(se)::value;
        dl_static_for<SB, SE>([&](auto tag) {
          constexpr int S = decltype(tag)::value, SN = S + DL_RING;
          consume(tag);
          if constexpr (SN < Gm::NSLOT) issue(w, std::integral_constant<int, SN>{});
          else issue(wn, std::integral_constant<int, SN - Gm::NSLOT>{});
        });
      };

      // (A phase is four workgroup barriers: "the grid barrier has been passed" - "x is in LDS" - "the partial sums are in LDS" - "the
      // rows have been stored". Tried: two, with helper wave 0 doing everything between a phase's sums and the next RMSNorm alone and every
      // helper wave polling the grid barrier itself before a copied x - A/B on one box 2.37-2.38 -> 2.44-2.62 ms / token: four polling
      // waves per workgroup instead of one cost more than the two barriers.)
      // ================= phase 0: q|k|v = W_qkv . RMSNorm(h)
      DL_STAMP(0);
      if constexpr (!STREAM) if (hw == 0 && l > 0) dl_grid_wait(a.bar, a.status, base, lane);             // every row of h (previous layer's phase 4)
      DL_STAMP(1);
      __syncthreads();
      if constexpr (!STREAM) if (hw == 0) xprep_rms(l == 0 ? a.h : nullptr, a.hx, lnw1);
      __syncthreads();
      DL_STAMP(2);
      if constexpr (!STREAM) if (hw == 3) {
        // norm weights through LDS: this layer's second one (read after two more barriers) and the NEXT layer's first one (its reader,
        // wave 0, is done with the current one: it was read before the barrier above)
#pragma unroll
        for (int i = 0; i < H / 512; ++i) {
          const u32x4 v2 = *(dl_gvec)(w.ln2 + (lane + 64 * i) * 8), v1 = *(dl_gvec)(wn.ln1 + (lane + 64 * i) * 8);
          lnw2[lane + 64 * i] = v2;
          lnw1[lane + 64 * i] = v1;
        }
      }
      if constexpr (STREAM) stream_phase(std::integral_constant<int, Gm::B0>{}, std::integral_constant<int, Gm::B2>{});
      __syncthreads();
      DL_STAMP(3);
      if constexpr (!STREAM) if (hw == 0) {
        static_assert(Gm::R0 % 2 == 0 && Gm::R2 % 2 == 0 && Gm::RG % 2 == 0 && Gm::R4 % 2 == 0 && Gm::RG <= 64, "rows leave in pairs, from helper wave 0");
        dl_st_pair(a.qkv + wg * Gm::R0, ht, Gm::R0, row_sum(ht < Gm::R0 ? ht : 0, std::integral_constant<int, H / 512>{}));
        dl_stores_done();
      }
      __syncthreads();
      DL_STAMP(4);
      if constexpr (!STREAM) if (lead) dl_grid_arrive(a.bar, wg);
      if constexpr (!STREAM) if (hw == 0) dl_grid_wait(a.bar, a.status, base + DL_G, lane);
      DL_STAMP(5);

      // ================= phase 1: attention of head wg (helpers of workgroups 0 .. HQ - 1), decode.hip's order of operations
      __syncthreads();
      DL_STAMP(20);
      if constexpr (!STREAM) if (attn && T > 128) load_rows(Kl, 128, kpre[1]);      // the second chunk of key rows: under the q / k / v preparation
      if constexpr (!STREAM) if (attn && hw < 3) {
        // wave 0: q' of head wg; wave 1: k' of its kv head; wave 2: v. Lane i holds elements i and i + 64 (the rotate_half pair).
        const int hk = wg / (HQ / HKV);
        const bf16_t* src = a.qkv + (hw == 0 ? wg * DL_D : (hw == 1 ? (HQ + hk) * DL_D : (HQ + HKV + hk) * DL_D));
        const unsigned w1 = dl_ld32(src + (lane & ~1)), w2 = dl_ld32(src + 64 + (lane & ~1));       // the pair holding element lane / lane + 64
        const float x1 = __builtin_bit_cast(float, (lane & 1) ? (w1 & 0xffff0000u) : (w1 << 16));
        const float x2 = __builtin_bit_cast(float, (lane & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
        float o1 = x1, o2 = x2;
        if (hw < 2) {
          const float rs = rsqrtf(wave_sum(x1 * x1 + x2 * x2) / (float)DL_D + a.eps);
          const float n1 = rbf(nw1 * rbf(x1 * rs));
          const float n2 = rbf(nw2 * rbf(x2 * rs));
          o1 = rbf(rbf(n1 * rc1) + rbf(-n2 * rs1));
          o2 = rbf(rbf(n2 * rc2) + rbf(n1 * rs2));
        }
        qs[hw * DL_D + lane] = o1;
        qs[hw * DL_D + lane + 64] = o2;
        if (hw > 0 && wg % (HQ / HKV) == 0) {                        // one workgroup per kv head appends the row to the cache
          bf16_t* dst = (hw == 1 ? Kl : Vl) + (long)pos * DL_D;
          dst[lane] = f2bf(o1);
          dst[lane + 64] = f2bf(o2);
        }
      }
      __syncthreads();
      DL_STAMP(21);
      float mx = -INFINITY;
      if constexpr (!STREAM) if (attn) {
        float qf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = qs[dl + e];
        // 128 keys per chunk and workgroup (8 row loads per lane), the next chunk requested before this one is multiplied; chunk 0 was
        // requested during phase 0
        auto scores = [&](int c0, const u32x4 (&kr)[8]) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lk = c0 + hw * 4 + 64 * (it >> 2) + 16 * (it & 3) + sub;
            float kf[8];
            dl_unpack8(kr[it], kf);
            if (lk == pos) {
#pragma unroll
              for (int e = 0; e < 8; ++e) kf[e] = qs[DL_D + dl + e];
            }
            float sv = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sv = fmaf(qf[e], kf[e], sv);
            sv = dl_row16_sum(sv) * a.scale;                             // the key's 16 lanes -> its lane 15 (DPP; a shuffle is an LDS round trip)
            if (lk < T && (lane & 15) == 15) {
              sc[lk] = sv;
              mx = fmaxf(mx, sv);
            }
          }
        };
        for (int c0 = 0; c0 < T; c0 += 256) {
          if (c0 > 0 && c0 + 128 < T) load_rows(Kl, c0 + 128, kpre[1]);
          scores(c0, kpre[0]);
          if (c0 + 256 < T) load_rows(Kl, c0 + 256, kpre[0]);
          if (c0 + 128 < T) scores(c0 + 128, kpre[1]);
        }
        asm volatile("" ::: "memory");                                // (the same registers: no value row is requested before the last key row was multiplied)
        load_rows(Vl, 0, kpre[0]);                                    // land under the two softmax steps
        if (T > 128) load_rows(Vl, 128, kpre[1]);
        mx = wave_max(mx);
        if (lane == 0) red[hw] = mx;
      }
      __syncthreads();
      DL_STAMP(22);
      float sum = 0.f;
      if constexpr (!STREAM) if (attn) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        for (int lk = ht; lk < T; lk += DL_TH) {
          const float e = __expf(sc[lk] - mx);
          sc[lk] = e;
          sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) red[4 + hw] = sum;
      }
      __syncthreads();
      DL_STAMP(23);
      if constexpr (!STREAM) if (attn) {
        const float inv = 1.f / (((0.f + red[4]) + red[5]) + red[6] + red[7]);
        float of[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] = 0.f;
        auto pv = [&](int c0, const u32x4 (&vr)[8]) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int lk = c0 + hw * 4 + 64 * (it >> 2) + 16 * (it & 3) + sub;
            const float p = lk < T ? rbf(sc[lk] * inv) : 0.f;
            float vf[8];
            dl_unpack8(vr[it], vf);
            if (lk == pos) {
#pragma unroll
              for (int e = 0; e < 8; ++e) vf[e] = qs[2 * DL_D + dl + e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) of[e] = fmaf(p, vf[e], of[e]);
          }
        };
        for (int c0 = 0; c0 < T; c0 += 256) {
          if (c0 > 0 && c0 + 128 < T) load_rows(Vl, c0 + 128, kpre[1]);
          pv(c0, kpre[0]);
          if (c0 + 256 < T) load_rows(Vl, c0 + 256, kpre[0]);
          if (c0 + 128 < T) pv(c0 + 128, kpre[1]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          of[e] += __shfl_xor(of[e], 16, 64);
          of[e] += __shfl_xor(of[e], 32, 64);
        }
        if (sub == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) opart[hw * DL_D + dl + e] = of[e];
        }
      }
      __syncthreads();
      DL_STAMP(24);
      if constexpr (!STREAM) if (attn && hw < 2) {
        dl_st_pair(a.ao + wg * DL_D, ht, DL_D, opart[ht] + opart[DL_D + ht] + opart[2 * DL_D + ht] + opart[3 * DL_D + ht]);
        dl_stores_done();
      }
      __syncthreads();
      DL_STAMP(25);
      DL_STAMP(6);
      if constexpr (!STREAM) if (lead && wg < HQ) dl_grid_arrive(a.bar, wg);
      if constexpr (!STREAM) if (hw == 0) dl_grid_wait(a.bar, a.status, base + DL_G + HQ, lane);
      DL_STAMP(7);

      // ================= phase 2: h_mid = h + W_o . attn
      __syncthreads();
      if constexpr (!STREAM) xprep_copy(a.ao, Gm::KO);
      __syncthreads();
      DL_STAMP(8);
      if constexpr (STREAM) stream_phase(std::integral_constant<int, Gm::B2>{}, std::integral_constant<int, Gm::B3>{});
      __syncthreads();
      DL_STAMP(9);
      if constexpr (!STREAM) if (hw == 0) {
        hmid_own = rbf(rbf(row_sum(ht < Gm::R2 ? ht : 0, std::integral_constant<int, Gm::KO / 512>{})) + h_own);      // (lanes past the rows: unused)
        dl_st_pair(a.hmid + wg * Gm::R2, ht, Gm::R2, hmid_own);
        dl_stores_done();
      }
      __syncthreads();
      DL_STAMP(10);
      if constexpr (!STREAM) if (lead) dl_grid_arrive(a.bar, wg);
      if constexpr (!STREAM) if (hw == 0) dl_grid_wait(a.bar, a.status, base + 2 * DL_G + HQ, lane);
      DL_STAMP(11);

      // ================= phase 3: act = SwiGLU(W_gate|up . RMSNorm(h_mid)) - a workgroup owns gate AND up of its 38 features
      __syncthreads();
      if constexpr (!STREAM) if (hw == 0) xprep_rms(nullptr, a.hmid, lnw2);
      __syncthreads();
      DL_STAMP(12);
      if constexpr (STREAM) stream_phase(std::integral_constant<int, Gm::B3>{}, std::integral_constant<int, Gm::B4>{});
      __syncthreads();
      DL_STAMP(13);
      if constexpr (!STREAM) if (hw == 0) {
        const int r = ht < Gm::RG ? ht : 0;
        const float g = rbf(row_sum(r, std::integral_constant<int, H / 512>{})), up = rbf(row_sum(Gm::RG + r, std::integral_constant<int, H / 512>{}));
        dl_st_pair(a.act + wg * Gm::RG, ht, Gm::RG, rbf(silu_f(g)) * up);
        dl_stores_done();
      }
      __syncthreads();
      DL_STAMP(14);
      if constexpr (!STREAM) if (lead) dl_grid_arrive(a.bar, wg);
      if constexpr (!STREAM) if (hw == 0) dl_grid_wait(a.bar, a.status, base + 3 * DL_G + HQ, lane);
      DL_STAMP(15);

      // ================= phase 4: h = h_mid + W_down . act
      __syncthreads();
      if constexpr (!STREAM) xprep_copy(a.act, I);
      __syncthreads();
      DL_STAMP(16);
      if constexpr (STREAM) stream_phase(std::integral_constant<int, Gm::B4>{}, std::integral_constant<int, Gm::NSLOT>{});
      __syncthreads();
      DL_STAMP(17);
      if constexpr (!STREAM) if (hw == 0) {
        h_own = rbf(rbf(row_sum(ht < Gm::R4 ? ht : 0, std::integral_constant<int, I / 512>{})) + hmid_own);
        dl_st_pair(a.hx + wg * Gm::R4, ht, Gm::R4, h_own);
        if (l == a.nl - 1 && ht < Gm::R4) a.h[wg * Gm::R4 + ht] = f2bf(h_own);
        dl_stores_done();
      }
      __syncthreads();
      DL_STAMP(18);
      if constexpr (!STREAM) if (lead) dl_grid_arrive(a.bar, wg);
      DL_STAMP(19);
#ifdef VQ3_DL_STAMPS
      if (wg == 0 && lead && l < 62) dl_layer_end[1 + l] = wall_clock64();
#endif
    }
  };
  if (streamer) run(std::true_type{});
  else run(std::false_type{});
}

using Qwen3_4B = Geo<2560, 9728, 32, 8>;
constexpr int DL_LDS_BYTES = 96 * 1024;      // more than half a CU's 160 KiB: one workgroup per CU, whatever the register count
static_assert(Qwen3_4B::LDS_USED <= DL_LDS_BYTES, "LDS layout fits the request");

bool dl_device_ok() {
  static int ok = -1;
  if (ok < 0) {
    int dev = 0, cus = 0;
    ok = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= DL_G) {
      auto k = decode_layers_kernel<2560, 9728, 32, 8>;
      int nb = 0;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS_BYTES) == hipSuccess &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, DL_TS + DL_TH, DL_LDS_BYTES) == hipSuccess && nb >= 1)
        ok = 1;
    }
    (void)hipGetLastError();
  }
  return ok == 1;
}

}  // namespace

extern "C" int vq3_qwen_decode_layers_supported(int32_t hidden, int32_t intermediate, int32_t Hq, int32_t Hkv, int32_t head_dim,
                                                int32_t Lmax) {
  if (hidden != 2560 || intermediate != 9728 || Hq != 32 || Hkv != 8 || head_dim != DL_D) return 0;
  if (Lmax < 1 || Lmax > DL_LMAX) return 0;
  return dl_device_ok() ? 1 : 0;
}

extern "C" int64_t vq3_qwen_decode_layers_workspace_bytes(void) {
  return (int64_t)sizeof(bf16_t) * (Qwen3_4B::NQKV + Qwen3_4B::KO + 2 * 2560 + 9728);
}

extern "C" int vq3_qwen_decode_layers(const vq3_decode_layers_desc* d, void* stream) {
  VQ3_CHECK_ARG(d, "decode_layers: null descriptor");
  VQ3_CHECK_ARG(d->weights && d->h && d->workspace && d->cos && d->sin && d->lens && d->Kcache && d->Vcache && d->barrier && d->status,
                "decode_layers: null pointer");
  VQ3_CHECK_ARG((uintptr_t)d->workspace % 16 == 0 && (uintptr_t)d->h % 16 == 0, "decode_layers: h and workspace must be 16-byte aligned");
  VQ3_CHECK_ARG(d->layers >= 1, "decode_layers: layers must be >= 1, got %d", d->layers);
  VQ3_CHECK_ARG(vq3_qwen_decode_layers_supported(d->hidden, d->intermediate, d->Hq, d->Hkv, d->head_dim, d->Lmax),
                "decode_layers: unsupported shape / device (hidden %d, intermediate %d, heads %d / %d x %d, Lmax %d; needs Qwen3-4B's "
                "2560 / 9728 / 32 / 8 x 128, Lmax <= %d and %d CUs) - use the per-projection launches",
                d->hidden, d->intermediate, d->Hq, d->Hkv, d->head_dim, d->Lmax, DL_LMAX, DL_G);
  VQ3_CHECK_ARG(d->cache_layer_stride >= (int64_t)d->Hkv * d->Lmax * DL_D, "decode_layers: cache_layer_stride smaller than one layer's cache");
  Args a;
  a.w = reinterpret_cast<const LayerW*>(d->weights);
  a.h = (bf16_t*)d->h;
  a.qkv = (bf16_t*)d->workspace; a.ao = a.qkv + Qwen3_4B::NQKV; a.hmid = a.ao + Qwen3_4B::KO; a.act = a.hmid + 2560; a.hx = a.act + 9728;
  a.cs = (const bf16_t*)d->cos; a.sn = (const bf16_t*)d->sin; a.lens = d->lens;
  a.Kc = (bf16_t*)d->Kcache; a.Vc = (bf16_t*)d->Vcache; a.cache_stride = d->cache_layer_stride;
  a.bar = (unsigned*)d->barrier; a.status = (unsigned*)d->status;
  a.nl = d->layers; a.Lmax = d->Lmax; a.eps = d->eps; a.scale = d->scale;
  hipLaunchKernelGGL((decode_layers_kernel<2560, 9728, 32, 8>), dim3(DL_G), dim3(DL_TS + DL_TH), DL_LDS_BYTES, (hipStream_t)stream, a);
  VQ3_CHECK_LAUNCH("decode_layers");
  return 0;
}
