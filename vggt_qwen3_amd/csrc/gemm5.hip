// Stream-K variant of the NT loader kernel (gemm2.hip, 128x128 tile, 8 compute + 2 DMA-loader waves, 4-stage ring) for
// shapes whose tile count does not fill the chip: the four [1200, 2560]-output GEMMs of a Qwen3-4B layer have 200 tiles
// for 256 CUs (78 %). OPT-IN (VQ3_GEMM_STREAMK=1): a new synchronisation structure wants a long race screen before it
// may carry the headline path.
//
// Decomposition (K kept aligned): with T tiles and G = #CUs workgroups, workgroup w < T owns tile w and computes its first
// ka = nk * T / G K steps - all tile owners walk K in lock step, so the A / W panels they share are reused out of L2 exactly
// as in the per-tile kernel; the G - T helper workgroups compute the remaining nk - ka steps of T / (G - T) tiles each (the
// same amount of work) and hand every tail over as an f32 partial in the tile's workspace slot, each compute wave
// releasing its own flag (epoch value, so nothing is ever reset). The owner acquires wave by wave, adds and runs the
// normal epilogue: no atomics on C, no cross-wave barrier in the fix-up, and no circular wait (helpers never wait), so
// residency of all G workgroups is not required. (First form tried: contiguous runs of the tile-major K-step sequence per
// workgroup - correct, but workgroups starting mid-tile walk K out of step, the L2 reuse is lost and it ran 1.4-1.8x
// slower than the per-tile kernel.)
#include "gemm_common.h"
#include "vq3_hip.h"

namespace vq3gemm {
namespace {

constexpr int BK = 64;
constexpr int BM = 128, BN = 128, WM = 4, WN = 2, NSTAGE = 4, NLOAD = 2;
constexpr int NW = WM * WN;
constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
constexpr int STAGE = (BM + BN) * 128;
constexpr int NPIECE = (BM + BN) / 8;
constexpr int PPW = NPIECE / NLOAD;
constexpr int WS_FLOATS = BM * BN;                 // one partial tile per workgroup
constexpr int SMEM = NSTAGE * STAGE;

struct SkArgs {
  float* ws;          // [G][BM*BN] f32 partials
  int* flags;         // [G][NW] epoch flags
  int epoch;
  int ka_permille;    // owner's share of K in 1/1000 (0 = balanced T/G)
};

__global__ __launch_bounds__(64 * (NW + NLOAD), (NW + NLOAD + 3) / 4) void gemm_sk_kernel(GemmParams p, SkArgs sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= NW;
  const int iw = loader ? wid - NW : 0;
  const int wm = (wid % NW) / WN, wn = wid % WN;
  const int nk = p.K / BK;
  const int G = gridDim.x, w = blockIdx.x;    // (giving each XCD one contiguous stretch of the sequence measured slower)
  const int T = p.mtiles * p.ntiles;                     // T < G (launcher)
  const int Hn = G - T;
  int ka = sk.ka_permille > 0 ? (int)((long)nk * sk.ka_permille / 1000) : (int)(((long)nk * T + G / 2) / G);
  ka = ka < 1 ? 1 : (ka > nk - 1 ? nk - 1 : ka);
  const bool owner = w < T;
  const int t_first = owner ? w : (int)((long)T * (w - T) / Hn);
  const int t_last = owner ? w + 1 : (int)((long)T * (w - T + 1) / Hn);
  const int fr = lane & 15, fq = lane >> 4;

  int a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * (BM / WM) + i * 16 + fr;
    a_off[i] = row * 128 + ((fq ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * (BN / WN) + j * 16 + fr;
    b_off[j] = BM * 128 + row * 128 + ((fq ^ (row & 7)) << 4);
  }

  for (int tile = t_first; tile < t_last; ++tile) {
    const int k0 = owner ? 0 : ka, k1 = owner ? ka : nk;
    const int nt = k1 - k0, last = nt - 1;
    int m0, n0;                       // tile id = its owner's block id: the per-tile kernels' XCD-aware order applies as is
    tile_coords_id(p, tile, BM, BN, m0, n0);

    if (loader) {
      const int prow = lane >> 3;
      const int kch = (lane & 7) ^ prow;
      const bf16_t* gsrc[PPW];
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int pi = iw + NLOAD * j;
        const int trow = pi * 8 + prow;
        if (trow < BM) {
          int r = m0 + trow; r = r < p.M ? r : p.M - 1;
          gsrc[j] = p.A + (long)r * p.lda + kch * 8 + (long)k0 * BK;
        } else {
          int r = n0 + (trow - BM); r = r < p.N ? r : p.N - 1;
          gsrc[j] = p.B + (long)r * p.ldb + kch * 8 + (long)k0 * BK;
        }
      }
      auto issue = [&](int t, int stage) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          const int pi = iw + NLOAD * j;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (long)t * BK),
                                           (__attribute__((address_space(3))) void*)(sb + pi * 1024), 16, 0, 0);
        }
      };
      // every compute wave has left the previous piece's last stage (piece barrier below) before the ring is refilled
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < NSTAGE - 1; ++i)
        if (i < nt) issue(i, i);
      int stage = 0;
      for (int t = 0; t < nt; ++t) {
        const int newer = (last - t) < (NSTAGE - 2) ? (last - t) : (NSTAGE - 2);
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 <= last) {
          int s2 = stage + NSTAGE - 1; s2 = s2 >= NSTAGE ? s2 - NSTAGE : s2;
          issue(t + NSTAGE - 1, s2);
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
      }
      continue;
    }

    // ---------------- compute waves
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();            // piece barrier (pairs with the loaders' one above)
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* sb = smem + stage * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xa[TM], wb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(sb + (a_off[i] ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(sb + (b_off[j] ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
      }
      stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }

    if (k0 > 0) {
      // not the start of its tile: leave the partial for the head's owner. Fragment-major layout: the reader is the
      // same wave id / lane of another workgroup, so both sides move 16 contiguous bytes per lane.
      // Payload goes out WRITE-THROUGH (sc0 sc1: past this XCD's L2), the wave drains its own stores, then signals with a
      // relaxed agent-scope store - no release fence (a fence here writes back the whole L2: measured 150 us per launch).
      float* slot = sk.ws + (long)tile * WS_FLOATS + ((long)wid * TM * TN * 64 + lane) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float* dst = slot + (long)(i * TN + j) * 256;
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(acc[i][j]) : "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(sk.flags + (long)tile * NW + wid, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    {
      // the tile's tail comes from a helper workgroup, which has nothing else to wait for
      const int* f = sk.flags + (long)tile * NW + wid;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(4);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE invalidate after the match, then plain loads
      const float* slot = sk.ws + (long)tile * WS_FLOATS + ((long)wid * TM * TN * 64 + lane) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(slot + (long)(i * TN + j) * 256);
          acc[i][j] += v;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (BM / WM) + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 16 + 4 * fq;
        if (n >= p.N) continue;
        store_quad<false>(p, 0, 0, m, n, acc[i][j]);
      }
    }
  }
}

float* g_ws = nullptr;
int* g_flags = nullptr;
long g_ws_floats = 0;
int g_epoch = 0;
hipStream_t g_stream = nullptr;      // the one stream whose launches may use the workspace (launches on it are ordered)
bool g_stream_set = false;

}  // namespace

// 0 = launched; 1 = not applicable (caller falls back to the per-tile kernels)
int launch_gemm_streamk(GemmParams& p, int ncu, hipStream_t stream) {
  if (!g_ws || p.out_f32 || p.nsplit != 1 || p.K % BK != 0) return 1;
  if (!g_stream_set) { g_stream = stream; g_stream_set = true; }
  if (stream != g_stream) return 1;          // a second stream could overlap two users of the one workspace
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  const long tiles = (long)p.mtiles * p.ntiles;
  p.xm = choose_xm(p.mtiles, p.ntiles);
  const int G = ncu;
  // only the under-filled single-round case, with enough K per piece to amortise a pipeline fill
  if (tiles * 10 < (long)G * 6 || tiles * 10 > (long)G * 9 || p.K / BK < 16 || (long)G * (WS_FLOATS + NW) > g_ws_floats) return 1;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 1;
    attr_done = true;
  }
  static int ka_pm = -1;
  if (ka_pm < 0) {
    const char* e = getenv("VQ3_STREAMK_KA");     // owner's share of K in 1/1000; helpers pay a pipeline fill per tail
    ka_pm = e ? atoi(e) : 0;
  }
  SkArgs sk{g_ws, g_flags, ++g_epoch, ka_pm};
  hipLaunchKernelGGL(gemm_sk_kernel, dim3(G), dim3(64 * (NW + NLOAD)), SMEM, stream, p, sk);
  return 0;
}

}  // namespace vq3gemm

// Workspace for the stream-K GEMM: `bytes` of zero-initialised device memory owned by the caller (>= n_cu * (64 KiB + 32 B)).
// Passing NULL disables the variant again.
extern "C" int vq3_gemm_set_workspace(void* ptr, int64_t bytes) {
  using namespace vq3gemm;
  g_stream_set = false;
  if (!ptr || bytes <= 0) {
    g_ws = nullptr; g_flags = nullptr; g_ws_floats = 0;
    return 0;
  }
  VQ3_CHECK_ARG((uintptr_t)ptr % 16 == 0 && bytes % 4 == 0, "gemm_set_workspace: need a 16-byte aligned buffer");
  const long floats = bytes / 4;
  // layout: [flags: floats/ (WS_FLOATS + NW) * NW ints][partials]; keep it simple: flags first, 4 KiB-aligned partials after
  const long slots = floats / (WS_FLOATS + NW + 1);
  VQ3_CHECK_ARG(slots >= 1, "gemm_set_workspace: buffer too small");
  g_flags = (int*)ptr;
  long flag_ints = (slots * NW + 1023) / 1024 * 1024;
  g_ws = (float*)ptr + flag_ints;
  g_ws_floats = (floats - flag_ints) / WS_FLOATS * (WS_FLOATS + NW);     // capacity expressed the way the launcher checks it
  return 0;
}
