// GEMM v8 for gfx950: role-specialised persistent kernel whose ROW-WISE EPILOGUE RUNS UNDER THE NEXT TILE'S MAIN LOOP.
//
// Why: at K = 1024 (the VGGT tower: 16 K tiles per output tile) the 8-phase kernels (gemm6.hip) spend 25-45 % of a tile's time in
// their epilogue with the MFMA pipe idle - staging the tile through LDS, the activation / q|k|v LayerNorm + RoPE arithmetic, the row
// stores (and, with a residual, a read burst every CU issues at the same moment): tools/gemm_stamps.py, 49 392 x 4096 x 1024 with
// GELU: main loop 27 us, epilogue 15 us. One workgroup per CU owns all 160 KiB of LDS there, so nothing else can run meanwhile.
//
// Here a workgroup is two teams of four waves (one wave of each team per SIMD):
//   * COMPUTE waves 0-3 own the 128 x 256 tile as 64 x 128 per wave (128 accumulator registers): ds_read_b128 fragments + MFMA only,
//     plus a small share of the DMA issue. At the end of a tile they apply alpha / folded LayerNorm / bias in f32, round to bf16 and lay
//     the tile out in a 64 KiB LDS image - 0.5 us - and go straight on with the next tile.
//   * SERVICE waves 4-7 issue most of the LDS-DMA (global_load_lds_dwordx4) and, one 8-row pass per K tile, take the PREVIOUS tile's
//     image out of LDS: activation, LayerScale, residual / accumulate, output LayerNorm statistics or the fused q|k|v epilogue (per-head
//     LayerNorm + 2-D RoPE), 16-byte row stores. Their VALU and memory instructions overlap the compute team's MFMAs on every SIMD.
// The K tiles of consecutive output tiles form ONE stream through a two-buffer ring (2 x 48 KiB, one barrier per K tile, tile t+1 in
// flight while tile t is multiplied), so a tile never starts with an empty pipeline.
//
// Same contract as the other NT kernels (C = epilogue(alpha * A[M,K] . B[N,K]^T), bf16 in, f32 MFMA accumulate, K % 64 == 0), bf16
// output through the staged path only (host_staged_ok), N % 8 == 0, one batch. Arithmetic and rounding points are those of
// gemm_common.h: stage_quad + staged_store / vit_qkv_store (bit-identical results).
//
// Memory-operation counting (cdna guide 5.7): the service waves' waits are COUNTED (the DMA of the next K tile must have landed, the
// stores and operand prefetches issued after it may still be in flight), so every global access of the row pass is a raw buffer
// load / store that the wave ALWAYS issues - lanes without work point past the descriptor's range and are dropped by the range check.
#include "gemm_common.h"

namespace vq3gemm {
namespace {

constexpr int V8_BM = 128, V8_BN = 256, V8_BK = 64;
constexpr int V8_ABYTES = V8_BM * 128, V8_BBYTES = V8_BN * 128, V8_BUF = V8_ABYTES + V8_BBYTES;   // 16 + 32 = 48 KiB per K tile
constexpr int V8_IMG = V8_BM * V8_BN * 2;                                                       // 64 KiB bf16 C image
constexpr int V8_SMEM = 2 * V8_BUF + V8_IMG;                                                    // 160 KiB
constexpr int V8_NPASS = V8_BM / 8;                                                             // 8 image rows per pass (256 threads x 16 B)

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}
__device__ __forceinline__ void buf_store16(rsrc_t rs, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_store8(rsrc_t rs, unsigned off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, 0, 0);
}
constexpr unsigned OOB = 0xFFFFFFF0u;     // past every descriptor's range: the access is dropped / reads zero

__device__ __forceinline__ void wait_vm(int n) {          // counted wait with a wave-uniform run-time count (0..3)
  if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
}

// EK: 0 = plain C epilogue, 3 = plain behind a folded LayerNorm, 1 = fused q|k|v epilogue (LayerNorm fold optional)
template <int EK>
__global__ __launch_bounds__(512, 2) void gemm_v8_kernel(GemmParams p) {
  constexpr bool HAS_LN = (EK == 1 || EK == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const img = smem + 2 * V8_BUF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool compute = wid < 4;
  const int ntile = p.mtiles * p.ntiles;
  const int nt = p.K / V8_BK;
  const bf16_t* A = p.A;
  const bf16_t* B = p.B;

  // ---- DMA pieces (1 KiB = 8 rows x 128 B of the stacked [A rows | B rows] image): service wave s issues pieces 8 s .. 8 s + 7,
  // compute wave c pieces 32 + 4 c .. + 3. Lane l of a piece reads k-chunk (l & 7) ^ (l >> 3) of row l >> 3 (read-side swizzle on the source).
  constexpr int NP = 8;
  const int np = compute ? 4 : 8;
  const int p0 = compute ? 32 + 4 * wid : 8 * (wid - 4);
  const int prow = lane >> 3, kch = (lane & 7) ^ prow;
  unsigned poff[NP];
  auto set_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = p0 + (j < np ? j : 0);
      const int r = q * 8 + prow;                         // row of the stacked tile
      if (q < 16) {
        int ra = m0 + r; ra = ra < p.M ? ra : p.M - 1;
        poff[j] = (unsigned)(((long)ra * p.lda + kch * 8) * 2);
      } else {
        int rb = n0 + (r - V8_BM); rb = rb < p.N ? rb : p.N - 1;
        poff[j] = (unsigned)(((long)rb * p.ldb + kch * 8) * 2);
      }
    }
  };
  auto issue = [&](int ktile, int buf) {
    char* dst = smem + buf * V8_BUF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (j < np) {
        const int q = p0 + j;
        const char* g = reinterpret_cast<const char*>(q < 16 ? A : B) + (long)ktile * (V8_BK * 2) + poff[j];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      }
    }
  };

  int tile = blockIdx.x;                  // (< ntile: the grid never exceeds the tile count)
  int m0, n0;
  tile_coords_id(p, tile, V8_BM, V8_BN, m0, n0);
  set_offsets(m0, n0);
  issue(0, 0);
  int g = 0;                              // K tiles consumed so far by this workgroup: ring buffer = g & 1
  const int fr = lane & 15, fq = lane >> 4;

  // The two roles are two separate loops over the same (tile, K tile) sequence with the same barriers - one s_barrier per K tile, two
  // (E1, E2) at every tile hand-over - so that neither role's registers are live in the other's code (128 accumulators + fragments
  // on one side, the row pass's operands and column constants on the other).
  if (compute) {
    // ================================================================================================= COMPUTE TEAM
    // wave (wr, wc) of 2 x 2 owns rows wr * 64 .. + 63, columns wc * 128 .. + 127
    const int wr = (wid >> 1) & 1, wc = wid & 1;
    const int a_base = (wr * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    const int b_base = V8_ABYTES + (wc * 128 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
    f32x4 acc[4][8];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    float ln_mu = 0.f, ln_rs = 1.f;                     // lane (fr, fq): statistics of tile row wr * 64 + fq * 16 + fr
    while (true) {
      const int next_tile = tile + (int)gridDim.x;
      const bool has_next = next_tile < ntile;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (HAS_LN && p.ln_in) ln_row(p, m0 + wr * 64 + fq * 16 + fr, ln_mu, ln_rs);
      for (int t = 0; t < nt; ++t, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of K tile g has landed
        __builtin_amdgcn_s_barrier();          // K tile g is complete in LDS; everybody has finished reading the other buffer
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // request the next K tile of the stream (this tile's t + 1, or K tile 0 of the workgroup's next output tile)
        if (t + 1 < nt) {
          issue(t + 1, (g + 1) & 1);
        } else if (has_next) {
          int nm0, nn0;
          tile_coords_id(p, next_tile, V8_BM, V8_BN, nm0, nn0);
          set_offsets(nm0, nn0);
          issue(0, (g + 1) & 1);
        }
        const char* buf = smem + (g & 1) * V8_BUF;
        // A fragments of both k halves stay in registers for the K tile; the B fragments stream through two register pairs: column
        // group j + 1 is read while group j multiplies, so only the first reads of a K tile are exposed (reading all 12 fragments of
        // a k half before its 32 MFMAs left the LDS latency in the open twice per K tile: 23 us per 128 x 256 x 1024 tile).
        // The reads are inline asm with hand-counted waits (cdna guide 5.7 item 1, form ii: every wait names the registers it releases):
        // hipcc sinks compiler-visible reads to just before their first use and waits lgkmcnt(0) there.
        const unsigned la0 = lds0 + (unsigned)((g & 1) * V8_BUF) + (unsigned)a_base, la1 = lds0 + (unsigned)((g & 1) * V8_BUF) + (unsigned)(a_base ^ 64);
        const unsigned lb0 = lds0 + (unsigned)((g & 1) * V8_BUF) + (unsigned)b_base, lb1 = lds0 + (unsigned)((g & 1) * V8_BUF) + (unsigned)(b_base ^ 64);
        bf16x8 xa0[4], xa1[4], w0[2], w1[2];
#define V8_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#pragma unroll
        for (int i = 0; i < 4; ++i) { V8_RD(xa0[i], la0, i * 2048); V8_RD(xa1[i], la1, i * 2048); }
        V8_RD(w0[0], lb0, 0); V8_RD(w1[0], lb1, 0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j + 1 < 8) { V8_RD(w0[(j + 1) & 1], lb0, (j + 1) * 2048); V8_RD(w1[(j + 1) & 1], lb1, (j + 1) * 2048); }
          if (j == 0) {
            asm volatile("s_waitcnt lgkmcnt(2)"
                         : "+v"(xa0[0]), "+v"(xa0[1]), "+v"(xa0[2]), "+v"(xa0[3]), "+v"(xa1[0]), "+v"(xa1[1]), "+v"(xa1[2]), "+v"(xa1[3]),
                           "+v"(w0[0]), "+v"(w1[0]));
          } else if (j + 1 < 8) {
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0[j & 1]), "+v"(w1[j & 1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0[j & 1]), "+v"(w1[j & 1]));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j & 1], xa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j & 1], xa1[i], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef V8_RD
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- hand-over. E1: the service team has taken the previous image out; E2: the new image is complete
      float mu4[4], rs4[4];
      if (HAS_LN && p.ln_in) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {      // row mt * 16 + fr belongs to lane (fr, mt)
          mu4[mt] = __shfl(ln_mu, mt * 16 + fr, 64);
          rs4[mt] = __shfl(ln_rs, mt * 16 + fr, 64);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();             // E1
      asm volatile("" ::: "memory");
      // alpha / folded LayerNorm / bias in f32, rounded to bf16 -> C image (stage_quad's rounding points up to the activation)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int n = n0 + wc * 128 + j * 16 + 4 * fq;
        n = n + 3 < p.N ? n : (p.N >= 4 ? p.N - 4 : 0);          // columns past N are never stored
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (HAS_LN && p.ln_in) cv = *reinterpret_cast<const f32x4*>(p.ln_c + n);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 a = acc[mt][j];
          if (HAS_LN && p.ln_in) a = ln_apply(a, mu4[mt], rs4[mt], cv);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
          if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
          }
          u32x2 o;
          o[0] = pack2bf(v[0], v[1]);
          o[1] = pack2bf(v[2], v[3]);
          const int ml = wr * 64 + mt * 16 + fr, nl = wc * 128 + j * 16 + 4 * fq;
          *reinterpret_cast<u32x2*>(img + cstage_off<V8_BN>(ml, nl >> 3) + ((nl & 4) << 1)) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();             // E2
      asm volatile("" ::: "memory");
      if (!has_next) break;
      tile = next_tile;
      tile_coords_id(p, tile, V8_BM, V8_BN, m0, n0);
    }
    return;
  }

  // ===================================================================================================== SERVICE TEAM
  // thread owns 16-byte chunk sc of image row 8 * pass + srow
  const int stid = tid - 256;
  const int srow = stid >> 5, sc = stid & 31;
  // descriptors of the row pass (range-checked: masked lanes use OOB)
  const rsrc_t rsC = make_rsrc(p.C, (unsigned)(((long)(p.M - 1) * p.ldc + p.N) * 2));
  const rsrc_t rsR = make_rsrc(p.R ? p.R : p.C, p.R ? (unsigned)(((long)(p.M - 1) * p.ldr + p.N) * 2) : 0u);
  const rsrc_t rsS = make_rsrc(p.st_out ? (const void*)p.st_out : (const void*)p.C, p.st_out ? (unsigned)((long)p.M * (p.N >> 7) * 8) : 0u);
  int pend_m0 = 0, pend_n0 = 0, pend_pass = V8_NPASS;      // the image still being taken out (V8_NPASS = none)
  u32x4 pre_a = {0u, 0u, 0u, 0u}, pre_b = {0u, 0u, 0u, 0u};   // operands of the NEXT pass: residual | old C  (EK 1: cos | sin rows)
  int vm_since = 0;                                         // memory operations issued since this wave's last DMA piece
  const VitQkvEpi& e = p.vit;
  float col8[16];                    // EK 1: LayerNorm weight | bias of the thread's 8 features; else [0..7] = LayerScale of its 8 columns
  int v_head = 0, v_d0 = 0;
  bool v_norm = false, v_rope = false, v_neg = false;
  float invN = 0.f, invP = 0.f, invW = 0.f;
  rsrc_t rsQ = rsC, rsCos = rsC, rsSin = rsC;
  int v_gi = 0, v_ti = 0;            // EK 1: (group, token) of the NEXT pass's row (computed with its prefetch)
  if (EK == 1) {
    invN = 1.f / (float)e.N; invP = 1.f / (float)(e.P > 0 ? e.P : 1); invW = 1.f / (float)(e.Wp > 0 ? e.Wp : 1);
    // (cos / sin tables [maxpos + 1, 32] bf16: small; 2^20 bytes bound them - offsets are computed from valid positions only)
    rsCos = make_rsrc(e.cos ? (const void*)e.cos : (const void*)p.C, e.cos ? (1u << 20) : 0u);
    rsSin = make_rsrc(e.sin ? (const void*)e.sin : (const void*)p.C, e.sin ? (1u << 20) : 0u);
  }
  auto tile_setup = [&](int tn0) {      // per tile: column constants of this thread's 8 columns
    if (EK != 1) {
      if (p.colscale) {
        const int n = tn0 + sc * 8;
        const int nc = n + 7 < p.N ? n : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) col8[j] = p.colscale[nc + j];
      }
      return;
    }
    const int Cw = e.NH * 64;                     // (host: Cw % 256 == 0, so a tile lies inside one of q / k / v: `which` is wave-uniform)
    const int which = __builtin_amdgcn_readfirstlane(tn0 / Cw);
    const int n = tn0 + sc * 8;
    const int rem = (n < p.N ? n : tn0) - which * Cw;
    v_head = rem >> 6; v_d0 = rem & 63;
    v_norm = which < 2 && e.use_norm; v_rope = which < 2 && e.use_rope;
    v_neg = (v_d0 & 16) == 0;
    if (v_norm) {
      const float* wp = (which == 0 ? e.qn_w : e.kn_w) + v_d0;
      const float* bp = (which == 0 ? e.qn_b : e.kn_b) + v_d0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { col8[j] = wp[j]; col8[8 + j] = bp[j]; }
    }
    const bf16_t* base = which == 0 ? e.Q : (which == 1 ? e.K : e.V);
    rsQ = make_rsrc(base, (unsigned)((long)(p.M + e.m_off) * Cw * 2));
  };

  // operands of pass `ps` of the pending image (issued one pass ahead); returns the number of loads issued (wave-uniform)
  auto prefetch_pass = [&](int ps) -> int {
    const int row = 8 * ps + srow, m = pend_m0 + row, n = pend_n0 + sc * 8;
    const bool ok = ps < V8_NPASS && m < p.M && n < p.N;
    if (EK == 1) {
      divmod_f((m < p.M ? m : p.M - 1) + e.m_off, e.N, invN, v_gi, v_ti);
      if (!e.use_rope) return 0;
      unsigned off = OOB;
      if (ok && v_rope) {
        int frm, tp = v_ti;
        if (e.P != e.N) divmod_f(v_ti, e.P, invP, frm, tp);
        int py = 0, px = 0;
        if (tp >= e.patch_start) { divmod_f(tp - e.patch_start, e.Wp, invW, py, px); ++py; ++px; }
        const int pos = v_d0 < 32 ? py : px;
        off = (unsigned)((pos * 32 + (v_d0 & 31)) * 2);
      }
      pre_a = buf_load16(rsCos, off);
      pre_b = buf_load16(rsSin, off);
      return 2;
    }
    int nl = 0;
    if (p.R) { pre_a = buf_load16(rsR, ok ? (unsigned)(((long)m * p.ldr + n) * 2) : OOB); ++nl; }
    if (p.accumulate) { pre_b = buf_load16(rsC, ok ? (unsigned)(((long)m * p.ldc + n) * 2) : OOB); ++nl; }
    return nl;
  };

  // one pass of the pending image: 8 rows, this thread's 16-byte chunk. Returns the number of stores issued (wave-uniform).
  auto row_pass = [&](int ps) -> int {
    const int row = 8 * ps + srow, m = pend_m0 + row, n = pend_n0 + sc * 8;
    const bool ok = m < p.M && n < p.N;
    const u32x4 sv = *reinterpret_cast<const u32x4*>(img + cstage_off<V8_BN>(row, sc));
    float x[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[2 * k] = bf2f((bf16_t)(sv[k] & 0xffff)); x[2 * k + 1] = bf2f((bf16_t)(sv[k] >> 16)); }
    if (EK == 1) {
      // arithmetic of gemm_common.h: vit_qkv_store (itself that of vit_qkprep4_kernel)
      if (v_norm) {
        const float mean = sum8(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 64.f);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] -= mean; sq = fmaf(x[j], x[j], sq); }
        const float rs = rsqrtf(sum8(sq) * (1.f / 64.f) + e.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = rbf(x[j] * rs * col8[j] + col8[8 + j]);
      }
      if (v_rope) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cj = bf2f((bf16_t)((pre_a[j >> 1] >> ((j & 1) * 16)) & 0xffff));
          const float sj = bf2f((bf16_t)((pre_b[j >> 1] >> ((j & 1) * 16)) & 0xffff));
          const float pj = dpp_mov<0x4E>(x[j]);                                                  // feature e ^ 16: lane ^ 2
          const float rj = v_neg ? -pj : pj;
          x[j] = rbf(rbf(x[j] * cj) + rbf(rj * sj));
        }
      }
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pack2bf(x[2 * k], x[2 * k + 1]);
      buf_store16(rsQ, ok ? (unsigned)(((((long)v_gi * e.NH + v_head) * e.N + v_ti) * 64 + v_d0) * 2) : OOB, o);
      return 1;
    }
    // plain: activation -> LayerScale -> residual -> accumulate (rounding points of stage_quad / staged_store)
    if (p.act == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2_t gq = gelu_erf2(f32x2_t{x[2 * k], x[2 * k + 1]});
        x[2 * k] = rbf(gq[0]); x[2 * k + 1] = rbf(gq[1]);
      }
    } else if (p.act) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = rbf(apply_act(x[j], p.act));
    }
    if (p.colscale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = rbf(x[j] * col8[j]);
    }
    if (p.R) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x[2 * k] = rbf(x[2 * k] + bf2f((bf16_t)(pre_a[k] & 0xffff)));
        x[2 * k + 1] = rbf(x[2 * k + 1] + bf2f((bf16_t)(pre_a[k] >> 16)));
      }
    }
    if (p.accumulate) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { x[2 * k] += bf2f((bf16_t)(pre_b[k] & 0xffff)); x[2 * k + 1] += bf2f((bf16_t)(pre_b[k] >> 16)); }
    }
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(x[2 * k], x[2 * k + 1]);
    buf_store16(rsC, ok ? (unsigned)(((long)m * p.ldc + n) * 2) : OOB, o);
    int ns = 1;
    if (p.st_out) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = bf2f((bf16_t)(o[k] & 0xffff)), hi = bf2f((bf16_t)(o[k] >> 16));
        sm += lo + hi;
        sq = fmaf(lo, lo, fmaf(hi, hi, sq));
      }
      sm = sum16(sm); sq = sum16(sq);
      u32x2 pr;
      pr[0] = __float_as_uint(sm); pr[1] = __float_as_uint(sq);
      buf_store8(rsS, (ok && (sc & 15) == 0) ? (unsigned)((((long)m * (p.N >> 7) + (n >> 7)) * 2) * 4) : OOB, pr);
      ++ns;
    }
    return ns;
  };

  // passes per K tile so that an image is out before the next one is written: 16 passes over nt K tiles
  const int ppk = (V8_NPASS + nt - 1) / nt;
  const int nl_pass = (EK == 1) ? (e.use_rope ? 2 : 0) : ((p.R ? 1 : 0) + (p.accumulate ? 1 : 0));   // operand loads per pass
  bool first_step = true;
  while (true) {
    const int next_tile = tile + (int)gridDim.x;
    const bool has_next = next_tile < ntile;
    for (int t = 0; t < nt; ++t, ++g) {
      // this wave's share of K tile g has landed (the row pass's own stores / prefetches issued after it may stay in flight:
      // a smaller count than the truth only waits longer)
      if (first_step) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else wait_vm(vm_since > 3 ? 3 : vm_since);
      first_step = false;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bool issued = false;
      if (t + 1 < nt) {
        issue(t + 1, (g + 1) & 1);
        issued = true;
      } else if (has_next) {
        int nm0, nn0;
        tile_coords_id(p, next_tile, V8_BM, V8_BN, nm0, nn0);
        set_offsets(nm0, nn0);
        issue(0, (g + 1) & 1);
        issued = true;
      }
      if (issued) vm_since = 0;      // (what was outstanding at the step-top wait stays counted otherwise)
      for (int k = 0; k < ppk; ++k) {
        if (pend_pass < V8_NPASS) {
          // the operands of this pass were requested a pass ago. First pass of a step: they are older than the 8 DMA pieces just
          // issued, so a count of 8 leaves only those in flight; otherwise wait for everything.
          if (nl_pass) {
            if (k == 0 && issued) {
              asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              vm_since = 0;
            }
          }
          vm_since += row_pass(pend_pass);
          ++pend_pass;
          vm_since += prefetch_pass(pend_pass);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- hand-over
    while (pend_pass < V8_NPASS) {        // (nt * ppk >= 16 always: kept as a guard)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      vm_since = 0;
      vm_since += row_pass(pend_pass);
      ++pend_pass;
      vm_since += prefetch_pass(pend_pass);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // E1: the previous image is out
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();             // E2: the new image is complete
    asm volatile("" ::: "memory");
    pend_m0 = m0; pend_n0 = n0; pend_pass = 0;
    tile_setup(n0);
    vm_since += prefetch_pass(0);             // (issued after the DMA of the next step, requested in the last K step above)
    if (!has_next) break;
    tile = next_tile;
    tile_coords_id(p, tile, V8_BM, V8_BN, m0, n0);
  }
  // ---- drain: the last image
  while (pend_pass < V8_NPASS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    row_pass(pend_pass);
    ++pend_pass;
    prefetch_pass(pend_pass);
  }
}

}  // namespace

// 128 x 256 tiles, persistent (one workgroup per CU). Returns -1 when the launch does not fit this kernel's contract (the caller picks another).
int launch_gemm_v8(GemmParams& p, int nbatch, hipStream_t stream) {
  if (nbatch != 1 || p.out_f32 || !host_staged_ok(p) || p.epi == 2 || (p.N % 8) != 0 || p.K % V8_BK != 0 || p.nsplit != 1) return -1;
  if ((long)p.M * p.ldc * 2 >= (1L << 32) || (p.R && (long)p.M * p.ldr * 2 >= (1L << 32)) || (long)p.M * p.lda * 2 >= (1L << 32) ||
      (long)p.N * p.ldb * 2 >= (1L << 32))
    return -1;                                   // 32-bit byte offsets in the DMA sources and buffer descriptors
  if (p.epi == 1 && ((long)(p.M + p.vit.m_off) * p.vit.NH * 64 * 2 >= (1L << 32) || (p.vit.NH * 64) % V8_BN != 0)) return -1;
  if (p.st_out && (p.N % 128) != 0) return -1;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e0 = hipFuncSetAttribute((const void*)gemm_v8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, V8_SMEM);
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_v8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, V8_SMEM);
    hipError_t e3 = hipFuncSetAttribute((const void*)gemm_v8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, V8_SMEM);
    if (e0 != hipSuccess || e1 != hipSuccess || e3 != hipSuccess) {
      vq3_set_error("gemm v8: hipFuncSetAttribute failed");
      return 2;
    }
    attr_done = true;
  }
  p.mtiles = (p.M + V8_BM - 1) / V8_BM;
  p.ntiles = (p.N + V8_BN - 1) / V8_BN;
  choose_tile_order(p, V8_BM, V8_BN, 1);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    (void)hipGetLastError();
    ncu = n / 8 * 8;
  }
  int nwg = p.mtiles * p.ntiles;
  if (nwg > ncu) nwg = ncu;
  dim3 grid(nwg, 1, 1);
  if (p.epi == 1)
    hipLaunchKernelGGL((gemm_v8_kernel<1>), grid, dim3(512), V8_SMEM, stream, p);
  else if (p.ln_in)
    hipLaunchKernelGGL((gemm_v8_kernel<3>), grid, dim3(512), V8_SMEM, stream, p);
  else
    hipLaunchKernelGGL((gemm_v8_kernel<0>), grid, dim3(512), V8_SMEM, stream, p);
  return 0;
}

}  // namespace vq3gemm
