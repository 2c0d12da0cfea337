// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T)
//
// Both operands are K-contiguous ("NT"): this is torch.nn.functional.linear(x, W) with x = A, W = B, the
// shape every contraction on the VGGT / Perceiver / Qwen3 path takes (QKV/O/FFN projections, lm_head, and
// - by swapping operand roles or feeding transposed copies - every dgrad/wgrad and attention product).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Staging: global_load_dwordx4 -> registers -> ds_write_b128 into a double-buffered, XOR-swizzled LDS image
// (128-byte rows, 16-byte chunk index ^ (row & 7): conflict-free for both the ds_write_b128 and the
// ds_read_b128 fragment reads). One barrier per K tile; next tile's global loads are issued before the MFMAs.
// The MFMA is issued with W as the A operand and x as the B operand so each lane ends up with 4 consecutive
// output columns of one output row -> 8/16-byte epilogue stores and row-wise epilogues without shuffles.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include <cstdio>
#include "gemm_common.h"
#include "vq3_hip.h"

using namespace vq3gemm;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int SMEM_BYTES = 4 * TILE_BYTES;       // 2 stages x (A,B) = 64 KiB

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  int m0, n0;
  tile_coords(p, BM, BN, m0, n0);
  const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z % p.nb2;
  const bf16_t* A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* B = p.B + b1 * p.sB1 + (long)(b2 / p.b2divB) * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;
  const long roff = b1 * p.sR1 + b2 * p.sR2;

  // ---- staging map: chunk c = tid + 256*i -> row = c>>3 (0..127), 16B chunk kc = c&7 ----
  const int srow = tid >> 3, kc = tid & 7;
  const bf16_t* ga[4];
  const bf16_t* gb[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = srow + 32 * i;
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
    int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
    ga[i] = A + (long)ra * p.lda + kc * 8;
    gb[i] = B + (long)rb * p.ldb + kc * 8;
    lds_off[i] = row * 128 + ((kc ^ (row & 7)) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (within one operand tile), k-step 0; k-step 1 flips chunk bit 2
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 64 + i * 16 + fr;
    a_off[i] = row * 128 + ((fq ^ (row & 7)) << 4);
    const int rowb = wn * 64 + i * 16 + fr;
    b_off[i] = rowb * 128 + ((fq ^ (rowb & 7)) << 4);
  }

  const int nt = p.K / BK;
  // Two named register sets: while tile t is multiplied out of LDS, tile t+1 sits in one set (already requested
  // a step ago) and tile t+2 is being requested into the other: global-load latency gets two MFMA phases of cover.
  u32x4 s0a[4], s0b[4], s1a[4], s1b[4];
  auto gload = [&](u32x4 (&ra)[4], u32x4 (&rb)[4], int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(ga[i] + (long)tile * BK);
      rb[i] = *reinterpret_cast<const u32x4*>(gb[i] + (long)tile * BK);
    }
  };
  auto lstore = [&](const u32x4 (&ra)[4], const u32x4 (&rb)[4], int buf) {
    char* Ad = smem + buf * (2 * TILE_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(Ad + lds_off[i]) = ra[i];
      *reinterpret_cast<u32x4*>(Ad + TILE_BYTES + lds_off[i]) = rb[i];
    }
  };
  auto compute = [&](int buf) {
    const char* As = smem + buf * (2 * TILE_BYTES);
    const char* Bs = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // chunk (ks*4 + fq) ^ (row&7): ks*4 only touches bit 2 -> xor 64 bytes
        xa[i] = *reinterpret_cast<const bf16x8*>(As + (a_off[i] ^ (ks << 6)));
        wb[i] = *reinterpret_cast<const bf16x8*>(Bs + (b_off[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
  };

  // Loads and LDS stores are issued UNCONDITIONALLY (tile index clamped to the last tile): with branches around them
  // hipcc's s_waitcnt insertion merges the two paths and drains vmcnt(0) before every ds_write; straight-line code
  // gets the counted vmcnt(8..15) that leaves the newest tile's loads in flight. The redundant tail loads/stores
  // touch only the last tile / the LDS buffer nobody reads any more.
  const int last = nt - 1;
  gload(s0a, s0b, 0);
  lstore(s0a, s0b, 0);
  gload(s0a, s0b, 1 < last ? 1 : last);
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    // even step: LDS[0] = tile t, set0 = tile t+1, request tile t+2 into set1
    gload(s1a, s1b, t + 2 < last ? t + 2 : last);
    compute(0);
    lstore(s0a, s0b, 1);
    __syncthreads();
    if (t + 1 >= nt) break;
    // odd step: LDS[1] = tile t+1, set1 = tile t+2, request tile t+3 into set0
    gload(s0a, s0b, t + 3 < last ? t + 3 : last);
    compute(1);
    lstore(s1a, s1b, 0);
    __syncthreads();
  }

  // ---- epilogue: lane owns C[m][n..n+3], m = m0+wm*64+i*16+fr, n = n0+wn*64+j*16+4*fq ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * fq;
      if (n >= p.N) continue;
      store_quad<OUT_F32>(p, coff, roff, m, n, acc[i][j]);
    }
  }
}

bool g_attr_set = false;

// Kernel choice per shape, from measurements on MI355X (tools/bench_gemm.py, tools/bench_gemm_layouts.py,
// tools/diag/run_stamps.py). Candidates:
//   cfg 7  : 128x128 tile, 8 waves, 2 stages, two workgroups per CU (512 tile slots)  - best for short K (<= 1536)
//   cfg 13 : 128x128 tile, 8 compute + 2 DMA-loader waves, 4-stage ring, one workgroup per CU (256 slots)
//   cfg 11 : 256x128 tile, 8 compute + 2 DMA-loader waves, 3-stage ring, one workgroup per CU (256 slots)
// score = (measured relative speed of the schedule) x (fraction of tile slots busy over all rounds).
// VQ3_GEMM_CFG overrides (benchmarking; -1 = v1 register-staged kernel).
double fill(long tiles, long slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); }

int g_forced_cfg = -2;   // -2 = read VQ3_GEMM_CFG on first use, -3 = automatic, >= -1 = forced

int choose_config(int M, int N, int K, int nbatch) {
  if (g_forced_cfg == -2) {
    const char* e = getenv("VQ3_GEMM_CFG");
    g_forced_cfg = e ? atoi(e) : -3;
  }
  if (g_forced_cfg >= -1) return g_forced_cfg;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nbatch;
  const long t256 = (long)((M + 255) / 256) * ((N + 127) / 128) * nbatch;
  const bool shortk = K <= 1536;
  const double s7 = (shortk ? 1.00 : 0.93) * fill(t128, 512);
  const double s12 = (shortk ? 0.85 : 0.95) * fill(t128, 256);
  const double s11 = (shortk ? 0.90 : 1.00) * fill(t256, 256) * ((double)M / (double)(((M + 255) / 256) * 256)) /
                     ((double)M / (double)(((M + 127) / 128) * 128));
  // (the measured choice usually lands on the 256 x 256 8-phase kernel once its tile grid fills most of a round: this is what a launch
  // gets when nothing may be measured - under graph capture, or in a multi-rank job without a table entry)
  if ((long)((M + 255) / 256) * ((N + 255) / 256) * nbatch >= 192 && K >= 256) return 20;
  if (s11 >= s12 && s11 >= s7) return 11;
  return s12 >= s7 ? 13 : 7;
}

// v3 (k-major operands / K tails) has 128x128 tiles only: 3-stage ring + loader waves vs 2 stages x 2 workgroups.
int g_forced_v3 = 0;     // vq3_gemm_force_config(102 / 103 / 105): pin the k-major kernel's schedule

int choose_v3_stages(int M, int N, int K, int nbatch) {
  if (g_forced_v3) return g_forced_v3;
  static int forced = -1;
  if (forced < 0) {
    const char* e = getenv("VQ3_GEMM_V3_STAGES");   // benchmarking override: 2, 3 or 5 (256x128 tile)
    forced = e ? atoi(e) : 0;
  }
  if (forced == 2 || forced == 3 || forced == 5) return forced;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * nbatch;
  const bool shortk = K <= 1536;
  const double s2 = (shortk ? 1.00 : 0.93) * fill(t128, 512);
  const double s3 = (shortk ? 0.85 : 0.95) * fill(t128, 256);
  return s3 >= s2 ? 3 : 2;
}

// ---- measured kernel choice -------------------------------------------------------------------------------------------
// The heuristics above rank tile configurations by fill; what a shape really gets also depends on K depth, epilogue and
// cache state, and the ranking is wrong often enough to matter (Perceiver ffn1: 130 us chosen, 94 us available). So the
// FIRST call with a new (shape, layout, epilogue kind) times every candidate on the caller's stream - on scratch output, with
// a 320 MB fill between launches so that the weights come from HBM as they do inside a training step - and the winner
// is remembered for the life of the process. Never while the stream is being captured into a graph (the heuristic answers
// then), never when VQ3_GEMM_AUTOTUNE=0 or a configuration is forced. The first call of a shape therefore synchronises
// the stream and may allocate scratch; every later call only enqueues.
typedef std::tuple<int, int, int, int, int> TuneKey;   // M, N, K, nbatch, flags
std::map<TuneKey, int> g_tuned;
std::mutex g_tune_mutex;
void* g_scratch_c = nullptr;
size_t g_scratch_c_bytes = 0;
void* g_flush = nullptr;
constexpr size_t FLUSH_BYTES = 320u << 20;
int g_autotune = -1;
// vq3_gemm_tune_workspace: trial output + cache-flush buffer come from memory the CALLER owns (a torch tensor) - no hipMalloc / hipFree
// from inside a GEMM call. A registered workspace that is too small for a shape means "do not measure this shape" (the table or the
// heuristic answers), never a silent allocation.
void* g_ws = nullptr;
size_t g_ws_bytes = 0;
int g_ws_dev = -1;                    // device the registered workspace lives on (-1: whatever the caller registered by hand)
// vq3_gemm_workspace_provider: the caller's allocator (a torch tensor factory) asked for device memory the moment a measurement - or a
// split-K launch's first use of a stream - actually needs it: an inference-only process or a multi-rank job (hold on) never pays for it
vq3_ws_provider_t g_provider = nullptr;
// vq3_gemm_autotune_hold: nothing is measured while it is on (multi-rank jobs: a measurement synchronises the device under in-flight
// collectives and every rank would rank near-ties on its own; the table and the heuristic are the same function on every rank)
int g_hold = 0;

bool autotune_on(hipStream_t s) {
  if (g_autotune < 0) {
    const char* e = getenv("VQ3_GEMM_AUTOTUNE");
    g_autotune = e ? atoi(e) : 1;
  }
  if (!g_autotune) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  return true;
}

// cfg 30: the 256 x 256 kernel on the row tiles that make WHOLE rounds of the chip, the remaining rows as a second small-tile launch.
// M = 49 392 (8 views x 6 x 1029 tokens), N = 1024 is 193 x 4 = 772 tiles = 3 rounds of 256 CUs + 4 tiles: the 4 cost a fourth round
// (fc2: 500 us where 3 rounds take 375). Rows 0 .. 192 * 256 - 1 are exactly 3 rounds; the last 240 rows are 16 tiles of 128 x 128.
int g_num_cu = 0;
int num_cus() {
  if (!g_num_cu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    (void)hipGetLastError();
    g_num_cu = n;
  }
  return g_num_cu;
}
// rows the 256 x 256 launch takes (0 = the split does not apply to this shape)
int split_rows_main(const GemmParams& p, int nbatch) {
  if (nbatch != 1 || p.epi == 2 || p.M < 512) return 0;
  const long nt = (p.N + 255) / 256, mt = (p.M + 255) / 256, cu = num_cus();
  const long rounds = mt * nt / cu, rem = mt * nt % cu;
  if (rounds < 1 || rem == 0 || rem * 2 > cu) return 0;         // a last round at least half full is left alone
  const long mt_main = rounds * cu / nt;
  const long tail = p.M - mt_main * 256;
  if (mt_main < 1 || tail <= 0 || tail > 2048) return 0;
  return (int)(mt_main * 256);
}
int choose_config(int M, int N, int K, int nbatch);
int launch_split_rows(GemmParams p, int nbatch, hipStream_t s) {
  const int m_main = split_rows_main(p, nbatch);
  if (!m_main) return launch_gemm_v6(p, 0, nbatch, s);
  GemmParams a = p, b = p;
  a.M = m_main;
  b.M = p.M - m_main;
  const size_t esz = p.out_f32 ? 4 : 2;
  b.A = p.A + (long)m_main * p.lda;
  b.C = (char*)p.C + (size_t)m_main * p.ldc * esz;
  if (p.R) b.R = (const char*)p.R + (size_t)m_main * p.ldr * esz;
  if (p.epi == 1) { b.C = p.C; b.vit.m_off = m_main; }      // Q / K / V are addressed by token index, C is a placeholder
  if (p.ln_in) b.ln_in = p.ln_in + (long)m_main * p.ln_parts * 2;
  if (p.st_out) b.st_out = p.st_out + (long)m_main * (p.N >> 7) * 2;
  // (tried: the row tail on a side stream forked from / joined to `s` by events so that it overlaps the main launch instead of following it -
  // no gain, fc1 +3 %: the dispatcher does not interleave the two grids; gpurun_out r3_diag9 epi_30 vs epi_30_serial. Second try: the tail
  // FIRST, on a HIGH-priority side stream, the main launch behind it on `s` - fc1 519 -> 507 us, q|k|v-shaped 338 -> 335, fc2 374 -> 379,
  // proj 156 -> 166: the two event hops cost what the overlap gives)
  int rc = launch_gemm_v6(a, 0, nbatch, s);
  if (rc) return rc;
  const long t128 = (long)((b.M + 127) / 128) * ((b.N + 127) / 128);
  return launch_gemm_v2(b, (p.K <= 1536 && t128 > num_cus()) ? 7 : 13, nbatch, s);
}

// launches one candidate; cand < 100: NT config (v2 cfg id or 20 = v6); cand >= 100: v3 with (cand - 100) stages
int launch_candidate(GemmParams p, int cand, int transA, int transB, int nbatch, hipStream_t s) {
  if (cand == 106 || cand == 107) return (transA && transB) ? launch_gemm_v6_km(p, nbatch, s, cand == 107) : -1;
  if (cand >= 100) return launch_gemm_v3(p, transA, transB, cand - 100, nbatch, s);
  if (cand == 30) return launch_split_rows(p, nbatch, s);
  if (cand >= 20 && cand <= 22) return launch_gemm_v6(p, cand - 20, nbatch, s);
  if (cand == 24) return launch_gemm_v7(p, nbatch, s);          // (-1: outside its contract - the tuner skips it)
  if (cand == 25) return launch_gemm_v6(p, 3, nbatch, s);       // (likewise: 256 x 256 with the last round's tiles split along K)
  return launch_gemm_v2(p, cand, nbatch, s);
}

// returns the fastest candidate, or -1 when tuning could not run (allocation failure, ...)
int tune(const GemmParams& p0, const std::vector<int>& cands, int transA, int transB, int nbatch, hipStream_t s) {
  GemmParams p = p0;
  // tall products are measured on their first 65 536 rows (>= 4 rounds of 256 x 256 tiles for N >= 1024: the ranking no longer depends on
  // M there, and config C4's 395 136-row tower GEMMs would otherwise need 3.2 GB of trial output and 30 ms per trial launch)
  if (!transA && p.M > 65536 && p.epi != 2 && p.epi != 3) p.M = 65536;
  const size_t esz = p.out_f32 ? 4 : 2;
  const size_t need = ((size_t)p.M * p.N * esz * (size_t)nbatch + 256 + 255) & ~(size_t)255;
  void* scratch_c = nullptr;
  void* flush = nullptr;
  if (g_provider) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_ws || g_ws_dev != dev || need + FLUSH_BYTES > g_ws_bytes) {
      // lazily, for THIS device, at least what this shape needs (the provider rounds up to its default size and keeps the tensor alive)
      void* w = g_provider((int64_t)(need + FLUSH_BYTES), dev, 0);
      if (!w) return -1;                                       // no memory to measure in: the table / heuristic answers
      g_ws = w; g_ws_bytes = need + FLUSH_BYTES; g_ws_dev = dev;
    }
  }
  if (g_ws) {
    if (need + FLUSH_BYTES > g_ws_bytes) return -1;           // the caller's workspace decides what may be measured
    scratch_c = g_ws;
    flush = (char*)g_ws + need;
  } else {
    // no workspace registered (a plain C caller): the library's own buffers, allocated on first use
    if (need > g_scratch_c_bytes) {
      if (g_scratch_c) (void)hipFree(g_scratch_c);
      g_scratch_c = nullptr; g_scratch_c_bytes = 0;
      if (hipMalloc(&g_scratch_c, need) != hipSuccess) { (void)hipGetLastError(); return -1; }
      g_scratch_c_bytes = need;
    }
    if (!g_flush && hipMalloc(&g_flush, FLUSH_BYTES) != hipSuccess) { (void)hipGetLastError(); g_flush = nullptr; return -1; }
    scratch_c = g_scratch_c;
    flush = g_flush;
  }
  // trial output: dense scratch, no read-modify-write operands (an in-place residual or accumulate target must not be touched)
  if (p.epi == 0) p.C = scratch_c;      // (epi == 1 writes Q / K / V, epi == 3 gate|up and act: idempotent, no read-modify-write)
  if (p.epi != 3) {
    p.ldc = p.N;
    p.sC1 = (long)p.M * p.N * p.nb2; p.sC2 = (long)p.M * p.N;
    p.vec_ok = (p.N % 4 == 0) ? 1 : 0;
  }
  p.R = nullptr; p.ldr = 0; p.accumulate = 0;
  p.sR1 = p.sR2 = 0;
  (void)hipDeviceSynchronize();   // measure alone: work queued on other streams (the wgrad stream) would skew the ranking
  (void)hipGetLastError();
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return -1; }
  int best = -1;
  float best_ms = 1e30f;
  const int reps = 5;       // (the minimum of 5 cold launches: with 3 the ranking of near-ties flipped from box to box)
  for (int cand : cands) {
    if (launch_candidate(p, cand, transA, transB, nbatch, s) != 0) continue;      // warm-up (attribute set-up, code load)
    float tmin = 1e30f;
    for (int r = 0; r < reps; ++r) {
      (void)hipMemsetAsync(flush, r, FLUSH_BYTES, s);
      (void)hipEventRecord(e0, s);
      if (launch_candidate(p, cand, transA, transB, nbatch, s) != 0) { tmin = 1e30f; break; }
      (void)hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) { tmin = 1e30f; break; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      tmin = std::min(tmin, ms);
    }
    if (tmin < best_ms) { best_ms = tmin; best = cand; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (hipGetLastError() != hipSuccess) return -1;
  if (getenv("VQ3_GEMM_AUTOTUNE_LOG"))
    fprintf(stderr, "[vq3 gemm autotune] M=%d N=%d K=%d batch=%d tA=%d tB=%d -> %d (%.1f us)\n", p.M, p.N, p.K, nbatch, transA, transB,
            best, best_ms * 1e3f);
  return best;
}

// VQ3_GEMM_TUNE_FILE=<path>: the table is read from that file before the first choice and every newly measured entry is appended to it
// ("M N K batch flags cfg" per line) - a second run, or the other ranks of a job pointed at the same file, then make the SAME choices
// (same kernels, same summation order) without measuring. Without it the choice is timing-based and may differ from run to run and
// from rank to rank (replicas still agree bit for bit: they apply the same all-reduced gradient); VQ3_GEMM_AUTOTUNE=0 turns measuring off.
bool g_tune_file_read = false;
int tune_table_read(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int m, n, k, b, fl, cfg, cnt = 0;
  char line[256];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    if (sscanf(line, "%d %d %d %d %d %d", &m, &n, &k, &b, &fl, &cfg) == 6) { g_tuned[TuneKey(m, n, k, b, fl)] = cfg; ++cnt; }
  }
  fclose(f);
  return cnt;
}
void tune_file_read() {
  g_tune_file_read = true;
  const char* path = getenv("VQ3_GEMM_TUNE_FILE");
  if (!path || !*path) return;
  (void)tune_table_read(path);
}
void tune_file_append(const TuneKey& key, int cfg) {
  const char* path = getenv("VQ3_GEMM_TUNE_FILE");
  if (!path || !*path) return;
  FILE* f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%d %d %d %d %d %d\n", std::get<0>(key), std::get<1>(key), std::get<2>(key), std::get<3>(key), std::get<4>(key), cfg);
  fclose(f);
}

int tuned_choice(const GemmParams& p, int transA, int transB, int nbatch, hipStream_t s, const std::vector<int>& cands, int fallback,
                 bool v3_path = false) {
  // (the key carries what selects a code path inside the candidates: layout, epilogue kind, and whether the 4-wide / whole-row stores apply)
  const bool rows16 = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                      (!p.R || ((p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.R) & 15) == 0)));
  const int flags = (transA ? 1 : 0) | (transB ? 2 : 0) | (p.out_f32 ? 4 : 0) | (p.accumulate ? 8 : 0) | (p.R ? 16 : 0) |
                    (p.bias ? 32 : 0) | (p.act << 6) | (p.colscale ? 256 : 0) | (p.epi << 9) | (p.ln_in ? 4096 : 0) | (p.st_out ? 8192 : 0) |
                    (p.vec_ok ? 0 : 16384) | (rows16 ? 0 : 32768) | (v3_path ? 65536 : 0);
  // small extents are remembered per bucket (M < 256 in steps of 32, K < 512 in steps of 64): the labelled rows of a pass - the lm_head's M,
  // its weight gradient's K - change from pass to pass, and a measurement (a device synchronisation, ~10 ms of trial launches) for every new
  // count landed inside training steps: bench.py's instrumented window showed a 15.6 ms "launch" of the [149, 151 937, 2560] product
  const int mkey = p.M < 256 ? (p.M + 31) / 32 * 32 : p.M;
  const int kkey = p.K < 512 ? (p.K + 63) / 64 * 64 : p.K;
  const TuneKey key(mkey, p.N, kkey, nbatch, flags);
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  if (!g_tune_file_read) tune_file_read();
  static int log_on = -1;
  if (log_on < 0) log_on = getenv("VQ3_GEMM_AUTOTUNE_LOG") ? 1 : 0;
  auto it = g_tuned.find(key);
  // (an entry is only as good as the candidate list it was measured against: a table written by another build, or by the other
  // dispatch path under the same key, may name a configuration this call cannot run - such an entry is measured again / falls back)
  static std::map<TuneKey, int> said;      // (VQ3_GEMM_AUTOTUNE_LOG: every key's choice once, so that two ranks' logs can be compared)
  if (it != g_tuned.end() && std::find(cands.begin(), cands.end(), it->second) != cands.end()) {
    if (log_on && said.emplace(key, it->second).second)
      fprintf(stderr, "[vq3 gemm choice] M=%d N=%d K=%d batch=%d flags=%d -> %d (table)\n", mkey, p.N, kkey, nbatch, flags, it->second);
    return it->second;
  }
  if (!autotune_on(s) || g_hold) {
    // not remembered: a later, tunable call may still measure
    if (log_on && said.emplace(key, fallback).second)
      fprintf(stderr, "[vq3 gemm choice] M=%d N=%d K=%d batch=%d flags=%d -> %d (heuristic)\n", mkey, p.N, kkey, nbatch, flags, fallback);
    return fallback;
  }
  int best = tune(p, cands, transA, transB, nbatch, s);
  if (best < 0) best = fallback;
  g_tuned[key] = best;
  tune_file_append(key, best);
  return best;
}

}  // namespace

static char kSwigluNoGu;      // vq3_gemm_swiglu_fwd without a gate|up output (a forward that no backward follows): "mode on, pointer null"
static int gemm_dispatch(const vq3_gemm_desc* d, const vq3_vit_qkv_epilogue* ve, void* stream, const void* sw_gu = nullptr,
                         void* sw_dgu = nullptr, const vq3_gemm_ln_fold* ln = nullptr, void* sw_fwd_gu = nullptr) {
  VQ3_CHECK_ARG(d != nullptr, "gemm: null descriptor");
  VQ3_CHECK_ARG(d->A && d->B && (d->C || ve || sw_dgu), "gemm: null operand pointer");
  VQ3_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  VQ3_CHECK_ARG(d->K % 8 == 0 && d->K >= 8, "gemm: K=%d must be a positive multiple of 8", d->K);
  VQ3_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm: lda=%d / ldb=%d must be multiples of 8", d->lda, d->ldb);
  VQ3_CHECK_ARG(d->lda >= (d->transA ? d->M : d->K) && d->ldb >= (d->transB ? d->N : d->K),
                "gemm: leading dims smaller than the operand's inner extent");
  VQ3_CHECK_ARG(!d->transA || (d->M % 8 == 0), "gemm: a k-major A needs M %% 8 == 0 (M=%d)", d->M);
  VQ3_CHECK_ARG(!d->transB || (d->N % 8 == 0), "gemm: a k-major B needs N %% 8 == 0 (N=%d)", d->N);
  VQ3_CHECK_ARG(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->B % 16 == 0), "gemm: A/B must be 16-byte aligned");
  VQ3_CHECK_ARG(d->sA1 % 8 == 0 && d->sA2 % 8 == 0 && d->sB1 % 8 == 0 && d->sB2 % 8 == 0,
                "gemm: batch strides of A/B must be multiples of 8 elements");
  VQ3_CHECK_ARG(ve || sw_dgu || sw_fwd_gu || d->ldc >= d->N, "gemm: ldc=%d < N=%d", d->ldc, d->N);
  VQ3_CHECK_ARG(d->nb1 >= 1 && d->nb2 >= 1 && d->b2divB >= 1, "gemm: bad batch dims");
  VQ3_CHECK_ARG((long)d->nb1 * d->nb2 <= 65535, "gemm: too many batches");
  VQ3_CHECK_ARG(d->act >= 0 && d->act <= 2, "gemm: bad activation %d", d->act);
  VQ3_CHECK_ARG(!(d->R) || d->ldr >= d->N, "gemm: ldr < N");
  VQ3_CHECK_ARG((!d->bias || (uintptr_t)d->bias % 16 == 0) && (!d->colscale || (uintptr_t)d->colscale % 16 == 0),
                "gemm: bias / colscale must be 16-byte aligned");

  GemmParams p;
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C;
  p.bias = (const float*)d->bias; p.colscale = (const float*)d->colscale; p.R = d->R;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.sA1 = d->sA1; p.sA2 = d->sA2; p.sB1 = d->sB1; p.sB2 = d->sB2;
  p.sC1 = d->sC1; p.sC2 = d->sC2; p.sR1 = d->sR1; p.sR2 = d->sR2;
  p.nb2 = d->nb2; p.b2divB = d->b2divB;
  p.act = d->act; p.out_f32 = d->out_f32; p.accumulate = d->accumulate; p.alpha = d->alpha;
  p.kper = d->K; p.nsplit = 1;
  p.epi = 0;
  p.stamps = nullptr;
  p.nbw = 1;
  p.stagger = 0;
  p.sk_full = p.sk_rem = p.sk_s = 0; p.sk_ws = nullptr; p.sk_cnt = nullptr; p.sk_err = nullptr; p.sk_spin = 1u << 23; p.f8_rs = p.f8_cs = nullptr;
  if (ve) {
    VQ3_CHECK_ARG(ve->Q && ve->K && ve->V, "gemm_vit_qkv: null output pointer");
    VQ3_CHECK_ARG(!d->transA && !d->transB && d->K % BK == 0 && !d->out_f32 && !d->accumulate && !d->R && !d->colscale && d->act == 0 &&
                      d->ksplit <= 1 && d->nb1 == 1 && d->nb2 == 1,
                  "gemm_vit_qkv: plain NT bf16 GEMM with K %% 64 == 0 and at most a bias");
    VQ3_CHECK_ARG(ve->NH > 0 && d->N == 3 * ve->NH * 64 && ve->N > 0 && d->M % ve->N == 0, "gemm_vit_qkv: N must be 3*NH*64 and M a multiple of the group length");
    VQ3_CHECK_ARG(!ve->use_norm || (ve->qn_w && ve->qn_b && ve->kn_w && ve->kn_b), "gemm_vit_qkv: norm weights missing");
    VQ3_CHECK_ARG(!ve->use_rope || (ve->cos && ve->sin && ve->tokens_per_frame > 0 && ve->Wp > 0), "gemm_vit_qkv: rope tables missing");
    VQ3_CHECK_ARG(((uintptr_t)ve->Q | (uintptr_t)ve->K | (uintptr_t)ve->V | (uintptr_t)ve->cos | (uintptr_t)ve->sin) % 16 == 0,
                  "gemm_vit_qkv: Q/K/V and the rope tables must be 16-byte aligned");
    p.epi = 1;
    p.vit.Q = (bf16_t*)ve->Q; p.vit.K = (bf16_t*)ve->K; p.vit.V = (bf16_t*)ve->V;
    p.vit.qn_w = ve->qn_w; p.vit.qn_b = ve->qn_b; p.vit.kn_w = ve->kn_w; p.vit.kn_b = ve->kn_b;
    p.vit.cos = (const bf16_t*)ve->cos; p.vit.sin = (const bf16_t*)ve->sin;
    p.vit.N = ve->N; p.vit.NH = ve->NH; p.vit.P = ve->tokens_per_frame; p.vit.patch_start = ve->patch_start; p.vit.Wp = ve->Wp;
    {   // the largest table row the epilogue asks for is max(patch rows, patch columns) (positions are 1-based; 0 = special tokens)
      const int P_ = ve->tokens_per_frame > 0 ? ve->tokens_per_frame : ve->N, np = P_ - ve->patch_start;
      const int hp = ve->Wp > 0 && np > 0 ? (np + ve->Wp - 1) / ve->Wp : 0;
      p.vit.rope_rows = (hp > ve->Wp ? hp : ve->Wp) + 1;
    }
    p.vit.use_norm = ve->use_norm; p.vit.use_rope = ve->use_rope; p.vit.eps = ve->eps;
    p.vit.m_off = 0;
    p.C = ve->Q; p.ldc = d->N;          // placeholders: the staged epilogue checks their alignment, nothing is stored through them
    p.sC1 = p.sC2 = 0;
  }
  p.sw_gu = nullptr; p.sw_dgu = nullptr;
  if (sw_dgu) {
    VQ3_CHECK_ARG(sw_gu != nullptr, "gemm_swiglu_bwd: null gate|up pointer");
    VQ3_CHECK_ARG(!d->out_f32 && !d->accumulate && !d->R && !d->colscale && !d->bias && d->act == 0 && d->ksplit <= 1 && d->nb1 == 1 &&
                      d->nb2 == 1 && d->N % 8 == 0 && d->alpha == 1.f,
                  "gemm_swiglu_bwd: plain bf16 GEMM (no epilogue, no batch), N %% 8 == 0");
    VQ3_CHECK_ARG(((uintptr_t)sw_gu | (uintptr_t)sw_dgu) % 16 == 0, "gemm_swiglu_bwd: gu / dgu must be 16-byte aligned");
    p.epi = 2;
    p.sw_gu = (const bf16_t*)sw_gu; p.sw_dgu = (bf16_t*)sw_dgu;
    p.C = sw_dgu; p.ldc = d->N;         // placeholders (alignment checks only)
    p.sC1 = p.sC2 = 0;
  }
  if (sw_fwd_gu) {
    VQ3_CHECK_ARG(!ve && !sw_dgu && !ln, "gemm_swiglu_fwd: no other fused epilogue");
    VQ3_CHECK_ARG(!d->transA && !d->transB && d->K % BK == 0 && !d->out_f32 && !d->accumulate && !d->R && !d->colscale && !d->bias &&
                      d->act == 0 && d->ksplit <= 1 && d->nb1 == 1 && d->nb2 == 1 && d->alpha == 1.f,
                  "gemm_swiglu_fwd: plain NT bf16 GEMM (no epilogue, no batch), K %% 64 == 0");
    VQ3_CHECK_ARG(d->N % 256 == 0 && d->ldc >= d->N / 2 && d->ldc % 8 == 0, "gemm_swiglu_fwd: N = 2 I with I %% 128 == 0, act rows of >= I elements (ld %% 8)");
    if (sw_fwd_gu == (void*)&kSwigluNoGu) sw_fwd_gu = nullptr;
    VQ3_CHECK_ARG(((uintptr_t)sw_fwd_gu | (uintptr_t)d->C) % 16 == 0, "gemm_swiglu_fwd: gate|up / act must be 16-byte aligned");
    p.epi = 3;
    p.sw_dgu = (bf16_t*)sw_fwd_gu;                 // null: the epilogue stores act only
    p.sC1 = p.sC2 = 0;
  }
  const int esz = d->out_f32 ? 4 : 2;
  bool vec = (p.ldc % 4 == 0) && (p.sC1 % 4 == 0) && (p.sC2 % 4 == 0) && ((uintptr_t)p.C % (4 * esz) == 0);
  if (d->R) vec = vec && (d->ldr % 4 == 0) && (d->sR1 % 4 == 0) && (d->sR2 % 4 == 0) && ((uintptr_t)d->R % (4 * esz) == 0);
  p.vec_ok = vec ? 1 : 0;

  hipStream_t s = (hipStream_t)stream;
  const int nbatch = d->nb1 * d->nb2;
  p.ln_in = nullptr; p.ln_c = nullptr; p.ln_parts = 0; p.ln_eps = 0.f; p.st_out = nullptr;
  if (ln && (ln->stats_in || ln->stats_out)) {
    // both halves live in the LDS-staged epilogue: the conditions of gemm_common.h: staged_ok must hold, or the kernels would
    // silently take the register epilogue
    VQ3_CHECK_ARG(!sw_dgu && !d->out_f32 && d->ksplit <= 1 && nbatch == 1 && p.vec_ok && p.ldc % 8 == 0 && ((uintptr_t)p.C % 16 == 0) &&
                      (!d->R || (d->ldr % 8 == 0 && (uintptr_t)d->R % 16 == 0)) && g_forced_cfg != -1 && g_forced_cfg != 4 && g_forced_cfg != 5,
                  "gemm ln fold: needs the staged bf16 epilogue (bf16 C, one batch, 16-byte aligned rows, tile width >= 128)");
    if (ln->stats_in) {
      VQ3_CHECK_ARG(ln->colsum && ln->parts_in > 0 && ln->parts_in <= 64 && ((uintptr_t)ln->stats_in % 8 == 0) && ((uintptr_t)ln->colsum % 16 == 0),
                    "gemm ln fold: stats_in needs colsum [N] (16-byte aligned) and 1..64 (sum, sumsq) pairs per row");
      p.ln_in = ln->stats_in; p.ln_c = ln->colsum; p.ln_parts = ln->parts_in; p.ln_eps = ln->eps;
    }
    if (ln->stats_out) {
      VQ3_CHECK_ARG(!ve && d->N % 128 == 0 && ((uintptr_t)ln->stats_out % 8 == 0), "gemm ln fold: stats_out needs N %% 128 == 0 and a plain C epilogue");
      p.st_out = ln->stats_out;
    }
  }
  if (d->ksplit > 1) {
    VQ3_CHECK_ARG(d->out_f32 && !d->bias && !d->colscale && !d->R && d->act == 0 && !d->accumulate,
                  "gemm: split-K needs a zero-initialised f32 C and no epilogue");
    int kper = ((d->K + d->ksplit - 1) / d->ksplit + 63) / 64 * 64;
    p.kper = kper;
    p.nsplit = (d->K + kper - 1) / kper;
    VQ3_CHECK_ARG(p.nsplit <= 65535, "gemm: too many K slices");
    const int rc = launch_gemm_v3(p, d->transA, d->transB, 2, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_bf16_nt(v3 split-K)");
    return 0;
  }
  if (d->transA || d->transB || d->K % BK != 0) {
    int nstage = choose_v3_stages(d->M, d->N, d->K, nbatch);
    if (!g_forced_v3 && getenv("VQ3_GEMM_V3_STAGES") == nullptr && (long)d->M * d->N * d->K >= (1L << 24)) {
      std::vector<int> cands = {102, 103, 105};
      if (d->transA && d->transB && (long)((d->M + 255) / 256) * ((d->N + 255) / 256) >= 64) cands.push_back(106);
      nstage = tuned_choice(p, d->transA, d->transB, nbatch, s, cands, 100 + nstage, true) - 100;
    }
    if ((nstage == 6 || nstage == 7) && !(d->transA && d->transB)) nstage = 5;
    if (nstage == 6 || nstage == 7) {          // both operands k-major on the 256 x 256 8-phase kernel (gemm6.hip)
      const int rc6 = launch_gemm_v6_km(p, nbatch, s, nstage == 7);
      if (rc6 > 0) return rc6;
      if (rc6 == 0) {
        VQ3_CHECK_LAUNCH("gemm_bf16_nt(v6 k-major)");
        return 0;
      }
      nstage = 5;
    }
    const int rc = launch_gemm_v3(p, d->transA, d->transB, nstage, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_bf16_nt(v3)");
    return 0;
  }
  if (p.epi == 3) {
    // the 8-phase kernels only (the epilogue needs gate and up of a feature in ONE tile: gemm6.hip stages the two weight row ranges)
    int cfg = ((g_forced_cfg >= 20 && g_forced_cfg <= 22) || g_forced_cfg == 24) ? g_forced_cfg : 20;
    if (g_forced_cfg == -3 && (long)d->M * d->N * d->K >= (1L << 24)) cfg = tuned_choice(p, 0, 0, nbatch, s, {20, 21, 22}, 20);
    if (cfg == 24) {
      const int rc7 = launch_gemm_v7(p, nbatch, s);
      if (rc7 > 0) return rc7;
      if (rc7 == 0) {
        VQ3_CHECK_LAUNCH("gemm_swiglu_fwd(v7)");
        return 0;
      }
      cfg = 20;
    }
    const int rc = launch_gemm_v6(p, cfg - 20, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_swiglu_fwd(v6)");
    return 0;
  }
  int cfg = choose_config(d->M, d->N, d->K, nbatch);
  if (g_forced_cfg == -3 && (long)d->M * d->N * d->K >= (1L << 24)) {
    // candidates: 2-stage x 2 workgroups (7), 256x128 / 128x128 loader rings (11, 13), 8-wave 128x128 ring (9), and the
    // 256x256 8-phase kernel (20) once the tile grid can feed it
    std::vector<int> cands = {7, 11, 13, 9};
    if ((long)((d->M + 255) / 256) * ((d->N + 255) / 256) * nbatch >= 64) cands.push_back(20);
    if ((long)((d->M + 255) / 256) * ((d->N + 127) / 128) * nbatch >= 64) cands.push_back(21);
    if ((long)((d->M + 127) / 128) * ((d->N + 255) / 256) * nbatch >= 64) cands.push_back(22);
    if (split_rows_main(p, nbatch)) cands.push_back(30);
    cands.push_back(25);          // (skipped by the tuner where the last-round split does not apply: the launcher returns -1)
    if ((long)((d->M + 255) / 256) * ((d->N + 127) / 128) * nbatch >= 256) cands.push_back(24);     // two workgroups per CU (gemm7.hip)
    cfg = tuned_choice(p, 0, 0, nbatch, s, cands, cfg);
  }
  if (cfg == 24) {
    const int rc = launch_gemm_v7(p, nbatch, s);
    if (rc > 0) return rc;
    if (rc == 0) {
      VQ3_CHECK_LAUNCH("gemm_bf16_nt(v7)");
      return 0;
    }
    cfg = 13;      // outside v7's contract: a kernel that takes everything
  }
  if (cfg == 25) {
    const int rc = launch_gemm_v6(p, 3, nbatch, s);
    if (rc > 0) return rc;
    if (rc == 0) {
      VQ3_CHECK_LAUNCH("gemm_bf16_nt(v6, last round split)");
      return 0;
    }
    cfg = 20;      // the split does not apply to this shape / stream state: the plain 256 x 256 launch
  }
  if (cfg == 30) {
    const int rc = launch_split_rows(p, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_bf16_nt(v6 + row tail)");
    return 0;
  }
  if (cfg >= 20 && cfg <= 22) {
    const int rc = launch_gemm_v6(p, cfg - 20, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_bf16_nt(v6)");
    return 0;
  }
  if (cfg >= 0) {
    const int rc = launch_gemm_v2(p, cfg, nbatch, s);
    if (rc) return rc;
    VQ3_CHECK_LAUNCH("gemm_bf16_nt(v2)");
    return 0;
  }
  VQ3_CHECK_ARG(p.epi == 0, "gemm_vit_qkv: not available on the register-staged reference kernel (cfg -1)");
  if (!g_attr_set) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_nt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_nt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      vq3_set_error("gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
      return 2;
    }
    g_attr_set = true;
  }
  p.mtiles = (d->M + BM - 1) / BM;
  p.ntiles = (d->N + BN - 1) / BN;
  choose_tile_order(p, BM, BN, 2);
  VQ3_CHECK_ARG((long)p.mtiles * p.ntiles < (1L << 31), "gemm: too many tiles");
  dim3 grid(p.mtiles * p.ntiles, 1, nbatch);
  if (d->out_f32)
    hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), SMEM_BYTES, s, p);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), SMEM_BYTES, s, p);
  VQ3_CHECK_LAUNCH("gemm_bf16_nt");
  return 0;
}

extern "C" int vq3_gemm_bf16_nt(const vq3_gemm_desc* d, void* stream) { return gemm_dispatch(d, nullptr, stream); }

extern "C" int vq3_gemm_vit_qkv(const vq3_gemm_desc* d, const vq3_vit_qkv_epilogue* epi, void* stream) {
  VQ3_CHECK_ARG(epi != nullptr, "gemm_vit_qkv: null epilogue descriptor");
  return gemm_dispatch(d, epi, stream);
}

extern "C" int vq3_gemm_bf16_nt_ln(const vq3_gemm_desc* d, const vq3_gemm_ln_fold* ln, void* stream) {
  return gemm_dispatch(d, nullptr, stream, nullptr, nullptr, ln);
}

extern "C" int vq3_gemm_vit_qkv_ln(const vq3_gemm_desc* d, const vq3_vit_qkv_epilogue* epi, const vq3_gemm_ln_fold* ln, void* stream) {
  VQ3_CHECK_ARG(epi != nullptr, "gemm_vit_qkv_ln: null epilogue descriptor");
  return gemm_dispatch(d, epi, stream, nullptr, nullptr, ln);
}

extern "C" int vq3_gemm_swiglu_bwd(const vq3_gemm_desc* d, const void* gu, void* dgu, void* stream) {
  VQ3_CHECK_ARG(dgu != nullptr, "gemm_swiglu_bwd: null output pointer");
  return gemm_dispatch(d, nullptr, stream, gu, dgu);
}

extern "C" int vq3_gemm_swiglu_fwd(const vq3_gemm_desc* d, void* gu, void* stream) {
  VQ3_CHECK_ARG(d != nullptr && d->C != nullptr, "gemm_swiglu_fwd: null output pointer");
  return gemm_dispatch(d, nullptr, stream, nullptr, nullptr, nullptr, gu ? gu : (void*)&kSwigluNoGu);     // gu == NULL: act only
}

extern "C" int vq3_gemm_tile_order(int32_t M, int32_t N, int32_t bm, int32_t bn, int32_t wg_per_cu, int32_t* xm_out, int32_t* band_out,
                                   int32_t* order) {
  VQ3_CHECK_ARG(M > 0 && N > 0 && bm > 0 && bn > 0 && wg_per_cu >= 1, "gemm_tile_order: bad arguments");
  GemmParams p{};
  p.M = M; p.N = N;
  p.mtiles = (M + bm - 1) / bm;
  p.ntiles = (N + bn - 1) / bn;
  VQ3_CHECK_ARG((long)p.mtiles * p.ntiles < (1L << 28), "gemm_tile_order: too many tiles");
  choose_tile_order(p, bm, bn, wg_per_cu);
  if (xm_out) *xm_out = p.xm;
  if (band_out) *band_out = p.nbw;
  if (order) {
    const int n = p.mtiles * p.ntiles;
    for (int i = 0; i < n; ++i) {
      int m0 = 0, n0 = 0;
      tile_coords_id(p, i, bm, bn, m0, n0);
      order[2 * i] = m0 / bm;
      order[2 * i + 1] = n0 / bn;
    }
  }
  return 0;
}

extern "C" int vq3_gemm_split_plan(int32_t M, int32_t N, int32_t K, int32_t ncu, int32_t* full_out, int32_t* rem_out, int32_t* slices_out) {
  VQ3_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0 && ncu >= 8 && slices_out, "gemm_split_plan: bad arguments");
  GemmParams p{};
  p.M = M; p.N = N; p.K = K;
  int full = 0, rem = 0;
  const int sl = gemm_split_plan(p, 1, ncu, &full, &rem);
  *slices_out = sl >= 2 ? sl : 0;
  if (full_out) *full_out = sl >= 2 ? full : 0;
  if (rem_out) *rem_out = sl >= 2 ? rem : 0;
  return 0;
}

extern "C" int vq3_gemm_split_status(void* stream, int32_t* gave_up) {
  VQ3_CHECK_ARG(gave_up != nullptr, "gemm_split_status: null output");
  const int r = gemm_split_gave_up((hipStream_t)stream);
  VQ3_CHECK_ARG(r != -2, "gemm_split_status: could not synchronise the stream");
  *gave_up = r > 0 ? 1 : 0;
  return 0;
}

extern "C" int vq3_gemm_split_debug_spin_bound(int64_t polls) {
  VQ3_CHECK_ARG(polls >= 0 && polls <= (1ll << 31), "gemm_split_debug_spin_bound: polls out of range");
  gemm_split_set_spin_bound(polls ? (unsigned)polls : (1u << 23));
  return 0;
}

extern "C" int vq3_gemm_split_poll(int32_t* gave_up, int32_t clear) {
  VQ3_CHECK_ARG(gave_up != nullptr, "gemm_split_poll: null output");
  *gave_up = gemm_split_poll(clear != 0) > 0 ? 1 : 0;
  return 0;
}

extern "C" int vq3_gemm_tune_table_load(const char* path, int32_t* entries_out) {
  VQ3_CHECK_ARG(path && *path, "gemm_tune_table_load: null path");
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  const int n = tune_table_read(path);
  VQ3_CHECK_ARG(n >= 0, "gemm_tune_table_load: cannot read %s", path);
  if (entries_out) *entries_out = n;
  return 0;
}

extern "C" int vq3_gemm_tune_workspace(void* ptr, int64_t bytes) {
  VQ3_CHECK_ARG((ptr == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)ptr % 256 == 0), "gemm_tune_workspace: bad workspace");
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_ws = ptr; g_ws_bytes = (size_t)bytes; g_ws_dev = -1;
  return 0;
}

extern "C" int vq3_gemm_autotune_hold(int32_t on) {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_hold = on ? 1 : 0;
  return 0;
}

extern "C" int vq3_gemm_workspace_provider(vq3_ws_provider_t fn) {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_provider = fn;
  if (g_ws_dev >= 0) { g_ws = nullptr; g_ws_bytes = 0; g_ws_dev = -1; }      // a workspace the old provider gave is no longer ours
  gemm_split_set_provider(fn);
  return 0;
}

extern "C" int vq3_gemm_force_config(int32_t cfg) {
  if (cfg == 102 || cfg == 103 || cfg == 105 || cfg == 106 || cfg == 107) {       // schedule of the any-layout kernels (gemm3.hip; 106 / 107: gemm6.hip k-major)
    g_forced_v3 = cfg - 100;
    return 0;
  }
  VQ3_CHECK_ARG(cfg >= -3 && cfg <= 64 && cfg != -2, "gemm_force_config: cfg %d out of range", cfg);
  g_forced_cfg = cfg;
  if (cfg == -3) g_forced_v3 = 0;
  return 0;
}
