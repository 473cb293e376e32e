// frame.hip — host-side orchestration of ONE frame's depth-sliced bin / sort / composite pipeline and of its backward,
// behind two C-ABI entry points (gs_frame_forward / gs_frame_backward).
//
// Stands where the fork's Python layer stands between `project_gaussians` and the returned image ("the
// caller owns all tensors; the library allocates nothing; workspace passed in"): the ~20 launches of the depth
// pre-sort and the ~22 launches of every depth slice used to be issued one by one from Python (ops.sliced_forward: one
// ctypes call and several torch allocations each), and the GPU idled behind the host for 0.15 ms of the 2.4 ms headline
// step and 0.9 ms of the 11 ms fitted-model-like step.  Here the whole sequence is issued from C++ out of ONE
// caller-owned arena (bump allocation, nothing is ever freed inside a frame), the slice plan and the one "is any tile
// still open?" word per slice come back through caller-owned pinned memory, and the backward walks the slice table the
// forward left in a plain host struct.  Every launch goes through the same exported entry points the Python
// orchestration uses (binning.hip / raster.hip / raster_bwd.hip / project.hip), so the two orchestrations produce the
// same bytes (tests compare them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include "../../include/gsdeblur.h"

#define GS_OK 0
#define GS_ERR_INVALID 1
#define GS_ERR_WORKSPACE 3
#define GS_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kTile = 16;
constexpr int kKMax = GS_FRAME_MAX_SLICES;     // planned slices per frame (budget doubles per slice)
constexpr int kIdsPad = 8;                     // the scalar-cache compositors read their lists in aligned groups of four

struct Arena {
  char* base;
  long long cap, off;
  bool ok;
  Arena(void* b, long long c, long long start = 0) : base((char*)b), cap(c), off(start), ok(true) {}
  static long long up(long long x) { return (x + 255) & ~255ll; }
  // bytes the allocation WOULD end at (used to price a plan before committing to it)
  template <typename T> T* take(long long count) {
    const long long a = up(off), bytes = count * (long long)sizeof(T);
    if (!ok || a + bytes > cap) { ok = false; off = a + bytes; return nullptr; }
    off = a + bytes;
    return reinterpret_cast<T*>(base + a);
  }
  long long offset_of(const void* p) const { return p ? (long long)((const char*)p - base) : -1; }
};

inline int bits_for(long long n) {
  int b = 1;
  while ((1ll << b) < std::max(2ll, n)) ++b;
  return b;
}

inline int hip_status(hipError_t e) { return e == hipSuccess ? GS_OK : 1000 + (int)e; }

// ---- optional stage timing (debug / bench): HIP events on the launch stream -------------------------------------
// Process-wide, mutex-guarded (concurrent callers raced on these; they cannot be thread-local — the
// backward of a frame runs on torch's autograd thread while the reader sits on the thread that ran the forward).
struct StageEvents { int stage; hipEvent_t a, b; };
std::atomic<unsigned> g_profile_mask{0};
std::mutex g_profile_mu;
std::vector<StageEvents> g_events;            // recorded since the last read   (guarded by g_profile_mu)
std::vector<StageEvents> g_pool;              // reusable event pairs           (guarded by g_profile_mu)

struct StageScope {
  hipStream_t st;
  bool on;
  StageEvents ev;
  StageScope(int stage, hipStream_t s) : st(s), on((g_profile_mask.load(std::memory_order_relaxed) >> stage) & 1u) {
    if (!on) return;
    bool have = false;
    {
      std::lock_guard<std::mutex> lk(g_profile_mu);
      if (!g_pool.empty()) { ev = g_pool.back(); g_pool.pop_back(); have = true; }
    }
    if (!have) { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
    ev.stage = stage;
    (void)hipEventRecord(ev.a, st);
  }
  ~StageScope() {
    if (!on) return;
    (void)hipEventRecord(ev.b, st);
    std::lock_guard<std::mutex> lk(g_profile_mu);
    g_events.push_back(ev);
  }
};

// ---- read-backs without a stream synchronisation -------------------------------------------------------------------
// hipMemcpyAsync(device -> pinned) + hipStreamSynchronize costs a blit kernel, a completion signal and the runtime's
// wake-up before the host may issue the next launch (~40-50 us of GPU idle per read-back in the step timelines).
// Instead a one-block kernel copies the words straight into the caller's pinned buffer (device-visible host memory)
// and then releases a sequence word at system scope; the host polls that word.  If the word does not arrive within
// kPollTimeoutUs the host falls back to the stream synchronisation (the words are there after it either way).
constexpr long long kPollTimeoutUs = 50000;

__global__ __launch_bounds__(256) void publish_words_kernel(const unsigned* __restrict__ src, unsigned* dst, int n,
                                                            unsigned seq) {
  // dst[0] = sequence word, dst[1..n] = payload
  for (int i = threadIdx.x; i < n; i += 256)
    __hip_atomic_store(dst + 1 + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline long long now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000;
}

// shared-list frames: the binning's tile_done [T] = every sample's compositor has finished the tile
__global__ void and_tile_done_kernel(int S, int T, const unsigned char* __restrict__ done_s, unsigned char* __restrict__ done) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  unsigned char a = 1;
  for (int s = 0; s < S; ++s) a &= done_s[(size_t)s * T + t] != 0;
  done[t] = a;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
}

// device words [n] -> host_pinned[1..n] (host_pinned[0] is the sequence word); returns when they are readable.
// poll: host_pinned must be host-coherent pinned memory the GPU can write at system scope (hipHostMalloc default /
// torch pin_memory); with non-coherent pinned memory (HIP_HOST_COHERENT=0) the sequence word may never become visible
// and every read-back waits out kPollTimeoutUs before falling back to hipStreamSynchronize — pass poll_readback = 0 then.
// between: launched behind the copy / publish kernel and before the host starts to wait — work for the GPU while the
// words travel and the host wakes up (gs_frame_forward: the averaging of the sample images).
template <class Between>
inline int read_back(const unsigned* src_dev, unsigned* host_pinned, int n, bool poll, hipStream_t st, Between between) {
  if (!poll) {
    hipError_t e = hipMemcpyAsync(host_pinned + 1, src_dev, 4ll * n, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return 1000 + (int)e;
    int rb = between();
    if (rb != GS_OK) return rb;
    return hip_status(hipStreamSynchronize(st));
  }
  volatile unsigned* seqw = host_pinned;
  *seqw = 0u;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(256), 0, st, src_dev, host_pinned, n, 1u);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return 1000 + (int)e;
  int rb = between();
  if (rb != GS_OK) return rb;
  const long long t0 = now_us();
  unsigned spins = 0;
  while (__atomic_load_n(host_pinned, __ATOMIC_ACQUIRE) != 1u) {
    cpu_relax();
    if ((++spins & 1023u) == 0 && now_us() - t0 > kPollTimeoutUs) {
      int r = hip_status(hipStreamSynchronize(st));
      if (r != GS_OK) return r;
      return __atomic_load_n(host_pinned, __ATOMIC_ACQUIRE) == 1u ? GS_OK : GS_ERR_INVALID;
    }
  }
  return GS_OK;
}

inline int read_back(const unsigned* src_dev, unsigned* host_pinned, int n, bool poll, hipStream_t st) {
  return read_back(src_dev, host_pinned, n, poll, st, [] { return GS_OK; });
}

#define CHECK(call)            \
  do {                         \
    int _st = (call);          \
    if (_st != GS_OK) return _st; \
  } while (0)

}  // namespace

// stage ids of gs_frame_profile_*: depth_sort, count_scan, slice_plan, slice_count, emit, tile_sort, bin_edges,
// raster_fwd, slice_sat, raster_bwd, grad_reduce
enum { ST_DEPTH_SORT = 0, ST_COUNT_SCAN, ST_PLAN, ST_COUNT, ST_EMIT, ST_TILE_SORT, ST_BIN_EDGES, ST_RASTER_FWD, ST_SAT,
       ST_RASTER_BWD, ST_REDUCE, ST_N };

GS_EXPORT int gs_frame_profile_enable(unsigned stage_mask) {
  g_profile_mask.store(stage_mask, std::memory_order_relaxed);
  return GS_OK;
}

// Drains the event pairs recorded since the last call: stage_ids[i] / ms[i] for i < returned count (<= max_events;
// the rest is dropped).  Synchronises on the recorded events.  Process-wide state behind a mutex (measurement facility).
GS_EXPORT int gs_frame_profile_read(int max_events, int* stage_ids, float* ms) {
  int n = 0;
  std::lock_guard<std::mutex> lk(g_profile_mu);
  for (StageEvents& e : g_events) {
    if (n < max_events && stage_ids && ms) {
      (void)hipEventSynchronize(e.b);
      float t = 0.f;
      (void)hipEventElapsedTime(&t, e.a, e.b);
      stage_ids[n] = e.stage;
      ms[n] = t;
      ++n;
    }
    g_pool.push_back(e);
  }
  g_events.clear();
  return n;
}

// bytes of the arena one depth slice occupies in the forward (all of it stays allocated until the frame's backward ran)
static long long slice_bytes(const gs_frame_desc& d, long long n_k, long long I_k, bool masks, long long true_total) {
  const long long P = d.P, T = (long long)((d.W + kTile - 1) / kTile) * ((d.H + kTile - 1) / kTile);
  long long b = 0;
  auto add = [&](long long bytes) { b = Arena::up(b) + bytes; };
  if (n_k > 0) {
    add(4 * n_k); add(4 * n_k);                                   // slice_gi, counts
    if (masks) { add(8 * (true_total / 64 + n_k + 2)); add(4 * n_k); }
    add(4 * n_k); add(4); add(gs_scan_workspace_bytes(n_k));       // cum_k, total_k, scan workspace
  }
  if (I_k > 0) {
    add(4 * I_k); add(4 * (I_k + kIdsPad));                       // keys, vals
    add(4 * (I_k + kIdsPad)); add(4 * I_k); add(4 * (I_k + kIdsPad));   // v0, k1, v1
    add(4 * (I_k + kIdsPad)); add(4 * (I_k + kIdsPad));           // carried payload a / b
    add(gs_radix_sort_workspace_bytes(I_k, 0, bits_for(P * T + 1)));
    add(8 * (P * T + 1));                                          // bins
  } else {
    add(4); add(8 * (d.shared_list ? (long long)d.S : P) * T);
  }
  add(4ll * d.S * d.H * d.W);                                      // final index
  return Arena::up(b) + 256;
}

GS_EXPORT long long gs_frame_backward_bytes(const gs_frame_state* state) {
  if (!state) return 0;
  long long maxI = 0;
  for (int k = 0; k < state->n_slices; ++k) maxI = std::max(maxI, state->slice[k].I);
  const long long tpe = state->shared_list ? state->S : 1;          // gradient tuples per list entry
  long long b = Arena::up(48 * maxI * tpe) + Arena::up(maxI * tpe) + 512;
  if (state->n_slices > 1) b += 2 * Arena::up(4ll * state->S * state->H * state->W);
  return b;
}

// One frame, forward: depth pre-sort of the P*N (sub-pose, Gaussian) pairs, slice plan, and per depth slice exact
// counts -> [deferred SH colour] -> compact emission -> stable tile sort -> bin edges -> compositor.
// Replaces the binning + rasterize_forward part of the fork's rasterize_gaussians autograd.Function (SURVEY.md §8
// a4-a7; the Python original is ops.sliced_forward).  records / depth_keys / num_tiles_hit come from
// gs_project_fused_fwd or gs_project_pixvel_fwd; depth_keys is consumed.  Results: out_img [S,H,W,3], out_T [S,H,W]
// (caller-owned), the slice table in *state.  GS_ERR_WORKSPACE: the arena is too small, state->arena_required says what
// the frame needs as far as it is known (call again with a larger arena and FRESH projection outputs).
GS_EXPORT int gs_frame_forward(const gs_frame_desc* dp, float* records, unsigned* depth_keys, const int* num_tiles_hit,
                               const float* background, const int* band_edges, const unsigned char* band_tile_done,
                               const float* color_means, const float* color_sh, const float* color_sh_rest, int color_K,
                               int color_degree,
                               const float* color_viewmats, const float* pix_vel, const float* sample_times,
                               float* out_img, float* out_T,
                               float* out_depth, float* out_combined, void* arena_ptr, long long arena_bytes, void* host_pinned, long long host_pinned_bytes,
                               gs_frame_state* state, void* stream_) {
  if (!dp || !records || !depth_keys || !num_tiles_hit || !background || !band_edges || !out_img || !out_T ||
      !arena_ptr || !host_pinned || !state)
    return GS_ERR_INVALID;
  const gs_frame_desc d = *dp;
  // shared_list: ONE record set / depth sort / tile list (P == 1) for the S samples of a pixel-velocity frame
  const bool shared = d.shared_list != 0;
  if (d.N <= 0 || d.P <= 0 || d.S <= 0 || d.R <= 0 || d.H <= 0 || d.W <= 0 || d.P > 256 || d.S > 256) return GS_ERR_INVALID;
  if (shared ? (d.P != 1 || d.R != 1 || !pix_vel || !sample_times) : (d.P != d.S * d.R)) return GS_ERR_INVALID;
  if (d.R > 1 && !band_tile_done) return GS_ERR_INVALID;
  if ((long long)d.S * d.H * d.W >= (1ll << 30)) return GS_ERR_INVALID;   // 32-bit byte offsets into [S,H,W] (raster.hip)
  // exact per-row rolling shutter (pixel-velocity model, raster_rs.hip) and shared-list frames: the records' tile boxes
  // are swept boxes, so the lists are built from the boxes themselves (no ellipse test, no hit masks) and the rs
  // compositors run
  const bool rs = pix_vel != nullptr && (d.rolling_shutter_time != 0.f || shared);
  if (rs && d.R != 1) return GS_ERR_INVALID;
  // lazy records: the projection wrote none; every issued slice projects its own pairs' first (SE(3) sub-poses only)
  const gs_project_inputs* lazy = d.lazy_records;
  if (lazy && (pix_vel || shared)) return GS_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream_;
  const int P = d.P, N = d.N, S = d.S, R = d.R, H = d.H, W = d.W;
  const long long n = (long long)P * N;
  const int tx = (W + kTile - 1) / kTile, ty = (H + kTile - 1) / kTile;
  const long long T = (long long)tx * ty;
  const long long plan_ints = 2ll * P * kKMax + 2 * P + 1;
  if (host_pinned_bytes < 4 * (plan_ints + 1) + 64) return GS_ERR_INVALID;
  memset(state, 0, sizeof(*state));
  state->P = P; state->N = N; state->S = S; state->R = R; state->H = H; state->W = W;
  state->rolling_shutter_time = rs ? d.rolling_shutter_time : 0.f;
  state->shared_list = shared ? 1 : 0;
  Arena A(arena_ptr, arena_bytes);

  // ---- depth pre-sort (compacting, per sub-pose) + exclusive scan of the tile counts in rank order ----------------
  // Nearest-first selection (depth_select): phase 0 ranks only the pairs the first slice's budget reaches and issues
  // them as ONE slice; phase 1 — entered only if that slice leaves a tile open — ranks and plans the pairs behind them.
  // Without it there is one phase: every visible pair is ranked, the plan holds every slice.
  const bool planned = d.slice_base > 0;
  const bool select = planned && d.depth_select != 0;
  unsigned* k0s = select ? A.take<unsigned>(n) : nullptr;       // scratch keys: depth_keys stay intact for phase 1
  unsigned* v0 = A.take<unsigned>(n);
  unsigned* k1 = A.take<unsigned>(n);
  unsigned* v1 = A.take<unsigned>(n);
  unsigned* counts_r = A.take<unsigned>(n);
  unsigned* n_live = A.take<unsigned>(P);
  const int digit = d.depth_sort_digit >= 8 && d.depth_sort_digit <= 11 ? d.depth_sort_digit : 8;
  const long long sort_ws_b = gs_segmented_sort_compact_workspace_bytes(n, N, 0, 31, digit);
  char* sort_ws = A.take<char>(sort_ws_b);
  unsigned* cum = A.take<unsigned>(n);
  unsigned* total = A.take<unsigned>(1);
  const long long scan_ws_b = gs_scan_workspace_bytes(n);
  char* scan_ws = A.take<char>(scan_ws_b);
  float* live_T = A.take<float>((long long)S * H * W);
  int* sat = A.take<int>((long long)P * (ty + 1) * (tx + 1));
  const long long w64 = (tx + 63) / 64;
  unsigned long long* open_bits = A.take<unsigned long long>((long long)P * ty * w64);
  // ONE zero fill per frame: [selection workspace + its two words | tile_done of the first slice, one "tile holds an
  // opacity above the alpha clamp" flag per tile and planned slice | one "a tile is still open" word per slice]
  const long long sel_b = select ? Arena::up(gs_depth_select_workspace_bytes(P)) + 256 : 0;
  const long long flag_off = ((1 + kKMax) * P * T + 3) & ~3ll;
  char* zero_blk = A.take<char>(sel_b + flag_off + 4 * kKMax);
  unsigned char* zeros_u8 = reinterpret_cast<unsigned char*>(zero_blk) + sel_b;
  unsigned* sel_grand = reinterpret_cast<unsigned*>(zero_blk + sel_b - 256);      // frame total of bounding-box pairs
  unsigned* thr_dev = select ? A.take<unsigned>(P) : nullptr;
  int* plan_dev = A.take<int>(plan_ints);
  unsigned char* tile_done_rs = R > 1 ? A.take<unsigned char>(P * T) : nullptr;
  // shared list: the compositor's done flags are per (sample, tile); the binning's [T] is their AND
  unsigned char* tile_done_samples = shared ? A.take<unsigned char>((long long)S * T) : nullptr;
  if (!A.ok) { state->arena_required = 2 * A.off; return GS_ERR_WORKSPACE; }
  if (shared) CHECK(hip_status(hipMemsetAsync(tile_done_samples, 0, (long long)S * T, st)));
  CHECK(hip_status(hipMemsetAsync(zero_blk, 0, sel_b + flag_off + 4 * kKMax, st)));
  int* open_flags = reinterpret_cast<int*>(zeros_u8 + flag_off);
  // budget of the first slice per sub-pose = its OPEN tiles * slice_base box pairs (a band-clipped rolling-shutter
  // sub-pose owns T / R of the frame's tiles)
  const long long plan_tiles = (d.band_clipped && R > 1) ? (T + R - 1) / R : T;
  const long long base0 = plan_tiles * (long long)std::max(1, d.slice_base);

  const unsigned* sorted_gi = nullptr;
  // phase 0: everything (or the selection); phase 1: the pairs behind the selection
  // tail_cap: promise to the selective sort that no sub-pose keeps more pairs (desc.select_cap, from the last frame's
  // selection; 0: none) — checked against the counts the plan read-back brings, below
  bool bound_known = false;
  auto presort = [&](int phase, long long tail_cap) -> int {
    int res = 0;
    {
      StageScope sc(ST_DEPTH_SORT, st);
      // visible keys are positive floats: bit 31 is never set (the culled marker is dropped, not sorted)
      if (!select) {
        CHECK(gs_segmented_sort_compact_u32(n, N, depth_keys, v0, k1, v1, 0, 31, digit, 0xFFFFFFFFu, n_live,
                                            reinterpret_cast<const unsigned*>(num_tiles_hit), counts_r, sort_ws, sort_ws_b,
                                            &res, st));
      } else {
        // (once per frame: a re-sort after a broken promise keeps the bound — its workspace and the frame total it
        //  accumulates into are only zero the first time)
        if (phase == 0 && !bound_known) {
          CHECK(gs_depth_select(n, N, depth_keys, reinterpret_cast<const unsigned*>(num_tiles_hit), base0, thr_dev,
                                sel_grand, zero_blk, sel_b - 256, st));
          bound_known = true;
        }
        CHECK(gs_segmented_sort_select_u32(n, N, depth_keys, k0s, v0, k1, v1, 0, 31, digit, 0xFFFFFFFFu,
                                           phase == 0 ? nullptr : thr_dev, phase == 0 ? thr_dev : nullptr, n_live,
                                           reinterpret_cast<const unsigned*>(num_tiles_hit), counts_r, sort_ws, sort_ws_b,
                                           &res, phase == 0 ? tail_cap : 0, st));
      }
    }
    sorted_gi = res == 1 ? v1 : v0;
    StageScope sc(ST_COUNT_SCAN, st);
    CHECK(gs_exclusive_scan_segments_u32(n, N, n_live, counts_r, cum, total, scan_ws, scan_ws_b, st));
    return GS_OK;
  };
  long long tail_cap = (select && d.select_cap > 0) ? (long long)d.select_cap : 0;
  CHECK(presort(0, tail_cap));

  // ---- slice plan: the ONE read-back every frame needs ---------------------------------------------------------------
  // word 0 of the pinned buffer is the sequence word of the polled read-backs, the payload follows
  unsigned* hp = reinterpret_cast<unsigned*>(host_pinned) + 1;
  unsigned* hp_seq = reinterpret_cast<unsigned*>(host_pinned);
  const bool poll = d.poll_readback != 0;
  const long long PK = (long long)P * kKMax;
  std::vector<long long> NV(P), seg_totals(P);
  std::vector<std::vector<long long>> bnd(P), rel(P);
  long long n_total = 0, true_total = 0;
  int K = 1;
  bool use_masks = false, rs_exact = false;
  std::vector<std::vector<int>> begins, prefixes;
  std::vector<long long> n_of, I_of;
  // phase 0: the frame's plan (select: ONE slice, the selection); phase 1: the plan of the pairs behind the selection,
  // its first slice twice the budget of the selection
  auto make_plan = [&](int phase) -> int {
    if (planned) {
      StageScope sc(ST_PLAN, st);
      if (select && phase == 0)
        CHECK(gs_slice_plan_select(P, N, kKMax, cum, total, plan_dev, reinterpret_cast<unsigned*>(plan_dev + PK),
                                   reinterpret_cast<unsigned*>(plan_dev + 2 * PK), n_live,
                                   reinterpret_cast<unsigned*>(plan_dev + 2 * PK + P), sel_grand, st));
      else
        CHECK(gs_slice_plan(P, N, kKMax, cum, total, phase == 0 ? base0 : 2 * base0, plan_dev,
                            reinterpret_cast<unsigned*>(plan_dev + PK), reinterpret_cast<unsigned*>(plan_dev + 2 * PK),
                            n_live, reinterpret_cast<unsigned*>(plan_dev + 2 * PK + P), st));
      CHECK(read_back(reinterpret_cast<const unsigned*>(plan_dev), hp_seq, (int)plan_ints, poll, st));
      for (int p = 0; p < P; ++p) {
        bnd[p].resize(kKMax); rel[p].resize(kKMax);
        for (int k = 0; k < kKMax; ++k) { bnd[p][k] = hp[p * kKMax + k]; rel[p][k] = hp[PK + p * kKMax + k]; }
        seg_totals[p] = hp[2 * PK + p];
        NV[p] = std::min<long long>(N, hp[2 * PK + P + p]);
      }
      if (phase == 0) n_total = hp[plan_ints - 1];
      // (phase 1 keeps one planned slice for the selection's: tile_hot planes and open words are per issued slice)
      const int k_cap = phase == 0 ? kKMax : kKMax - 1;
      K = k_cap;
      for (int k = 0; k < k_cap; ++k) {
        bool all = true;
        for (int p = 0; p < P; ++p) all = all && bnd[p][k] >= NV[p];
        if (all) { K = k + 1; break; }
      }
    } else {
      CHECK(hip_status(hipMemcpyAsync(hp, total, 4, hipMemcpyDeviceToHost, st)));
      CHECK(hip_status(hipMemcpyAsync(hp + 1, n_live, 4 * P, hipMemcpyDeviceToHost, st)));
      CHECK(hip_status(hipStreamSynchronize(st)));
      n_total = hp[0];
      for (int p = 0; p < P; ++p) { NV[p] = std::min<long long>(N, hp[1 + p]); seg_totals[p] = 0; }
      K = 1;
    }
    true_total = 0;
    for (int p = 0; p < P; ++p) true_total += seg_totals[p];
    use_masks = planned && true_total < 4294967296ll - 64;
    // pixel-velocity compositors (rs): exact lists need the hit masks (the emission expands them; it has no swept test of
    // its own) — without them the lists hold the swept boxes whole, as in rounds 3-5
    rs_exact = rs && use_masks;
    if (rs && !rs_exact) use_masks = false;
    // slice descriptors (host arrays: they travel in the kernel arguments) and the slices' list capacities
    begins.assign(K, std::vector<int>(P));
    prefixes.assign(K, std::vector<int>(P + 1));
    n_of.assign(K, 0); I_of.assign(K, 0);
    for (int k = 0; k < K; ++k) {
      prefixes[k][0] = 0;
      long long I_k = 0;
      for (int p = 0; p < P; ++p) {
        const long long lo = k == 0 ? 0 : std::min(bnd[p].empty() ? NV[p] : bnd[p][k - 1], NV[p]);
        const long long hi = (k == K - 1) ? NV[p] : std::min(bnd[p][k], NV[p]);
        begins[k][p] = (int)((long long)p * N + lo);
        prefixes[k][p + 1] = prefixes[k][p] + (int)std::max(0ll, hi - lo);
        if (planned) {
          const long long hi_rel = (k == K - 1) ? seg_totals[p] : rel[p][k];
          const long long lo_rel = k == 0 ? 0 : rel[p][k - 1];
          I_k += (hi_rel - lo_rel) & 0xFFFFFFFFll;                  // upper bound: the ranks' bounding-box pairs
        }
      }
      n_of[k] = prefixes[k][P];
      I_of[k] = planned ? I_k : n_total;
      if (n_of[k] == 0) I_of[k] = 0;
    }
    return GS_OK;
  };
  CHECK(make_plan(0));
  if (tail_cap > 0) {
    // the promise is checked: a sub-pose that selected more pairs than the tail passes were sized for is sorted wrong
    // behind tail_cap — sort and plan again without it (one wasted sort; the caller learns the larger count below)
    long long most = 0;
    for (int p = 0; p < P; ++p) most = std::max(most, NV[p]);
    if (most > tail_cap) {
      state->select_overflow = 1;
      CHECK(presort(0, 0));
      CHECK(make_plan(0));
    }
  }
  for (int p = 0; p < P; ++p) state->max_selected = std::max<long long>(state->max_selected, select ? NV[p] : 0);
  state->n_total = n_total;
  state->depth_select = select ? 1 : 0;
  state->open_after_first = -1.f;
  // select: are there visible pairs behind the selection?  (the frame's total against the selection's, both mod 2^32
  // like every pair count of a frame)
  bool rest_behind = select && ((n_total - true_total) & 0xFFFFFFFFll) != 0;

  // what a slice needs is priced right before it is issued (a frame rarely needs all its planned slices: the headline
  // plans five and uses one); the arena must then also hold the backward's buffers for the slices issued so far
  long long maxI_issued = 0;
  auto fits = [&](long long n_s, long long I_s, int k_next) -> bool {
    if (I_s >= 2147483647ll - kIdsPad) return false;                 // caught separately below
    const long long mI = std::max(maxI_issued, I_s);
    const long long tpe = shared ? S : 1;                            // gradient tuples per list entry
    const long long bwd = d.reserve_backward ? Arena::up(48 * mI * tpe) + Arena::up(mI * tpe) + 2 * Arena::up(4ll * S * H * W) + 1024 : 0;
    const long long need = Arena::up(A.off) + slice_bytes(d, n_s, I_s, use_masks, true_total) + bwd;
    if (need <= arena_bytes) return true;
    // say what this slice AND the next planned one would take: one retry usually settles a new high-water mark
    state->arena_required = need + (k_next < K && I_of[k_next] < 2147483647ll
                                        ? slice_bytes(d, n_of[k_next], I_of[k_next], use_masks, true_total) + 49 * I_of[k_next]
                                        : 0);
    return false;
  };
  // Planned slices [k0, k1) issued as ONE slice.  The plan's budgets double because a tile usually saturates within
  // its first few hundred entries; a frame whose tiles do NOT saturate (a fitted model seen through many faint
  // splats) gains nothing from a slice boundary and pays ~0.15 ms for each (count, scan, colour, emit, two sort
  // passes, bin edges, a compositor launch with state reload, one more tuple reduce): when a slice leaves at least
  // merge_open_fraction of the tiles it started with open, the next issued slice spans twice as many planned ones.
  // Images do not depend on where the boundaries are (bit-identical); gradient sums change their order.
  auto merged = [&](int k0, int k1, std::vector<int>& beg, std::vector<int>& pre, long long& n_s, long long& I_s) {
    beg = begins[k0];
    pre.assign(P + 1, 0);
    I_s = 0;
    for (int p = 0; p < P; ++p) {
      int cnt = 0;
      for (int k = k0; k < k1; ++k) cnt += prefixes[k][p + 1] - prefixes[k][p];
      pre[p + 1] = pre[p] + cnt;
    }
    for (int k = k0; k < k1; ++k) I_s += I_of[k];
    n_s = pre[P];
    if (n_s == 0) I_s = 0;
  };

  unsigned char* tile_done = zeros_u8;                               // [P*T], all open
  const bool holes0 = R > 1;                                         // the very first slice already has closed tiles
  if (R > 1) {
    // rolling-shutter bands: sub-pose p = s*R + r only ever composites tile rows [edge[r], edge[r+1]); every other
    // tile of p is "done" from the start so the binning never emits for it
    tile_done = tile_done_rs;
    CHECK(hip_status(hipMemcpyAsync(tile_done, band_tile_done, P * T, hipMemcpyDeviceToDevice, st)));
    CHECK(gs_tile_open_sat(P, H, W, tile_done, sat, open_bits, nullptr, st));
  }
  const unsigned invalid_key = rs ? 0u : (unsigned)(P * T);
  const int key_bits = bits_for(P * T + 1);
  int n_out = 0;
  int span = 1;                                                      // planned slices per issued slice
  long long open_before = (long long)S * T;                          // tiles a compositor launch visits
  std::vector<int> beg_s, pre_s;
  int slice_no = 0;                                                  // issued-slice counter over both phases
  bool lazy_done = false;                                            // lazy records: every pair's record exists by now
  for (int phase = 0; phase < 2; ++phase) {
  long long open_left = 0;                                           // tiles the phase's last compositor left open
  for (int k = 0, k1 = 0; k < K; k = k1, ++slice_no) {
    long long n_k = 0, I_k = 0;
    k1 = std::min(K, k + span);
    merged(k, k1, beg_s, pre_s, n_k, I_k);
    while (k1 > k + 1 && I_k >= 2147483647ll - kIdsPad) merged(k, --k1, beg_s, pre_s, n_k, I_k);
    // the frame's last slice: the plan's last one — unless pairs wait behind a selection (phase 0 of a select frame)
    const bool first = slice_no == 0, last = k1 == K && !(phase == 0 && rest_behind);
    if (I_k >= 2147483647ll - kIdsPad) return GS_ERR_INVALID;        // a slice list is indexed with 31 bits: lower slice_base
    if (shared && I_k * (long long)S >= 4294967296ll) return GS_ERR_INVALID;   // 32-bit tuple offsets (entry * S)
    if (!fits(n_k, I_k, k1)) return GS_ERR_WORKSPACE;
    maxI_issued = std::max(maxI_issued, I_k);
    unsigned *slice_gi = nullptr, *counts = nullptr, *cum_k = nullptr, *total_k = nullptr, *mask_off = nullptr;
    unsigned long long* masks = nullptr;
    unsigned *vals = nullptr, *svals = nullptr, *sorted_ids = nullptr;
    int* bins = nullptr;
    unsigned char* tile_hot = zeros_u8 + (1 + slice_no) * P * T;
    int wave_per_g = 0;
    if (n_k > 0) {
      StageScope sc(ST_COUNT, st);
      slice_gi = A.take<unsigned>(n_k);
      counts = A.take<unsigned>(n_k);
      const bool have_holes = !first || holes0;
      // few Gaussians with large boxes (the nearest slice): one wave per Gaussian (boxes of up to 256 tiles are walked
      // by single lanes where a wave holds several)
      long long box_total = 0;
      if (first) {
        if (!planned) box_total = n_total;
        else if (K == 1) box_total = true_total;
        else for (int p = 0; p < P; ++p) box_total += rel[p][0];
      }
      wave_per_g = (first && box_total > 128 * n_k) ? 1 : 0;
      if (use_masks) {
        masks = A.take<unsigned long long>(true_total / 64 + n_k + 2);
        mask_off = A.take<unsigned>(n_k);
      }
      cum_k = A.take<unsigned>(n_k);
      total_k = A.take<unsigned>(1);
      const long long ws_b = gs_scan_workspace_bytes(n_k);
      char* ws = A.take<char>(ws_b);
      if (!A.ok) { state->arena_required = 2 * A.off; return GS_ERR_WORKSPACE; }
      if (lazy && !lazy_done) {
        // lazy records: the FIRST issued slice projects its own pairs (tens of thousands, by gathers); a frame that goes
        // on projects everything that is left in one coalesced pass of the eager kernel (round 6)
        if (slice_no == 0)
          CHECK(gs_slice_project_records((int)n_k, P, N, beg_s.data(), pre_s.data(), sorted_gi, lazy, H, W, records, st));
        else {
          CHECK(gs_project_records(P, N, lazy, H, W, records, st));
          lazy_done = true;
        }
      }
      if (rs_exact)
        // round 6: the swept alpha >= 1/255 ellipse instead of the whole swept box (the box lists held ~3x the entries a
        // sample's pixels can blend: every one of them cost the compositors a full four-pixel evaluation)
        CHECK(gs_slice_counts_exact_swept((int)n_k, P, N, beg_s.data(), pre_s.data(), sorted_gi, records,
                                          have_holes ? sat : nullptr, have_holes ? tile_done : nullptr, H, W, slice_gi,
                                          counts, wave_per_g, cum, masks, mask_off, have_holes ? open_bits : nullptr,
                                          pix_vel, shared ? d.sweep_t_min : 0.f, shared ? d.sweep_t_max : 0.f,
                                          d.rolling_shutter_time, st));
      else if (rs)
        CHECK(gs_slice_counts((int)n_k, P, N, beg_s.data(), pre_s.data(), sorted_gi, records,
                              have_holes ? sat : nullptr, H, W, slice_gi, counts, st));
      else
        CHECK(gs_slice_counts_exact((int)n_k, P, N, beg_s.data(), pre_s.data(), sorted_gi, records,
                                    have_holes ? sat : nullptr, have_holes ? tile_done : nullptr, H, W, slice_gi, counts,
                                    wave_per_g, masks ? cum : nullptr, masks, mask_off, have_holes ? open_bits : nullptr,
                                    nullptr, st));
      CHECK(gs_exclusive_scan_u32(n_k, counts, cum_k, total_k, ws, ws_b, st));
      if (color_means && color_sh && color_viewmats)
        // deferred SH colour for exactly the Gaussians this slice emits
        CHECK(gs_slice_colors((int)n_k, slice_gi, counts, N, color_means, color_sh, color_sh_rest, color_K, color_degree,
                              color_viewmats, records, st));
    }
    if (I_k > 0) {
      unsigned* keys = A.take<unsigned>(I_k);
      vals = A.take<unsigned>(I_k + kIdsPad);
      unsigned* s_v0 = A.take<unsigned>(I_k + kIdsPad);
      unsigned* s_k1 = A.take<unsigned>(I_k);
      unsigned* s_v1 = A.take<unsigned>(I_k + kIdsPad);
      unsigned* pa = A.take<unsigned>(I_k + kIdsPad);
      unsigned* pb = A.take<unsigned>(I_k + kIdsPad);
      const long long ws_b = gs_radix_sort_workspace_bytes(I_k, 0, key_bits);
      char* ws = A.take<char>(ws_b);
      bins = A.take<int>(2 * (P * T + 1));
      if (!A.ok) { state->arena_required = 2 * A.off; return GS_ERR_WORKSPACE; }
      {
        StageScope sc(ST_EMIT, st);
        if (rs && !rs_exact && first && !holes0)
          // every tile is open and the counts are the boxes: the slice holds exactly its ranks' box pairs
          CHECK(gs_emit_intersects(n_k, N, H, W, slice_gi, cum_k, records, I_k, keys, vals, invalid_key, st));
        else
          CHECK(gs_emit_open_intersects((int)n_k, N, H, W, slice_gi, counts, cum_k, records,
                                        (!first || holes0) ? tile_done : nullptr, keys, vals, invalid_key,
                                        (rs && !rs_exact) ? 0 : 1, wave_per_g, masks, mask_off, rs ? nullptr : tile_hot, st));
      }
      int r1 = 0, r2 = 0;
      {
        // payload = emission index e (iota); the record index of a sorted entry travels as a second payload: the
        // final pass leaves it in sorted order for the scalar-cache compositors
        StageScope sc(ST_TILE_SORT, st);
        CHECK(gs_radix_sort_pairs_carry_u32(I_k, keys, s_v0, s_k1, s_v1, 1, 0, key_bits, ws, ws_b, &r1, vals, pa, pb, &r2,
                                            total_k, st));
      }
      const unsigned* skeys = r1 == 1 ? s_k1 : keys;
      svals = r1 == 1 ? s_v1 : s_v0;
      sorted_ids = r2 == 1 ? pb : pa;
      {
        StageScope sc(ST_BIN_EDGES, st);
        CHECK(gs_tile_bin_edges_u32(I_k, skeys, (int)(P * T + 1), bins, total_k, st));   // last row: culled pairs
      }
    }
    if (I_k == 0 && !(first || last)) continue;
    if (I_k == 0) {
      // (an empty first / last slice goes through the plain compositor, which looks up one bin per (sample, tile))
      const long long nb = (shared ? (long long)S : (long long)P) * T;
      svals = A.take<unsigned>(1 + kIdsPad);
      bins = A.take<int>(2 * nb);
      if (!A.ok) { state->arena_required = 2 * A.off; return GS_ERR_WORKSPACE; }
      CHECK(hip_status(hipMemsetAsync(svals, 0, 4 * (1 + kIdsPad), st)));
      CHECK(hip_status(hipMemsetAsync(bins, 0, 8 * nb, st)));
    }
    int* fidx = A.take<int>((long long)S * H * W);
    if (!A.ok) { state->arena_required = 2 * A.off; return GS_ERR_WORKSPACE; }
    // compositor form: fwd_variant 3 = the 4x4-block lock-step walk (raster.hip), measured on the fitted-model-like
    // scene at 2.46 ms against 2.49 ms for the all-pixels walk (run r3_run9): the 1.43x fewer steps are paid back by
    // the per-chunk block masks + list building and the costlier step — so it is never chosen automatically
    const int fwd_variant = (I_k == 0 && d.fwd_variant == 3) ? 0 : d.fwd_variant;
    {
      StageScope sc(ST_RASTER_FWD, st);
      if (rs && I_k > 0)
        CHECK(gs_rasterize_fwd_rs_slice(records, bins, band_edges, background, S, H, W, out_img, out_T, live_T, fidx,
                                        shared ? tile_done_samples : tile_done, first ? 1 : 0, last ? 1 : 0,
                                        reinterpret_cast<const int*>(sorted_ids), (int)std::min(n, 2147483647ll), out_depth,
                                        last ? nullptr : open_flags + slice_no, pix_vel, N, d.rolling_shutter_time,
                                        shared ? sample_times : nullptr, st));
      else
      CHECK(gs_rasterize_fwd_slice(records, reinterpret_cast<const int*>(svals), bins, band_edges, background, S, R, H, W,
                                   out_img, out_T, live_T, fidx, shared ? tile_done_samples : tile_done, first ? 1 : 0,
                                   last ? 1 : 0, I_k > 0 ? reinterpret_cast<const int*>(vals) : nullptr,
                                   reinterpret_cast<const int*>(sorted_ids), I_k > 0 ? (int)std::min(n, 2147483647ll) : 0,
                                   I_k > 0 ? out_depth : nullptr, I_k > 0 ? tile_hot : nullptr,
                                   last ? nullptr : open_flags + slice_no, fwd_variant, st));
    }
    if (I_k > 0) {
      if (n_out >= GS_FRAME_MAX_SLICES) return GS_ERR_INVALID;
      gs_frame_slice& sl = state->slice[n_out++];
      sl.I = I_k; sl.n = (int)n_k; sl.wave_per_gaussian = wave_per_g; sl.first = first; sl.last = last;
      sl.svals = A.offset_of(svals); sl.bins = A.offset_of(bins); sl.fidx = A.offset_of(fidx);
      sl.gi_of_e = A.offset_of(vals); sl.sorted_ids = A.offset_of(sorted_ids); sl.slice_gi = A.offset_of(slice_gi);
      sl.counts = A.offset_of(counts); sl.cum = A.offset_of(cum_k); sl.tile_hot = A.offset_of(tile_hot);
      sl.n_emitted_dev = A.offset_of(total_k);
    }
    // out_combined: the gamma-space average of the sample images, launched behind EVERY slice's compositor — behind the
    // open-tile word when one is read, so that it runs while the word travels and the host wakes up (26 us that used to
    // sit on the critical path between the forward and the backward); a slice that turns out not to be the last one
    // has its average overwritten by the next
    auto average = [&]() -> int {
      if (!out_combined) return GS_OK;
      return gs_combine_fwd(S, 3ll * H * W, out_img, d.combine_gamma, d.combine_min_level, out_combined, st);
    };
    if (last) CHECK(average());
    if (!last) {
      // one read-back per slice: are there open tiles for the next planned slice?  One word, written by the compositor
      // itself, read AFTER this slice's whole pipeline was issued
      CHECK(read_back(reinterpret_cast<const unsigned*>(open_flags + slice_no), hp_seq, 1, poll, st, average));
      const long long open_now = hp[0];                              // tiles the compositor left open
      open_left = open_now;
      if (slice_no == 0) state->open_after_first = (float)((double)open_now / (double)((long long)S * T));
      if (open_now == 0) { ++slice_no; break; }
      span = (d.merge_open_fraction > 0.f && (double)open_now >= (double)d.merge_open_fraction * (double)open_before)
                 ? 2 * (k1 - k) : 1;
      open_before = open_now;
      StageScope sc(ST_SAT, st);
      if (shared)
        hipLaunchKernelGGL(and_tile_done_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, S, (int)T,
                           tile_done_samples, tile_done);
      CHECK(gs_tile_open_sat(P, H, W, tile_done, sat, open_bits, nullptr, st));
    }
  }
  // phase 1 of a select frame: its one slice left tiles open and pairs wait behind the selection — rank and plan them
  if (!(phase == 0 && rest_behind && open_left > 0)) break;
  state->depth_select = 2;
  rest_behind = false;
  span = 1;
  CHECK(presort(1, 0));
  CHECK(make_plan(1));
  }
  state->n_slices = n_out;
  state->arena_used = Arena::up(A.off);
  return GS_OK;
}

// One frame, backward of the compositing: the slices in reverse, gradient tuples + segmented reduce into v_records
// [P*N,12] (plain stores; rows the compositor never touched are left as they are) and touched [P*N] (zeroed by the
// caller).  Replaces the rasterize_backward part of the fork's autograd.Function (Python original:
// ops.sliced_backward).  The arena is the forward's; this call allocates behind state->arena_used.
GS_EXPORT int gs_frame_backward(const gs_frame_state* state, const float* records, const float* background,
                                const int* band_edges, const float* out_T, const float* v_img, const float* v_alpha,
                                const float* cmb_scale, float cmb_gamma, float cmb_min_level, int bwd_variant,
                                float* v_records, unsigned char* touched, const float* pix_vel,
                                const float* sample_times, void* arena_ptr, long long arena_bytes, void* stream_) {
  if (!state || !records || !background || !band_edges || !out_T || !v_img || !v_records || !arena_ptr)
    return GS_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream_;
  const int S = state->S, R = state->R, H = state->H, W = state->W;
  const long long n_rec = (long long)state->P * state->N;
  Arena A(arena_ptr, arena_bytes, state->arena_used);
  char* base = (char*)arena_ptr;
  long long maxI = 0;
  for (int k = 0; k < state->n_slices; ++k) maxI = std::max(maxI, state->slice[k].I);
  // reverse-traversal state between slices: running T and (behind-colour . v_out), ONE float per pixel each; a frame
  // that needed a single slice (the common case) carries none
  float *bwd_T = nullptr, *bwd_B = nullptr;
  if (state->n_slices > 1) {
    bwd_T = A.take<float>((long long)S * H * W);
    bwd_B = A.take<float>((long long)S * H * W);
  }
  // ONE tuple buffer for all slices: a slice's tuples are reduced before the next (nearer) slice writes its own
  const bool shared = state->shared_list != 0;
  if (shared && (!pix_vel || !sample_times)) return GS_ERR_INVALID;
  const long long tpe = shared ? S : 1;                               // gradient tuples per list entry
  float* tuples = A.take<float>(12 * maxI * tpe);
  unsigned char* flags = A.take<unsigned char>((maxI * tpe + 15) & ~15ll);      // (whole 16-byte words: see the fill below)
  if (!A.ok) return GS_ERR_WORKSPACE;
  if (bwd_T) {
    CHECK(hip_status(hipMemcpyAsync(bwd_T, out_T, 4ll * S * H * W, hipMemcpyDeviceToDevice, st)));
    CHECK(hip_status(hipMemsetAsync(bwd_B, 0, 4ll * S * H * W, st)));
  }
  for (int k = state->n_slices - 1; k >= 0; --k) {
    const gs_frame_slice& sl = state->slice[k];
    // (a byte count that is a multiple of 16 is ONE fill launch instead of body + tail)
    CHECK(hip_status(hipMemsetAsync(flags, 0, (sl.I * tpe + 15) & ~15ll, st)));
    {
      StageScope sc(ST_RASTER_BWD, st);
      if (state->rolling_shutter_time != 0.f || shared) {
        if (!pix_vel) return GS_ERR_INVALID;
        CHECK(gs_rasterize_bwd_rs_slice(records, reinterpret_cast<const int*>(base + sl.svals),
                                        reinterpret_cast<const int*>(base + sl.bins), band_edges, background, S, H, W, out_T,
                                        reinterpret_cast<const int*>(base + sl.fidx), v_img, v_alpha, bwd_T, bwd_B, tuples,
                                        flags, reinterpret_cast<const int*>(base + sl.sorted_ids),
                                        (int)std::min(n_rec, 2147483647ll), bwd_variant & 256, cmb_scale, cmb_gamma,
                                        cmb_min_level, pix_vel, state->N, state->rolling_shutter_time,
                                        shared ? sample_times : nullptr, st));
      } else
      CHECK(gs_rasterize_bwd_slice(records, reinterpret_cast<const int*>(base + sl.svals),
                                   reinterpret_cast<const int*>(base + sl.bins), band_edges, background, S, R, H, W, out_T,
                                   reinterpret_cast<const int*>(base + sl.fidx), v_img, v_alpha, bwd_T, bwd_B, v_records,
                                   reinterpret_cast<const int*>(base + sl.gi_of_e), tuples, flags,
                                   reinterpret_cast<const int*>(base + sl.sorted_ids), (int)std::min(n_rec, 2147483647ll),
                                   reinterpret_cast<const unsigned char*>(base + sl.tile_hot), bwd_variant, cmb_scale,
                                   cmb_gamma, cmb_min_level, st));
    }
    {
      StageScope sc(ST_REDUCE, st);
      // kernel form (gs_reduce_grad_tuples takes the wave-per-Gaussian kernel when n_isect > 32 * n_slice): a wave per
      // Gaussian pays when a Gaussian owns tens of tuples AND the slice holds few Gaussians — the benchmark scene's
      // first slice (50 k Gaussians, 170 tuples each), a band-clipped rolling-shutter slice (88 k, 19 each: 0.023 ms
      // against 0.092 ms in the thread form, visit r5_prof) — and costs milliseconds on a slice of millions of small
      // splats (fitted-model-like scene: 5 M Gaussians, 8 tuples each).  The list capacity is all the host knows: the
      // exact count's own choice (few Gaussians with boxes of > 128 tiles), or a band-clipped slice of < 2^19 Gaussians.
      CHECK(gs_reduce_grad_tuples(sl.n, reinterpret_cast<const unsigned*>(base + sl.slice_gi),
                                  reinterpret_cast<const unsigned*>(base + sl.counts),
                                  reinterpret_cast<const unsigned*>(base + sl.cum), tuples, flags, v_records, touched,
                                  (sl.wave_per_gaussian || (R > 1 && sl.n < (1 << 19))) ? sl.I : 0, records, (int)tpe,
                                  st));
    }
  }
  return GS_OK;
}
