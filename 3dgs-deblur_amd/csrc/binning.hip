// binning.hip — Gaussian -> tile binning, radix sort and tile bin edges (integer
// work; results are bit-exact against the oracle).
//
// Restates (absent fork sources): gsplat map_gaussian_to_intersects,
// the torch.sort on int64 isect ids, get_tile_bin_edges and the cumsum in
// compute_cumulative_intersects (App. A "Keys").
//
// MI355X design.  Upstream sorts I (tile<<32 | depth_bits) int64 keys: 6-7 LSD
// passes over 12-byte pairs.  Here the same total order is produced with far less
// HBM traffic:
//   1. sort the P*N (sub-pose, depth_bits) keys          (N-sized, cheap)
//   2. emit the intersections in that depth order, key = p*T + tile (u32)
//   3. ONE stable LSD sort over ceil(log2(P*T)) bits of the 8-byte pairs
//      (2 passes at 1080p up to 10 sub-poses).
// A stable bucket-by-tile of a depth-ordered stream is exactly the
// (tile, depth, gaussian-id) order of the 64-bit sort; ties (same tile, same depth
// bits) resolve to ascending Gaussian id on both routes, the deterministic
// tiebreak SURVEY §7 asks for.  The 64-bit route is kept (gs_map_gaussian_to_intersects
// + gs_radix_sort_pairs_u64) for API parity and as the cross-check in tests.
//
// The radix sort is a 3-kernel LSD pass (histogram / scan / scatter) with
// wave64 match-any ranking (ballot per digit bit), one contiguous key run per wave
// so stability needs no block-wide exchange.
#include <algorithm>
#include <atomic>
#include "gs_common.h"

namespace gs {

// ---------------------------------------------------------------------------
// exclusive scan (u32)
// ---------------------------------------------------------------------------
constexpr int kScanItems = 8;                 // per thread
constexpr int kScanBlock = 256 * kScanItems;  // per block

// exclusive prefix of v over the 256 threads of the block; total returned to all.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned& total, unsigned* lds /*[8]*/) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  unsigned w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
  unsigned woff = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
  total = w0 + w1 + w2 + w3;
  __syncthreads();
  return woff + inc - v;
}

// device-side length of a radix-sort histogram: NB rows x ceil(keys / keys_per_block) blocks, keys read on the device
struct DevLen { const unsigned* keys; unsigned keys_per_block; unsigned rows; };

__device__ __forceinline__ size_t dev_len(size_t n, const DevLen& dl) {
  if (!dl.keys) return n;
  const size_t k = *dl.keys;
  return min(n, (size_t)dl.rows * ((k + dl.keys_per_block - 1) / dl.keys_per_block));
}

// Partly filled segments (the compacting depth pre-sort leaves cnt[s] live ranks at the start of every seg_len-long
// segment): elements behind a segment's live part count as ZERO and are never read.
struct SegMask { const unsigned* cnt; size_t seg_len; };

struct BlockLive {       // which elements of one scan block are live
  size_t vend;           // uniform case: live iff i < vend
  bool mixed;            // the block straddles a segment boundary: test every element
};

__device__ __forceinline__ BlockLive block_live(size_t base, size_t n, const SegMask& sm) {
  BlockLive bl{n, false};
  if (sm.cnt) {
    const size_t last = min(base + (size_t)kScanBlock, n) - 1;
    const size_t s0 = base / sm.seg_len, s1 = last / sm.seg_len;
    if (s0 == s1) bl.vend = min(n, s0 * sm.seg_len + sm.cnt[s0]);
    else bl.mixed = true;
  }
  return bl;
}

__device__ __forceinline__ bool elem_live(size_t i, size_t n, const BlockLive& bl, const SegMask& sm) {
  if (!bl.mixed) return i < bl.vend;
  if (i >= n) return false;
  const size_t sidx = i / sm.seg_len;
  return i - sidx * sm.seg_len < (size_t)sm.cnt[sidx];
}

__global__ __launch_bounds__(256) void scan_reduce_kernel(size_t n, const unsigned* __restrict__ in,
                                                          unsigned* __restrict__ bsum, DevLen dl, SegMask sm) {
  __shared__ unsigned lds[8];
  n = dev_len(n, dl);
  size_t base = (size_t)blockIdx.x * kScanBlock;
  if (base >= n) return;
  const BlockLive bl = block_live(base, n, sm);
  if (!bl.mixed && base >= bl.vend) {
    if (threadIdx.x == 0) bsum[blockIdx.x] = 0u;
    return;
  }
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    size_t i = base + (size_t)k * 256 + threadIdx.x;
    s += elem_live(i, n, bl, sm) ? in[i] : 0u;
  }
  unsigned total;
  block_excl_scan(s, total, lds);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place; writes the grand total
__global__ __launch_bounds__(256) void scan_bsums_kernel(unsigned nb, unsigned* __restrict__ bsum,
                                                         unsigned* __restrict__ total_out) {
  __shared__ unsigned lds[8];
  unsigned carry = 0;
  for (unsigned base = 0; base < nb; base += 256) {
    unsigned i = base + threadIdx.x;
    unsigned v = i < nb ? bsum[i] : 0u;
    unsigned total;
    unsigned ex = block_excl_scan(v, total, lds);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(size_t n, const unsigned* __restrict__ in,
                                                         const unsigned* __restrict__ bsum,
                                                         unsigned* __restrict__ out) {
  __shared__ unsigned lds[8];
  // blocked arrangement: thread t owns items [t*8, t*8+8) of the block
  size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
  unsigned v[kScanItems];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    size_t i = base + k;
    v[k] = i < n ? in[i] : 0u;
    s += v[k];
  }
  unsigned total;
  unsigned ex = block_excl_scan(s, total, lds) + bsum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    size_t i = base + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
}

// scan_apply with the block-sum scan folded in: block b sums the RAW sums of the blocks before it (256 threads,
// stride 256) instead of reading a pre-scanned array — one launch less per scan; the LSD sorts run one scan
// per digit pass, so a frame saves ~9 single-block launches.  Used while that prologue stays short.
constexpr size_t kScanFusedMaxBlocks = 8192;

__global__ __launch_bounds__(256) void scan_apply_fused_kernel(size_t n, const unsigned* __restrict__ in,
                                                               const unsigned* __restrict__ bsum_raw,
                                                               unsigned* __restrict__ out,
                                                               unsigned* __restrict__ total_out, DevLen dl,
                                                               SegMask sm) {
  __shared__ unsigned lds[8];
  n = dev_len(n, dl);
  if ((size_t)blockIdx.x * kScanBlock >= n) return;
  unsigned part = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += 256) part += bsum_raw[b];
  unsigned prefix;
  block_excl_scan(part, prefix, lds);              // total over the block = sum of the preceding block sums
  const BlockLive bl = block_live((size_t)blockIdx.x * kScanBlock, n, sm);
  size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
  const size_t last_block = n ? (n - 1) / kScanBlock : 0;
  if (sm.cnt && !bl.mixed && (size_t)blockIdx.x * kScanBlock >= bl.vend) {
    // segmented scan, a block without a live element (round 6: a nearest-first selection leaves a few live blocks per
    // segment, and writing the flat prefix behind them was 20 MB per frame): out is only defined for live ranks and at
    // every segment's first rank (what the slice plan reads) — a block inside one segment holds at most one of those
    const size_t b0 = (size_t)blockIdx.x * kScanBlock;
    const size_t first = ((b0 + sm.seg_len - 1) / sm.seg_len) * sm.seg_len;       // first segment start at or behind b0
    if (threadIdx.x == 0) {
      if (first < min(b0 + (size_t)kScanBlock, n)) out[first] = prefix;
      if (total_out && blockIdx.x == last_block) *total_out = prefix;
    }
    return;
  }
  unsigned v[kScanItems];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    size_t i = base + k;
    v[k] = elem_live(i, n, bl, sm) ? in[i] : 0u;
    s += v[k];
  }
  unsigned total;
  unsigned ex = block_excl_scan(s, total, lds) + prefix;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    size_t i = base + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
  if (total_out && blockIdx.x == last_block && threadIdx.x == 0) *total_out = prefix + total;
}

// a whole scan in ONE block (round 6): the histogram scans of the selective sort's tail passes hold a few thousand entries
// (256 digits x a handful of blocks), and two launches at their latency floor for that was 9 us per pass
constexpr size_t kScanSmallMax = 16384;
__global__ __launch_bounds__(256) void scan_small_kernel(size_t n, const unsigned* __restrict__ in, unsigned* __restrict__ out,
                                                         unsigned* __restrict__ total_out) {
  __shared__ unsigned lds[8];
  unsigned carry = 0;
  for (size_t base0 = 0; base0 < n; base0 += kScanBlock) {
    const size_t base = base0 + (size_t)threadIdx.x * kScanItems;
    unsigned v[kScanItems];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const size_t i = base + k;
      v[k] = i < n ? in[i] : 0u;
      s += v[k];
    }
    unsigned total;
    unsigned ex = block_excl_scan(s, total, lds) + carry;      // (in == out: everything of this chunk is read by now)
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const size_t i = base + k;
      if (i < n) out[i] = ex;
      ex += v[k];
    }
    carry += total;
  }
  if (total_out && threadIdx.x == 0) *total_out = carry;
}

static inline size_t scan_ws_bytes(size_t n) {
  size_t nb = (n + kScanBlock - 1) / kScanBlock;
  return (nb + 1) * sizeof(unsigned);
}

static int run_scan(size_t n, const unsigned* in, unsigned* out, unsigned* total_out, void* ws, hipStream_t st,
                    DevLen dl = DevLen{nullptr, 1u, 1u}, SegMask sm = SegMask{nullptr, 1}) {
  size_t nb = (n + kScanBlock - 1) / kScanBlock;
  if (n <= kScanSmallMax && !dl.keys && !sm.cnt) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(256), 0, st, n, in, out, total_out);
    return gs_launch_status();
  }
  unsigned* bsum = reinterpret_cast<unsigned*>(ws);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, in, bsum, dl, sm);
  if (nb <= kScanFusedMaxBlocks || dl.keys || sm.cnt) {
    hipLaunchKernelGGL(scan_apply_fused_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, in, bsum, out, total_out, dl,
                       sm);
  } else {
    hipLaunchKernelGGL(scan_bsums_kernel, dim3(1), dim3(256), 0, st, (unsigned)nb, bsum, total_out);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, in, bsum, out);
  }
  return gs_launch_status();
}

// ---------------------------------------------------------------------------
// radix sort (LSD, stable), KeyT in {u32,u64}, 32-bit payload
//
// One pass = histogram kernel (keys only) + scan + scatter kernel.  The scatter ranks keys
// with wave64 match-any (one ballot per digit bit), each wave owning a contiguous key run so
// stability needs no exchange, then REORDERS keys and payloads through LDS so that every
// digit's keys leave the block as one contiguous, coalesced run (a direct per-lane scatter
// wrote 64 different cache lines per instruction and ran at ~0.5 TB/s).
// ---------------------------------------------------------------------------
#ifndef GS_SORT_ROUNDS_U32
#define GS_SORT_ROUNDS_U32 16   // 4096 keys per block: 36 KB LDS -> 4 blocks / CU
#endif
template <typename KeyT> struct SortCfg;
template <> struct SortCfg<unsigned> { static constexpr int kRounds = GS_SORT_ROUNDS_U32; };  // keys / thread
template <> struct SortCfg<unsigned long long> { static constexpr int kRounds = 16; };  // 4096 keys / block

template <typename KeyT> constexpr int sort_block_keys() { return 256 * SortCfg<KeyT>::kRounds; }

// Segmented form: the input is `n / seg_len` independent segments of seg_len keys (seg_len == n: one
// segment); blocks never straddle a segment and the histogram is laid out [segment][digit][block], so ONE
// flat exclusive scan yields every block's global output offset (earlier segments contribute exactly
// their length).
struct SegInfo { size_t seg_len; unsigned nblk_seg; };

// Compacting segmented sort (depth pre-sort): most keys of a segment are the "culled" sentinel.  The FIRST pass
// skips them (they take no histogram count and no output slot) and leaves every segment's survivors packed at the
// segment's start, counting them in cnt_out[segment]; the later passes read that count back (cnt_in) and touch
// only the survivors.  Segments stay seg_len apart, so a block's global offset is its scanned histogram entry
// re-based from "keys before me in the whole array" to "keys before me in my segment".
struct SegDev {
  const unsigned* cnt_in;    // per-segment key count on the device (NULL: segments are full)
  unsigned* cnt_out;         // compacting pass: per-segment survivor count (written by the scatter kernel), else NULL
  const unsigned* total;     // compacting pass: grand total of the pass's histogram scan (device)
  unsigned long long skip;   // compacting pass: the key value that is dropped
  // compacting pass of a nearest-first selection (round 6, 32-bit keys): segment s only keeps keys in
  // [key_lo[s], key_hi[s]) — either bound nullable (device arrays: the bounds come out of gs_depth_select)
  const unsigned* key_lo;
  const unsigned* key_hi;
};

// which keys the compacting first pass of a block drops: the skip marker, and what lies outside its segment's range
template <typename KeyT>
struct KeepTest {
  KeyT skip, lo, hi;
  bool on, has_hi;
  __device__ __forceinline__ KeepTest(const SegDev& sd, unsigned seg)
      : skip((KeyT)sd.skip), lo(sd.key_lo ? (KeyT)sd.key_lo[seg] : (KeyT)0), hi(sd.key_hi ? (KeyT)sd.key_hi[seg] : (KeyT)0),
        on(sd.cnt_out != nullptr), has_hi(sd.key_hi != nullptr) {}
  __device__ __forceinline__ bool drop(KeyT k) const { return on && (k == skip || k < lo || (has_hi && k >= hi)); }
};

template <typename KeyT, int BITS>
__global__ __launch_bounds__(256) void radix_hist_kernel(size_t n, const KeyT* __restrict__ keys, int shift,
                                                         unsigned mask, SegInfo sg,
                                                         unsigned* __restrict__ ghist,
                                                         const unsigned* __restrict__ n_dev, SegDev sd) {
  constexpr int NB = 1 << BITS;
  constexpr int R = SortCfg<KeyT>::kRounds;
  // n_dev (nullable): the real element count lives on the device (n is then the capacity the grid was sized for);
  // blocks beyond it contribute an all-zero histogram
  if (n_dev) {
    // (unsegmented sorts only) the histogram is laid out for the blocks that really hold keys, so that the scan
    // between the two kernels costs what the real count costs, not what the capacity would
    n = min(n, (size_t)*n_dev);
    sg.nblk_seg = (unsigned)((n + 256 * R - 1) / (256 * R));
    if (blockIdx.x >= sg.nblk_seg) return;
  }
  __shared__ unsigned hist[NB];
  const unsigned seg = blockIdx.x / sg.nblk_seg, b = blockIdx.x % sg.nblk_seg;
  const size_t base = (size_t)seg * sg.seg_len + (size_t)b * (256 * R);
  size_t limit = min(n, (size_t)(seg + 1) * sg.seg_len);
  if (sd.cnt_in) limit = min(limit, (size_t)seg * sg.seg_len + sd.cnt_in[seg]);
  if (base >= limit) {                  // behind the segment's keys: an all-zero column
    for (int d = threadIdx.x; d < NB; d += 256) ghist[((size_t)seg * NB + d) * sg.nblk_seg + b] = 0u;
    return;
  }
  for (int d = threadIdx.x; d < NB; d += 256) hist[d] = 0;
  __syncthreads();
  // loads first, LDS atomics second: the compiler does not move a global load across an LDS atomic, and one
  // load + wait per round serialises R HBM latencies
  KeyT k[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    size_t i = base + (size_t)r * 256 + threadIdx.x;
    k[r] = i < limit ? keys[i] : (KeyT)0;
  }
  const int lane = lane_id();
  const KeepTest<KeyT> keep(sd, seg);
  // Same-address LDS atomics serialise (~5 cycles per lane): a digit that takes three values over a whole block
  // — the top byte of a depth key — made this kernel 2.5x slower than on uniform digits.  A wave whose first
  // round is that skewed counts every round's lanes in groups (ONE atomic per distinct digit among the first few
  // groups); the others keep the plain per-lane atomic, which is what uniform digits want.
  bool skewed;
  {
    const size_t i0 = base + threadIdx.x;
    const bool ok0 = i0 < limit && !keep.drop(k[0]);
    const unsigned d = (unsigned)(k[0] >> shift) & mask;
    const unsigned long long m = __ballot(ok0);
    const unsigned d0 = (unsigned)readlane_i((int)d, m ? __ffsll((long long)m) - 1 : 0);
    skewed = __popcll(__ballot(ok0 && d == d0)) >= 16;
  }
  if (skewed) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      size_t i = base + (size_t)r * 256 + threadIdx.x;
      const bool ok = i < limit && !keep.drop(k[r]);
      const unsigned d = (unsigned)(k[r] >> shift) & mask;
      unsigned long long todo = __ballot(ok);
#pragma unroll 1
      for (int it = 0; it < 4 && todo; ++it) {
        const int src = __ffsll((long long)todo) - 1;
        const unsigned d0 = (unsigned)readlane_i((int)d, src);
        const unsigned long long same = __ballot(ok && d == d0) & todo;
        if (lane == src) atomicAdd(&hist[d0], (unsigned)__popcll(same));
        todo &= ~same;
      }
      if ((todo >> lane) & 1ull) atomicAdd(&hist[d], 1u);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      size_t i = base + (size_t)r * 256 + threadIdx.x;
      if (i < limit && !keep.drop(k[r])) atomicAdd(&hist[(unsigned)(k[r] >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += 256) ghist[((size_t)seg * NB + d) * sg.nblk_seg + b] = hist[d];
}

// (the batched payload loads below take the 8-bit u32 kernel from 126 to ~140 VGPRs, i.e. from four to three waves per
//  SIMD; forcing it back to 128 with a launch bound spills 16 registers and was measured slower: tile sort 0.191 ms
//  against 0.186 without the bound and 0.199 before the loads were batched)
// PACK (compacting depth pre-sort, 32-bit keys): the tile count of every key rides in the free top bits of its payload
// instead of being gathered at random (gather_src[payload], one 64-byte sector per 4-byte count) by the last pass.
//   PACK = 1 (first pass, iota payload): payload = i | min(gather_src[i], cap) << pack_bits — a coalesced read;
//   PACK = 2 (last pass): vals_out = payload & index mask, gather_out = the packed count, or gather_src[index] for the
//                         few counts that reached cap = 2^(32 - pack_bits) - 1 (Gaussians covering hundreds of tiles).
// The passes in between carry the payload as it is.  PACK = 0: nothing of this is compiled in (the tile sort's kernels).
template <typename KeyT, int BITS, int PACK = 0>
__global__ __launch_bounds__(256) void radix_scatter_kernel(size_t n, const KeyT* __restrict__ keys_in,
                                                            const unsigned* __restrict__ vals_in,  // null => iota
                                                            KeyT* __restrict__ keys_out,
                                                            unsigned* __restrict__ vals_out, int shift, unsigned mask,
                                                            SegInfo sg,
                                                            const unsigned* __restrict__ ghist_scanned,
                                                            const unsigned* __restrict__ gather_src,  // nullable
                                                            unsigned* __restrict__ gather_out,
                                                            const unsigned* __restrict__ n_dev, SegDev sd,
                                                            const unsigned* __restrict__ p2_in,   // nullable: a second
                                                            unsigned* __restrict__ p2_out,        // payload per key
                                                            int pack_bits) {
  constexpr int NB = 1 << BITS;
  constexpr int R = SortCfg<KeyT>::kRounds;
  constexpr int BK = 256 * R;
  if (n_dev) {
    n = min(n, (size_t)*n_dev);
    sg.nblk_seg = (unsigned)((n + BK - 1) / BK);
    if (blockIdx.x >= sg.nblk_seg) return;
  }
  {
    const unsigned seg0 = blockIdx.x / sg.nblk_seg, blk0 = blockIdx.x % sg.nblk_seg;
    if (sd.cnt_in && (size_t)blk0 * BK >= (size_t)sd.cnt_in[seg0]) return;     // behind the segment's keys
  }
  constexpr int DPT = NB / 256;                 // digits per thread in the offset phase
  __shared__ KeyT s_keys[BK];
  __shared__ unsigned s_vals[BK];
  __shared__ unsigned cnt[4][NB];               // per-wave digit counts -> per-wave LDS offsets
  __shared__ unsigned s_dbase[NB];              // first LDS slot of each digit
  __shared__ unsigned s_gbase[NB];              // first global slot of each digit for this block
  __shared__ unsigned s_scan[8];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < 4 * NB; d += 256) (&cnt[0][0])[d] = 0;
  __syncthreads();
  const unsigned seg = blockIdx.x / sg.nblk_seg, blk = blockIdx.x % sg.nblk_seg;
  const size_t bbase = (size_t)seg * sg.seg_len + (size_t)blk * BK;
  size_t limit = min(n, (size_t)(seg + 1) * sg.seg_len);
  if (sd.cnt_in) limit = min(limit, (size_t)seg * sg.seg_len + sd.cnt_in[seg]);
  const bool compacting = sd.cnt_out != nullptr;
  const KeepTest<KeyT> keep(sd, seg);
  const size_t wbase = bbase + (size_t)wave * (R * 64);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  KeyT key[R];
  unsigned short pos[R];
  // all R loads are issued before the ranking loop: its wave barriers are scheduling barriers, so a load inside
  // it could not be hoisted and every round would wait out a full HBM latency on its own
#pragma unroll
  for (int r = 0; r < R; ++r) {
    size_t i = wbase + (size_t)r * 64 + lane;
    key[r] = i < limit ? keys_in[i] : (KeyT)0;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    size_t i = wbase + (size_t)r * 64 + lane;
    bool valid = i < limit && !keep.drop(key[r]);
    unsigned digit = (unsigned)(key[r] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      bool bit = (digit >> b) & 1u;
      unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    unsigned prefix = __popcll(peers & lt_mask);
    unsigned total = __popcll(peers);
    unsigned base = 0;
    if (valid) base = cnt[wave][digit];
    __builtin_amdgcn_wave_barrier();
    if (valid && prefix == 0) cnt[wave][digit] = base + total;
    __builtin_amdgcn_wave_barrier();
    pos[r] = (unsigned short)(base + prefix);
  }
  __syncthreads();
  // per digit: block count -> LDS base (exclusive scan over digits), per-wave offsets, global base
  {
    unsigned c[DPT], sum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      int d = threadIdx.x * DPT + j;
      c[j] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
      sum += c[j];
    }
    unsigned tot;
    unsigned ex = block_excl_scan(sum, tot, s_scan);
    if (threadIdx.x == 0) s_scan[4] = tot;          // keys this block really moves (block_excl_scan uses [0..3])
    const unsigned seg_org = (unsigned)((size_t)seg * sg.seg_len);
    unsigned gb[DPT];              // first global slot of each of this thread's digits for this block
    // the flat scan counts the keys of all earlier segments; the segment itself starts at seg*seg_len
    const size_t seg_first = (size_t)seg * NB * sg.nblk_seg;
    const unsigned seg_base = ghist_scanned[seg_first];
    if (compacting && blk == 0 && threadIdx.x == 0) {
      // survivors of this segment = what the flat scan counts between this segment's first entry and the next one's
      const unsigned nseg = gridDim.x / sg.nblk_seg;
      const unsigned next = seg + 1 < nseg ? ghist_scanned[seg_first + (size_t)NB * sg.nblk_seg] : *sd.total;
      sd.cnt_out[seg] = next - seg_base;
    }
#pragma unroll
    for (int j = 0; j < DPT; ++j)
      gb[j] = ghist_scanned[seg_first + (size_t)(threadIdx.x * DPT + j) * sg.nblk_seg + blk] - seg_base + seg_org;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      int d = threadIdx.x * DPT + j;
      s_dbase[d] = ex;
      s_gbase[d] = gb[j];
      unsigned run = ex;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned cw = cnt[w][d];
        cnt[w][d] = run;
        run += cw;
      }
      ex += c[j];
    }
  }
  __syncthreads();
  // stage keys + payloads in LDS in sorted-by-digit order.  The payload loads are issued TOGETHER, unconditionally
  // (clamped index), four at a time, before the LDS writes that use them: inside the per-key branch every load waited
  // for its own round trip (s_waitcnt vmcnt(0) sixteen times per thread — most of a pass's ~25 us latency floor).
  constexpr int RB = R >= 4 ? 4 : R;        // loads in flight per batch (more of them cost the kernel a wave per SIMD)
  unsigned pv[RB];
#pragma unroll
  for (int r0 = 0; r0 < R; r0 += RB) {
    if (PACK == 1 || vals_in) {
      const unsigned* __restrict__ src = PACK == 1 ? gather_src : vals_in;
      const size_t last = limit ? limit - 1 : 0;
#pragma unroll
      for (int q = 0; q < RB; ++q) pv[q] = src[min(wbase + (size_t)(r0 + q) * 64 + lane, last)];
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int r = r0 + q;
      size_t i = wbase + (size_t)r * 64 + lane;
      if (i < limit && !keep.drop(key[r])) {
        unsigned digit = (unsigned)(key[r] >> shift) & mask;
        unsigned slot = cnt[wave][digit] + pos[r];
        s_keys[slot] = key[r];
        if (PACK == 1) s_vals[slot] = (unsigned)i | (min(pv[q], (1u << (32 - pack_bits)) - 1u) << pack_bits);
        else s_vals[slot] = vals_in ? pv[q] : (unsigned)i;
      }
    }
  }
  __syncthreads();
  const unsigned nvalid = s_scan[4];
#pragma unroll 4
  for (int r = 0; r < R; ++r) {
    unsigned slot = (unsigned)r * 256 + threadIdx.x;
    if (slot < nvalid) {
      KeyT k = s_keys[slot];
      unsigned digit = (unsigned)(k >> shift) & mask;
      size_t dst = (size_t)s_gbase[digit] + (slot - s_dbase[digit]);
      keys_out[dst] = k;
      const unsigned v = s_vals[slot];
      if (PACK == 2) {
        const unsigned idx = v & ((1u << pack_bits) - 1u), c = v >> pack_bits;
        vals_out[dst] = idx;
        gather_out[dst] = c == (1u << (32 - pack_bits)) - 1u ? gather_src[idx] : c;
      } else {
        vals_out[dst] = v;
        // final pass of the tile sort: also leave gather_src[payload] in sorted order (the record index of every
        // sorted entry, fetched here among the scatter's own latencies instead of in a separate pass over the list)
        if (gather_out) gather_out[dst] = gather_src[v];
      }
    }
  }
  if (p2_out) {
    // the second payload takes the same trip through LDS (s_vals is free again once everyone has read it):
    // sequential traffic instead of a random gather by payload afterwards
    __syncthreads();
    // (loaded here, a batch at a time: held since the top of the kernel they cost 16 VGPRs through the ranking loop
    //  and a fourth of the occupancy)
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += RB) {
      const size_t last = limit ? limit - 1 : 0;
#pragma unroll
      for (int q = 0; q < RB; ++q) pv[q] = p2_in[min(wbase + (size_t)(r0 + q) * 64 + lane, last)];
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = r0 + q;
        size_t i = wbase + (size_t)r * 64 + lane;
        if (i < limit && !keep.drop(key[r])) {
          unsigned digit = (unsigned)(key[r] >> shift) & mask;
          s_vals[cnt[wave][digit] + pos[r]] = pv[q];
        }
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
      unsigned slot = (unsigned)r * 256 + threadIdx.x;
      if (slot < nvalid) {
        unsigned digit = (unsigned)(s_keys[slot] >> shift) & mask;
        p2_out[(size_t)s_gbase[digit] + (slot - s_dbase[digit])] = s_vals[slot];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The tail of a compacting segmented sort whose survivors are FEW (a nearest-first selection: ~11 k of 1 M keys per
// sub-pose on the headline): ONE block per segment finishes the sort — every key bit from begin_bit up — inside its
// registers and LDS, in one launch.  The three hist + scan + scatter passes it replaces were nine launches of ~25
// blocks each: 83 us of launch latency around microseconds of work.
//   * the block reads its segment's n = cnt_in[seg] survivors (n > kTailLocalCap: it leaves the segment alone — the
//     caller's promise was broken and it must sort again, exactly as with the tail_cap of the multi-block passes);
//   * wave w owns elements [w * Re * 64, (w + 1) * Re * 64), Re = ceil(n / 1024) rounds of 64: the same contiguous,
//     round-major ownership as radix_scatter_kernel, so the same ballot ranking is stable without an exchange;
//   * the digits are cut from (key - base) with base = the block's smallest key rounded down to 2^begin_bit (the passes
//     below begin_bit ordered the low bits of the KEY; a multiple of 2^begin_bit leaves them alone): a selection's keys
//     span a narrow depth range, so 24 key bits are typically 17-18 significant ones — two passes of 9 bits, not three;
//   * the payloads live in one 4-byte staging array between the passes; a pass lifts them into registers, sends the
//     keys through the array to their sorted slots and back, and drops the payloads into theirs (LDS holds 24 k words
//     + the per-wave digit counters, not two copies of the pairs; registers hold keys + payloads only transiently);
//   * the last step is the PACK = 2 epilogue of radix_scatter_kernel (index / packed count split).
// The result is the stable sort by key bits [begin_bit, end_bit): bit-identical to the multi-block passes.
// ---------------------------------------------------------------------------
constexpr int kTailR = 24;                               // keys per thread
constexpr int kTailLocalCap = kTailR * 1024;             // 24576 survivors per segment
constexpr int kTailMaxBits = 9;
constexpr size_t kTailLds = ((size_t)kTailLocalCap + 16 * (1 << kTailMaxBits) + 64) * sizeof(unsigned);

__global__ __launch_bounds__(1024) void seg_tail_sort_kernel(size_t seg_len, const unsigned* __restrict__ keys_in,
                                                             const unsigned* __restrict__ vals_in,
                                                             unsigned* __restrict__ keys_out,
                                                             unsigned* __restrict__ vals_out,
                                                             const unsigned* __restrict__ cnt_in, int begin_bit, int end_bit,
                                                             const unsigned* __restrict__ gather_src,
                                                             unsigned* __restrict__ gather_out, int pack_bits) {
  extern __shared__ unsigned tail_lds[];
  unsigned* const stage = tail_lds;                               // [kTailLocalCap]: the payloads between the passes
  unsigned* const cnt = stage + kTailLocalCap;                    // [16][1 << w]
  unsigned* const red = cnt + 16 * (1 << kTailMaxBits);           // [64]
  const unsigned seg = blockIdx.x;
  const unsigned n = cnt_in[seg];
  if (n == 0 || n > (unsigned)kTailLocalCap) return;
  const size_t org = (size_t)seg * seg_len;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const unsigned Re = (n + 1023u) >> 10;
  const unsigned wbase = (unsigned)wave * Re * 64u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned key[kTailR];
  unsigned pos2[kTailR / 2];                                     // two 16-bit ranks / slots per register
  unsigned kmin = 0xffffffffu, kmax = 0u;
  const unsigned emask = end_bit >= 32 ? 0xffffffffu : (1u << end_bit) - 1u;    // bits from end_bit up do not order
#pragma unroll
  for (int r = 0; r < kTailR; ++r) {
    const unsigned e = wbase + (unsigned)r * 64u + lane;
    const bool ok = (unsigned)r < Re && e < n;
    key[r] = ok ? keys_in[org + e] : 0u;
    if (ok) {
      stage[e] = vals_in[org + e];            // (read back by this very thread: no barrier in between)
      kmin = min(kmin, key[r] & emask);
      kmax = max(kmax, key[r] & emask);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
    kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o));
  }
  if (lane == 0) { red[wave] = kmin; red[16 + wave] = kmax; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) { kmin = min(kmin, red[w]); kmax = max(kmax, red[16 + w]); }
  const unsigned base = kmin & ~((1u << begin_bit) - 1u);
  const unsigned span = (kmax - base) >> begin_bit;
  const int rem = span ? 32 - __clz((int)span) : 0;               // significant bits left to order
  const int passes = (rem + kTailMaxBits - 1) / kTailMaxBits;
  const int wbits = passes ? (rem + passes - 1) / passes : 0;
  const unsigned nb = 1u << wbits, mask = nb - 1u;
  int shift = begin_bit;
  for (int p = 0; p < passes; ++p, shift += wbits) {
    for (unsigned d = threadIdx.x; d < 16u * nb; d += 1024u) cnt[d] = 0u;
    __syncthreads();
    unsigned* const wcnt = cnt + (unsigned)wave * nb;
#pragma unroll
    for (int r = 0; r < kTailR; ++r) {
      if ((unsigned)r < Re) {                                     // (uniform over the block)
        const unsigned e = wbase + (unsigned)r * 64u + lane;
        const bool valid = e < n;
        const unsigned digit = (((key[r] & emask) - base) >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kTailMaxBits; ++b) {
          if (b < wbits) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
          }
        }
        const unsigned prefix = __popcll(peers & lt_mask), total = __popcll(peers);
        unsigned c0 = 0;
        if (valid) c0 = wcnt[digit];
        __builtin_amdgcn_wave_barrier();
        if (valid && prefix == 0) wcnt[digit] = c0 + total;
        __builtin_amdgcn_wave_barrier();
        pos2[r >> 1] = (r & 1) ? (pos2[r >> 1] | ((c0 + prefix) << 16)) : (c0 + prefix);
      }
    }
    __syncthreads();
    {
      // digit d = thread d: its count over the 16 waves -> exclusive scan over the digits -> every wave's first slot
      const unsigned d = threadIdx.x;
      unsigned c = 0;
      if (d < nb) {
#pragma unroll
        for (int w = 0; w < 16; ++w) c += cnt[(unsigned)w * nb + d];
      }
      unsigned inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      if (lane == 63) red[32 + wave] = inc;
      __syncthreads();
      unsigned run = inc - c;
      for (int w = 0; w < wave; ++w) run += red[32 + w];
      if (d < nb) {
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          const unsigned cw = cnt[(unsigned)w * nb + d];
          cnt[(unsigned)w * nb + d] = run;
          run += cw;
        }
      }
    }
    __syncthreads();
    // the payloads leave the staging array for registers, the keys take the trip to their sorted slots and back to
    // their new owners, the payloads follow and STAY in the staging array (slot e = its owner's next-pass position)
    unsigned val[kTailR];
#pragma unroll
    for (int r = 0; r < kTailR; ++r) {
      const unsigned e = wbase + (unsigned)r * 64u + lane;
      if ((unsigned)r < Re && e < n) {
        val[r] = stage[e];
        const unsigned rank = (r & 1) ? pos2[r >> 1] >> 16 : pos2[r >> 1] & 0xffffu;
        const unsigned slot = wcnt[(((key[r] & emask) - base) >> shift) & mask] + rank;
        pos2[r >> 1] = (r & 1) ? ((pos2[r >> 1] & 0xffffu) | (slot << 16)) : ((pos2[r >> 1] & 0xffff0000u) | slot);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTailR; ++r) {
      const unsigned e = wbase + (unsigned)r * 64u + lane;
      if ((unsigned)r < Re && e < n) stage[(r & 1) ? pos2[r >> 1] >> 16 : pos2[r >> 1] & 0xffffu] = key[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTailR; ++r) {
      const unsigned e = wbase + (unsigned)r * 64u + lane;
      if ((unsigned)r < Re && e < n) key[r] = stage[e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTailR; ++r) {
      const unsigned e = wbase + (unsigned)r * 64u + lane;
      if ((unsigned)r < Re && e < n) stage[(r & 1) ? pos2[r >> 1] >> 16 : pos2[r >> 1] & 0xffffu] = val[r];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < kTailR; ++r) {
    const unsigned e = wbase + (unsigned)r * 64u + lane;
    if ((unsigned)r < Re && e < n) {
      keys_out[org + e] = key[r];
      const unsigned v = stage[e];
      if (pack_bits) {
        const unsigned idx = v & ((1u << pack_bits) - 1u), c = v >> pack_bits;
        vals_out[org + e] = idx;
        gather_out[org + e] = c == (1u << (32 - pack_bits)) - 1u ? gather_src[idx] : c;
      } else {
        vals_out[org + e] = v;
        if (gather_out) gather_out[org + e] = gather_src[v];
      }
    }
  }
}

template <typename KeyT>
static inline unsigned sort_nblk(size_t n) {
  return (unsigned)((n + sort_block_keys<KeyT>() - 1) / sort_block_keys<KeyT>());
}
// total blocks of a segmented sort (segments of seg_len keys; seg_len == 0 or >= n: one segment)
template <typename KeyT>
static inline unsigned sort_nblk_seg(size_t n, size_t seg_len, unsigned* nblk_seg) {
  if (seg_len == 0 || seg_len >= n) { *nblk_seg = sort_nblk<KeyT>(n); return *nblk_seg; }
  *nblk_seg = sort_nblk<KeyT>(seg_len);
  return (unsigned)((n + seg_len - 1) / seg_len) * *nblk_seg;
}

// digit plan for sorting `bits` key bits: fewest passes with digits <= 11 bits, equal widths
static inline void radix_plan(int bits, int* passes, int* per, int* tmpl_bits, int max_digit = 11) {
  if (max_digit < 8) max_digit = 8;
  if (max_digit > 11) max_digit = 11;
  int ps = (bits + max_digit - 1) / max_digit;
  if (ps < 1) ps = 1;
  int w = (bits + ps - 1) / ps;
  *passes = ps; *per = w; *tmpl_bits = w < 8 ? 8 : w;
}

template <typename KeyT>
static inline size_t radix_hist_bytes(size_t n, size_t seg_len, int bits, int max_digit = 11) {
  int ps, per, tb;
  radix_plan(bits, &ps, &per, &tb, max_digit);
  unsigned nbs;
  size_t b = ((size_t)1 << tb) * sort_nblk_seg<KeyT>(n, seg_len, &nbs) * sizeof(unsigned);
  return (b + 255) & ~(size_t)255;
}

template <typename KeyT>
static inline size_t radix_ws_bytes(size_t n, size_t seg_len, int bits, int max_digit = 11) {
  int ps, per, tb;
  radix_plan(bits, &ps, &per, &tb, max_digit);
  unsigned nbs;
  return radix_hist_bytes<KeyT>(n, seg_len, bits, max_digit) +
         scan_ws_bytes(((size_t)1 << tb) * sort_nblk_seg<KeyT>(n, seg_len, &nbs)) + 256;
}

// (A radix pass is three kernels: histogram, scan, scatter.  The single-pass form — decoupled look-back over per-block
//  digit counts — was built in round 3 and measured slower at this pipeline's sizes: tile sort 0.185 -> 0.213 ms, a pass
//  being one to three waves of co-resident blocks whose look-back wait sits on the critical path; removed in round 6,
//  profiles/r03_run15_*, r03_run16_* are the record.)
template <typename KeyT, int BITS>
static void radix_pass(size_t n, size_t seg_len, const KeyT* kin, const unsigned* vin, KeyT* kout, unsigned* vout,
                       int shift, unsigned mask, void* ws, size_t hist_bytes, hipStream_t st,
                       const unsigned* gather_src = nullptr, unsigned* gather_out = nullptr,
                       const unsigned* n_dev = nullptr, SegDev sd = SegDev{nullptr, nullptr, nullptr, 0ull, nullptr, nullptr},
                       const unsigned* p2_in = nullptr, unsigned* p2_out = nullptr, int pack = 0, int pack_bits = 0,
                       unsigned tail_blocks = 0) {
  SegInfo sg;
  unsigned nblk = sort_nblk_seg<KeyT>(n, seg_len, &sg.nblk_seg);
  sg.seg_len = (seg_len == 0 || seg_len >= n) ? n : seg_len;
  if (tail_blocks && tail_blocks < sg.nblk_seg) {
    // a pass over the SURVIVORS of a compacting sort whose caller bounds them (gs_segmented_sort_select_u32 tail_cap): the
    // grids, the [digit][block] histogram and its scan cover tail_blocks blocks per segment instead of the capacity's
    // (a selection of 11 k pairs per sub-pose sat in 245 blocks' worth of launches: 32 us per pass, all of it latency)
    nblk = nblk / sg.nblk_seg * tail_blocks;
    sg.nblk_seg = tail_blocks;
  }
  unsigned* ghist = reinterpret_cast<unsigned*>(ws);
  {
    size_t hn = ((size_t)1 << BITS) * nblk;
    void* scan_ws = reinterpret_cast<char*>(ws) + hist_bytes;
    hipLaunchKernelGGL((radix_hist_kernel<KeyT, BITS>), dim3(nblk), dim3(256), 0, st, n, kin, shift, mask, sg, ghist,
                       n_dev, sd);
    run_scan(hn, ghist, ghist, sd.cnt_out ? const_cast<unsigned*>(sd.total) : nullptr, scan_ws, st,
             DevLen{n_dev, (unsigned)sort_block_keys<KeyT>(), (unsigned)(1u << BITS)});
  }
  if constexpr (BITS == 8 && sizeof(KeyT) == 4) {
    // packed tile counts (see radix_scatter_kernel): only the 8-bit passes of the compacting 32-bit depth pre-sort
    if (pack == 1) {
      hipLaunchKernelGGL((radix_scatter_kernel<KeyT, BITS, 1>), dim3(nblk), dim3(256), 0, st, n, kin, vin, kout, vout,
                         shift, mask, sg, ghist, gather_src, gather_out, n_dev, sd, p2_in, p2_out, pack_bits);
      return;
    }
    if (pack == 2) {
      hipLaunchKernelGGL((radix_scatter_kernel<KeyT, BITS, 2>), dim3(nblk), dim3(256), 0, st, n, kin, vin, kout, vout,
                         shift, mask, sg, ghist, gather_src, gather_out, n_dev, sd, p2_in, p2_out, pack_bits);
      return;
    }
  }
  hipLaunchKernelGGL((radix_scatter_kernel<KeyT, BITS>), dim3(nblk), dim3(256), 0, st, n, kin, vin, kout, vout, shift,
                     mask, sg, ghist, gather_src, gather_out, n_dev, sd, p2_in, p2_out, 0);
}

// Sort bits [begin_bit, end_bit).  Ping-pongs between (k0,v0) and (k1,v1); returns the index
// (0/1) of the buffer pair that holds the result through *result_buf.
template <typename KeyT>
static int radix_sort(size_t n, size_t seg_len, KeyT* k0, unsigned* v0, KeyT* k1, unsigned* v1, int v0_is_iota,
                      int begin_bit, int end_bit, void* ws, size_t ws_bytes, int* result_buf, hipStream_t st,
                      int max_digit = 11, const unsigned* gather_src = nullptr, unsigned* gather_out = nullptr,
                      const unsigned* n_dev = nullptr, unsigned* seg_counts = nullptr,
                      unsigned long long skip_key = 0ull, const unsigned* p2_src = nullptr, unsigned* p2_a = nullptr,
                      unsigned* p2_b = nullptr, int* result_p2 = nullptr, const KeyT* k_src = nullptr,
                      const unsigned* key_lo = nullptr, const unsigned* key_hi = nullptr, unsigned tail_blocks = 0) {
  // k_src != NULL: the FIRST pass reads its keys from k_src (left intact) instead of k0 — k0 / k1 are both scratch then
  // key_lo / key_hi (compacting sorts only, device arrays per segment, either nullable): see SegDev
  // p2_src != NULL: a second payload travels with every key (p2_src[i] belongs to input element i); pass p writes it
  // to p2_a (even p) / p2_b (odd p), *result_p2 = 0 / 1 says which of the two holds the sorted result
  // seg_counts != NULL: compacting segmented sort (see SegDev) — keys equal to skip_key are dropped by the first
  // pass, seg_counts[segment] receives the survivors, later passes only move those
  int bits = end_bit - begin_bit;
  if (bits <= 0) return GS_ERR_INVALID;
  if (ws_bytes < radix_ws_bytes<KeyT>(n, seg_len, bits, max_digit)) return GS_ERR_WORKSPACE;
  int passes, per, tb;
  radix_plan(bits, &passes, &per, &tb, max_digit);
  const size_t hist_bytes = radix_hist_bytes<KeyT>(n, seg_len, bits, max_digit);
  KeyT* kk[2] = {k0, k1};
  unsigned* vv[2] = {v0, v1};
  int cur = 0, shift = begin_bit;
  // packed tile counts: the compacting depth pre-sort with its count gather, 8-bit digits, at least two passes, and at
  // least four payload bits to spare (cap >= 15)
  int pack_bits = 0;
  if (sizeof(KeyT) == 4 && gather_src && gather_out && seg_counts && v0_is_iota && !p2_src &&
      tb == 8 && passes >= 2) {
    while (((size_t)1 << pack_bits) < n) ++pack_bits;
    if (pack_bits > 28) pack_bits = 0;
  }
  for (int p = 0; p < passes; ++p) {
    int w = per;
    if (shift + w > end_bit) w = end_bit - shift;
    const unsigned mask = (1u << w) - 1u;
    const unsigned* vin = (p == 0 && v0_is_iota) ? nullptr : vv[cur];
    const int pack = pack_bits ? (p == 0 ? 1 : (p == passes - 1 ? 2 : 0)) : 0;
    const unsigned* gs_ = (p == passes - 1 || pack == 1) ? gather_src : nullptr;
    unsigned* go_ = (p == passes - 1) ? gather_out : nullptr;
    SegDev sd{nullptr, nullptr, nullptr, 0ull};
    if (seg_counts) {
      // the pass's grand total lives in the slack at the end of the workspace
      unsigned* total_dev = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) +
                                                        radix_ws_bytes<KeyT>(n, seg_len, bits, max_digit) - 128);
      if (p == 0) sd = SegDev{nullptr, seg_counts, total_dev, skip_key, key_lo, key_hi};
      else sd = SegDev{seg_counts, nullptr, nullptr, 0ull, nullptr, nullptr};
    }
    const unsigned* p2i = nullptr;
    unsigned* p2o = nullptr;
    if (p2_src) {
      p2o = (p & 1) ? p2_b : p2_a;
      p2i = p == 0 ? p2_src : ((p & 1) ? p2_a : p2_b);
    }
    const KeyT* kin = (p == 0 && k_src) ? k_src : kk[cur];
    switch (tb) {
      case 8:  radix_pass<KeyT, 8>(n, seg_len, kin, vin, kk[cur ^ 1], vv[cur ^ 1], shift, mask, ws, hist_bytes, st, gs_, go_, n_dev, sd, p2i, p2o, pack, pack_bits, p > 0 ? tail_blocks : 0u); break;
      case 9:  radix_pass<KeyT, 9>(n, seg_len, kin, vin, kk[cur ^ 1], vv[cur ^ 1], shift, mask, ws, hist_bytes, st, gs_, go_, n_dev, sd, p2i, p2o, 0, 0, p > 0 ? tail_blocks : 0u); break;
      case 10: radix_pass<KeyT, 10>(n, seg_len, kin, vin, kk[cur ^ 1], vv[cur ^ 1], shift, mask, ws, hist_bytes, st, gs_, go_, n_dev, sd, p2i, p2o, 0, 0, p > 0 ? tail_blocks : 0u); break;
      default: radix_pass<KeyT, 11>(n, seg_len, kin, vin, kk[cur ^ 1], vv[cur ^ 1], shift, mask, ws, hist_bytes, st, gs_, go_, n_dev, sd, p2i, p2o, 0, 0, p > 0 ? tail_blocks : 0u); break;
    }
    shift += w;
    cur ^= 1;
    if constexpr (sizeof(KeyT) == 4) {
      // few survivors promised (tail_blocks * 4096 <= kTailLocalCap): one block per segment finishes the sort
      if (p == 0 && passes >= 2 && seg_counts && !p2_src && tail_blocks &&
          (size_t)tail_blocks * sort_block_keys<KeyT>() <= (size_t)kTailLocalCap && (pack_bits || !gather_out || gather_src)) {
        // (131 KB of dynamic LDS needs the attribute once per device; a process may drive several)
        static std::atomic<unsigned long long> lds_set{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        bool lds_ok = (lds_set.load(std::memory_order_relaxed) & bit) != 0;
        if (!lds_ok && hipFuncSetAttribute(reinterpret_cast<const void*>(seg_tail_sort_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTailLds) == hipSuccess) {
          lds_set.fetch_or(bit, std::memory_order_relaxed);
          lds_ok = true;
        }
        if (lds_ok) {
          const size_t sl = (seg_len == 0 || seg_len >= n) ? n : seg_len;
          hipLaunchKernelGGL(seg_tail_sort_kernel, dim3((unsigned)((n + sl - 1) / sl)), dim3(1024), kTailLds, st, sl,
                             reinterpret_cast<const unsigned*>(kk[cur]), vv[cur], reinterpret_cast<unsigned*>(kk[cur ^ 1]),
                             vv[cur ^ 1], seg_counts, shift, end_bit, gather_src, gather_out, pack_bits);
          *result_buf = cur ^ 1;
          return gs_launch_status();
        }
      }
    }
  }
  *result_buf = cur;
  if (result_p2) *result_p2 = (passes - 1) & 1;
  return gs_launch_status();
}

// ---------------------------------------------------------------------------
// tile counts in depth order, intersection emission, bin edges
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_counts_kernel(size_t n, const unsigned* __restrict__ sorted_gi,
                                                            const int* __restrict__ ntiles,
                                                            unsigned* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (unsigned)ntiles[sorted_gi[i]];
}

// ---------------------------------------------------------------------------
// Exact tile culling.  The tile list of a Gaussian comes from the bounding BOX of its 3-sigma
// circle (upstream rule, kept for num_tiles_hit), but a pixel can only receive a contribution if
// alpha = o*exp(-sigma) >= 1/255, i.e. sigma <= tau = ln(255*o).  A tile whose pixel-centre rectangle
// lies entirely outside that ellipse is dead weight for the compositor (about half of all pairs for
// anisotropic Gaussians).  sigma is convex, so its minimum over the rectangle is 0 if the centre is
// inside, else the smallest of the four edge minima (1-D quadratics, clamped).  A relative+absolute
// slack keeps the test conservative against the compositor's float32 rounding: culled pairs
// contribute exactly nothing, so images are unchanged.
// ---------------------------------------------------------------------------
struct Ellipse { float gx, gy, a, b, c, tau, ra, rc, ustar, vstar, vext;
                 // swept form (round 6, pixel-velocity compositors): the centre moves, gx + t pvx, gy + t pvy, over the
                 // times the pixels of a tile row can see it: t in [st0 + tau(row), st1 + tau(row)], tau(y) = ((y + .5)/H - .5) srs
                 float pvx, pvy, st0, st1, srs; };
// ustar = largest |u| on the ellipse (reached at v = -b*ustar/c), vext = largest |v|, vstar = b*ustar/c

// everything the tile tests need beyond (gx, gy, a, b, c, tau): the same arithmetic wherever an ellipse is rebuilt
// from those six numbers (LDS staging, v_readlane broadcast), so every copy takes identical decisions
__device__ __forceinline__ void ellipse_derive(Ellipse& e) {
#pragma clang fp contract(off)
  e.ra = 1.0f / e.a; e.rc = 1.0f / e.c;
  // sigma = (a u^2 + 2 b u v + c v^2)/2 <= tau:  extent in u is sqrt(2 tau c / det), in v sqrt(2 tau a / det)
  const float det = e.a * e.c - e.b * e.b;
  if (e.tau >= 0.f && det > 0.f && e.a > 0.f && e.c > 0.f) {
    const float k = 2.0f * e.tau / det;
    e.ustar = sqrtf(k * e.c) * 1.0001f;
    e.vext = sqrtf(k * e.a) * 1.0001f;
    e.vstar = e.b * e.ustar * e.rc;
  } else {
    e.ustar = e.vext = 3.0e38f;       // degenerate conic: never cull
    e.vstar = 0.f;
  }
}

__device__ __forceinline__ Ellipse make_ellipse(const float* __restrict__ rec) {
#pragma clang fp contract(off)
  Ellipse e;
  e.gx = rec[0]; e.gy = rec[1]; e.a = rec[2]; e.b = rec[3]; e.c = rec[4];
  const float op = rec[5];
  e.tau = op > 0.f ? __logf(255.0f * op) : -1.f;
  // threshold with the conservative slack folded in (the tile tests compare against it directly)
  e.tau = e.tau < 0.f ? -1.f : e.tau * 1.001f + 1e-3f;
  ellipse_derive(e);
  return e;
}

// Columns of pixel centres a tile ROW can reach.  The part of the ellipse inside the band v in [v0,v1] is convex,
// so its projection on u is ONE interval [ul, ur]: a tile of that row is hit iff its pixel-centre columns intersect
// it.  ur is the ellipse's overall extreme ustar when the line v = -vstar (where it is attained) crosses the band,
// else the larger of the two chord ends at the band edges; ul likewise with +vstar.  One evaluation per row (two
// square roots) instead of four edge minima per tile; exact up to rounding, which the slack (the enlarged tau plus
// an explicit margin on the interval) keeps on the conservative side.  Contraction is disabled: the count and the
// emission must take the SAME decision for every tile wherever this is inlined.
struct RowSpan { float ul, ur; };

template <bool SWEEP = false>
__device__ __forceinline__ RowSpan row_span(const Ellipse& e, int ty, int H) {
#pragma clang fp contract(off)
  RowSpan r;
  r.ul = 1.f; r.ur = -1.f;                                 // empty
  if (e.tau < 0.f) return r;
  if (e.ustar > 1.0e37f) { r.ul = -3.0e38f; r.ur = 3.0e38f; return r; }
  float v0 = (float)(ty * K::kTile) + 0.5f - e.gy;
  float v1 = fminf((float)(ty * K::kTile + K::kTile) - 0.5f, (float)H - 0.5f) - e.gy;
  float ox0 = 0.f, ox1 = 0.f;
  if (SWEEP) {
    // SWEEP: the offsets the centre can have while a pixel of this tile row is exposed — x and y taken as independent
    // intervals (a box around the swept segment: conservative).  A pixel at (x, y) sees the splat at
    // (x - gx - ox, y - gy - oy): the row band widens by the y offsets, the column interval by the x offsets.
    const float invH = 1.0f / (float)H;
    const float ya = (float)(ty * K::kTile) + 0.5f, yb = fminf((float)(ty * K::kTile + K::kTile) - 0.5f, (float)H - 0.5f);
    const float ta = (ya * invH - 0.5f) * e.srs, tb = (yb * invH - 0.5f) * e.srs;
    const float t_lo = e.st0 + fminf(ta, tb), t_hi = e.st1 + fmaxf(ta, tb);
    const float slack = 1e-3f + 1e-5f * (fabsf(t_lo) + fabsf(t_hi)) * (fabsf(e.pvx) + fabsf(e.pvy));
    ox0 = fminf(t_lo * e.pvx, t_hi * e.pvx) - slack; ox1 = fmaxf(t_lo * e.pvx, t_hi * e.pvx) + slack;
    const float oy0 = fminf(t_lo * e.pvy, t_hi * e.pvy) - slack, oy1 = fmaxf(t_lo * e.pvy, t_hi * e.pvy) + slack;
    v0 -= oy1; v1 -= oy0;
  }
  if (v1 < -e.vext || v0 > e.vext) return r;
  const float w0 = fmaxf(v0, -e.vext), w1 = fminf(v1, e.vext);
  // chord of the ellipse on the line v = w: u = (-b w -/+ sqrt(2 a tau - det w^2)) / a
  const float det = e.a * e.c - e.b * e.b;
  const float k2 = 2.0f * e.a * e.tau;
  const float d0 = sqrtf(fmaxf(k2 - det * w0 * w0, 0.f)), d1 = sqrtf(fmaxf(k2 - det * w1 * w1, 0.f));
  const float c0 = -e.b * w0, c1 = -e.b * w1;
  float ur = fmaxf((c0 + d0) * e.ra, (c1 + d1) * e.ra);
  float ul = fminf((c0 - d0) * e.ra, (c1 - d1) * e.ra);
  if (-e.vstar >= w0 && -e.vstar <= w1) ur = e.ustar;
  if (e.vstar >= w0 && e.vstar <= w1) ul = -e.ustar;
  const float eps = 2e-3f + 2e-6f * (fabsf(e.gx) + e.ustar);
  r.ur = fmaxf(ur, ul) + eps + ox1;
  r.ul = fminf(ul, ur) - eps + ox0;
  return r;
}

// first / one-past-last tile column of row ty whose pixel centres intersect the span, clamped to [x0, x1)
__device__ __forceinline__ void span_tiles(const Ellipse& e, const RowSpan& sp, int x0, int x1, int& t0, int& t1) {
#pragma clang fp contract(off)
  if (sp.ur < sp.ul) { t0 = t1 = x0; return; }
  // tile tx holds pixel centres tx*16 + 0.5 .. tx*16 + 15.5 (the last column of the image may hold fewer: keeping
  // the full width there is conservative)
  const float lo = (sp.ul + e.gx - 15.5f) * (1.0f / (float)K::kTile);
  const float hi = (sp.ur + e.gx - 0.5f) * (1.0f / (float)K::kTile);
  const float flo = fminf(fmaxf(ceilf(lo), (float)x0), (float)x1);
  const float fhi = fminf(fmaxf(floorf(hi) + 1.0f, (float)x0), (float)x1);
  t0 = (int)flo; t1 = (int)fhi;
  if (t1 < t0) t1 = t0;
}

template <bool SWEEP = false>
__device__ __forceinline__ bool tile_hit(const Ellipse& e, int tx, int ty, int W, int H) {
  (void)W;
  const RowSpan sp = row_span<SWEEP>(e, ty, H);
  int t0, t1;
  span_tiles(e, sp, tx, tx + 1, t0, t1);
  return t1 > t0;
}

// Entry-parallel emission: a block owns kEmitChunk consecutive OUTPUT entries (so the grid scales with
// the number of intersections, not the number of Gaussians: the nearest 50k Gaussians of a slice can own
// 20M entries).  The block locates the Gaussians that cover its chunk with two binary searches in the
// exclusive scan, stages up to 256 of them in LDS, and every thread then finds the source Gaussian of
// its entries with a branch-free LDS search; tile coordinates come from a float reciprocal (exact for
// rem < 2^16) instead of integer division.  Chunks spanning more than 256 Gaussians walk them in
// windows.  Writes are perfectly coalesced.
constexpr int kEmitItems = 16;
constexpr int kEmitChunk = 256 * kEmitItems;

// last rank r in [0,n) with cum[r] <= e
__device__ __forceinline__ unsigned rank_of_entry(const unsigned* __restrict__ cum, unsigned n, unsigned e) {
  unsigned lo = 0, hi = n - 1;
  while (lo < hi) {
    unsigned mid = (lo + hi + 1) >> 1;
    if (cum[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void emit_kernel(size_t n_ranked, int N, int T, int tiles_x,
                                                   const unsigned* __restrict__ sorted_gi,
                                                   const unsigned* __restrict__ cum,   // exclusive, in rank order
                                                   const float* __restrict__ records, size_t n_isect,
                                                   unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                   int W, int H, unsigned invalid_key /*0: no exact culling*/) {
  __shared__ unsigned s_cum[257];
  __shared__ unsigned s_gi[256];
  __shared__ unsigned s_kbase[256];   // p*T
  __shared__ unsigned s_w[256];
  __shared__ unsigned s_xy0[256];
  __shared__ float s_rw[256];
  __shared__ float s_gx[256], s_gy[256], s_a[256], s_b[256], s_c[256], s_tau[256];
  __shared__ unsigned s_rng[2];
  const unsigned e_begin = blockIdx.x * (unsigned)kEmitChunk;
  const unsigned e_end = (unsigned)min((size_t)e_begin + kEmitChunk, n_isect);
  if (threadIdx.x < 2)
    s_rng[threadIdx.x] = rank_of_entry(cum, (unsigned)n_ranked, threadIdx.x == 0 ? e_begin : e_end - 1);
  __syncthreads();
  const unsigned g_lo = s_rng[0], g_hi = s_rng[1];
  for (unsigned wbase = g_lo; wbase <= g_hi; wbase += 256) {
    const unsigned r = wbase + threadIdx.x;
    unsigned gi = 0, c = 0xFFFFFFFFu, kbase = 0, w = 1, xy0 = 0;
    Ellipse el = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f, 0.f, 0.f, 0.f};
    if (r <= g_hi) {
      gi = sorted_gi[r];
      c = cum[r];
      const float* rec = records + (size_t)gi * kRecFloats;
      unsigned lo = (unsigned)__float_as_int(rec[10]);
      unsigned hi = (unsigned)__float_as_int(rec[11]);
      unsigned x0 = lo & 0xFFFFu, x1 = hi & 0xFFFFu;
      w = x1 > x0 ? x1 - x0 : 1u;
      xy0 = lo;
      kbase = (gi / (unsigned)N) * (unsigned)T;
      if (invalid_key) el = make_ellipse(rec);
    }
    __syncthreads();     // previous window fully consumed
    s_gi[threadIdx.x] = gi; s_kbase[threadIdx.x] = kbase; s_w[threadIdx.x] = w; s_rw[threadIdx.x] = 1.0f / (float)w;
    s_xy0[threadIdx.x] = xy0; s_cum[threadIdx.x] = c;
    s_gx[threadIdx.x] = el.gx; s_gy[threadIdx.x] = el.gy; s_a[threadIdx.x] = el.a; s_b[threadIdx.x] = el.b;
    s_c[threadIdx.x] = el.c; s_tau[threadIdx.x] = el.tau;
    if (threadIdx.x == 0) {
      unsigned nxt = wbase + 256;
      s_cum[256] = nxt <= g_hi ? cum[nxt] : 0xFFFFFFFFu;
    }
    __syncthreads();
    // entries of this chunk that belong to the window: [max(e_begin, cum[wbase]), min(e_end, cum[wbase+256]))
    const unsigned w_first = max(e_begin, s_cum[0]);
    const unsigned w_last = min(e_end, s_cum[256]);
    for (unsigned e = w_first + threadIdx.x; e < w_last; e += 256) {
      int a = 0;
#pragma unroll
      for (int step = 128; step > 0; step >>= 1) {
        int m = a + step;
        if (s_cum[m] <= e) a = m;          // padded slots hold 0xFFFFFFFF and are never selected
      }
      unsigned rem = e - s_cum[a];
      unsigned wa = s_w[a];
      unsigned q = (unsigned)(((float)rem + 0.5f) * s_rw[a]);
      unsigned x = rem - q * wa;
      const unsigned xy = s_xy0[a];
      const int tx = (int)((xy & 0xFFFFu) + x), ty = (int)((xy >> 16) + q);
      unsigned key = s_kbase[a] + (unsigned)(ty * tiles_x + tx);
      if (invalid_key) {
        Ellipse el2;
        el2.gx = s_gx[a]; el2.gy = s_gy[a]; el2.a = s_a[a]; el2.b = s_b[a]; el2.c = s_c[a]; el2.tau = s_tau[a];
        ellipse_derive(el2);
        if (!tile_hit(el2, tx, ty, W, H)) key = invalid_key;
      }
      keys[e] = key;
      vals[e] = s_gi[a];
    }
  }
}

// bins[t] = [start,end) of tile t in the sorted keys; one boundary test per element:
// a key change between i-1 and i closes tile key[i-1] and opens tile key[i].
// Self-zeroing (round 5): the bins of tiles WITHOUT a key — the gap in front of the first key, between two consecutive
// distinct keys, behind the last key — are written (0, 0) by the wave that sees the boundary, all 64 lanes striding over
// the gap; every bin field has exactly one writer, so the caller zeroes nothing (the hipMemsetAsync in front of this
// kernel was two fill launches per slice).  Keys must be < num_bins, as they always had to be.
template <typename KeyT, int SHIFT>
__global__ __launch_bounds__(256) void bin_edges_kernel(size_t n, const KeyT* __restrict__ keys,
                                                        int2* __restrict__ bins, unsigned num_bins,
                                                        const unsigned* __restrict__ n_dev = nullptr) {
  if (n_dev) n = min(n, (size_t)*n_dev);
  const int2 z = make_int2(0, 0);
  if (n == 0) {                       // (device-side count of zero: nothing but empty bins)
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < num_bins; b += (size_t)gridDim.x * 256) bins[b] = z;
    return;
  }
  const int lane = lane_id();
  // grid-stride over whole waves: with the count on the device the launch is sized for the CAPACITY (4x the real count
  // in the benchmark scene) — a capped grid that strides costs what the real count costs
  for (size_t base = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63u); base < n; base += (size_t)gridDim.x * 256) {
    const size_t i = base + lane;
    const bool valid = i < n;
    unsigned t = 0, lo = 0, hi = 0;         // this lane's gap of key-less bins [lo, hi)
    if (valid) {
      // a key >= num_bins (caller-supplied ids through the compat entry gs_tile_bin_edges_u64) owns no bin: it writes
      // nothing and its gaps end at num_bins (unclamped, a stray key made a whole wave zero-fill up to
      // 2^32 bins past the buffer)
      t = min((unsigned)(keys[i] >> SHIFT), num_bins);
      if (i == 0) {
        if (t < num_bins) bins[t].x = 0;
        hi = t;
      } else {
        const unsigned tp = min((unsigned)(keys[i - 1] >> SHIFT), num_bins);
        if (tp != t) {
          if (t < num_bins) bins[t].x = (int)i;
          if (tp < num_bins) bins[tp].y = (int)i;
          lo = min(tp + 1u, num_bins); hi = t;
        }
      }
      if (i == n - 1 && t < num_bins) bins[t].y = (int)n;
    }
    unsigned long long m = __ballot(hi > lo);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const unsigned g_lo = (unsigned)readlane_i((int)lo, src), g_hi = (unsigned)readlane_i((int)hi, src);
      for (unsigned b = g_lo + lane; b < g_hi; b += 64) bins[b] = z;
    }
    const unsigned long long last = __ballot(valid && i == n - 1);
    if (last) {
      const unsigned g_lo = min((unsigned)readlane_i((int)t, __ffsll((long long)last) - 1) + 1u, num_bins);
      for (unsigned b = g_lo + lane; b < num_bins; b += 64) bins[b] = z;
    }
  }
}

// the same boundary test plus, for the depth-sliced path, the record index of every sorted entry: the tile sort
// carries the EMISSION index e (the backward's tuple slot), the compositors want p*N+g = gi_of_e[e] through the
// scalar cache without a dependent gather, so it is materialised once here; ids[n .. n+8) is zero padding (the
// compositors read their lists in aligned groups of four)
__global__ __launch_bounds__(256) void bin_edges_ids_kernel(size_t n, const unsigned* __restrict__ keys,
                                                            int2* __restrict__ bins,
                                                            const unsigned* __restrict__ sorted_vals,
                                                            const unsigned* __restrict__ gi_of_e,
                                                            unsigned* __restrict__ ids) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) {
    if (i < n + 8) ids[i] = 0u;
    return;
  }
  ids[i] = gi_of_e[sorted_vals[i]];
  unsigned t = keys[i];
  if (i == 0) {
    bins[t].x = 0;
  } else {
    unsigned tp = keys[i - 1];
    if (tp != t) { bins[t].x = (int)i; bins[tp].y = (int)i; }
  }
  if (i == n - 1) bins[t].y = (int)n;
}

// ---------------------------------------------------------------------------
// depth-sliced binning (front-to-back slices; tiles whose pixels have all stopped are "done" and
// receive no further intersections).  With early termination only a few percent of the
// (Gaussian, tile) intersections are ever composited; slicing keeps the emit/sort/bin work
// proportional to what the compositor can still use.  Results are identical to the unsliced path:
// every pixel sees the same Gaussians in the same order.
// ---------------------------------------------------------------------------
// (SliceDesc / make_slice_desc / slice_rank: gs_common.h — the lazy record projection of project.hip walks the same ranks)

// slice boundaries: bounds[p*K + k] = first depth rank r of sub-pose p whose cumulative intersection count
// (cum[p*N + r] - cum[p*N], modulo 2^32) reaches base << k, and rels[p*K + k] = that cumulative count at the
// boundary (the sub-pose total when the boundary is N).  One thread per (p, k), binary search.  With the
// counts on the host the first slice needs no extra device->host sync for its size.
__global__ void slice_plan_kernel(int P, int N, int K, const unsigned* __restrict__ cum,
                                  const unsigned* __restrict__ total, unsigned long long base,
                                  int* __restrict__ bounds, unsigned* __restrict__ rels,
                                  unsigned* __restrict__ seg_totals, const unsigned* __restrict__ n_live,
                                  unsigned* __restrict__ tail /*nullable: [P] live ranks, then the grand total*/,
                                  const unsigned* __restrict__ grand /*nullable: the frame's total when `total` only
                                                                       covers a selection of its pairs*/) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * K) return;
  const int p = i / K, k = i % K;
  const unsigned* c = cum + (size_t)p * N;
  const unsigned long long tgt = base << k;
  const unsigned c0 = c[0];
  const int M = n_live ? (int)min((unsigned)N, n_live[p]) : N;     // ranks [M, N) hold nothing
  int lo = 0, hi = M;                 // first r in [0,M] with rel(r) >= tgt  (M if none)
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((unsigned long long)(unsigned)(c[mid] - c0) >= tgt) hi = mid; else lo = mid + 1;
  }
  bounds[i] = lo;
  const unsigned seg_end = (p + 1 < P) ? c[N] : *total;
  rels[i] = (lo < M ? c[lo] : seg_end) - c0;
  if (k == 0 && seg_totals) seg_totals[p] = seg_end - c0;     // the sub-pose's own total (mod 2^32)
  if (tail) {           // everything the host reads back sits in one buffer: one copy, no staging launches
    if (k == 0) tail[p] = (unsigned)M;
    if (i == 0) tail[P] = grand ? *grand : *total;
  }
}

// summed-area table of NOT-done tiles per sub-pose: sat[p][(y)*(tx+1)+x] = #open tiles in [0,y)x[0,x).
// One block per sub-pose, table built in LDS (dynamic, (tx+1)*(ty+1) ints), written out coalesced.
// open_bits (nullable) [p][tiles_y][ceil(tiles_x/64)] u64: bit x of row y set <=> tile (x, y) is still open — the exact
// count ANDs a whole tile row of a Gaussian's box against it instead of loading one byte per tile.
__global__ __launch_bounds__(256) void tile_sat_kernel(int tiles_x, int tiles_y, const unsigned char* __restrict__ done,
                                                       int* __restrict__ sat, unsigned long long* __restrict__ open_bits,
                                                       const int* __restrict__ gate) {
  extern __shared__ int s[];
  if (gate && *gate == 0) return;      // no tile is open any more: nobody will look at the table
  const int p = blockIdx.x;
  const int T = tiles_x * tiles_y, SW = tiles_x + 1, SH = tiles_y + 1;
  const unsigned char* d = done + (size_t)p * T;
  if (open_bits) {
    const int W64 = (tiles_x + 63) >> 6;
    for (int i = threadIdx.x; i < tiles_y * W64; i += 256) {
      const int y = i / W64, w = i - y * W64;
      unsigned long long m = 0ull;
      for (int b = 0; b < 64; ++b) {
        const int x = w * 64 + b;
        if (x < tiles_x && d[y * tiles_x + x] == 0) m |= 1ull << b;
      }
      open_bits[((size_t)p * tiles_y + y) * W64 + w] = m;
    }
  }
  for (int i = threadIdx.x; i < SW * SH; i += 256) {
    int y = i / SW, x = i % SW;
    s[i] = (y > 0 && x > 0) ? (d[(y - 1) * tiles_x + (x - 1)] ? 0 : 1) : 0;
  }
  __syncthreads();
  for (int y = threadIdx.x; y < SH; y += 256) {        // row prefix sums
    int run = 0;
    for (int x = 0; x < SW; ++x) { run += s[y * SW + x]; s[y * SW + x] = run; }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < SW; x += 256) {        // column prefix sums
    int run = 0;
    for (int y = 0; y < SH; ++y) { run += s[y * SW + x]; s[y * SW + x] = run; }
  }
  __syncthreads();
  int* o = sat + (size_t)p * SW * SH;
  for (int i = threadIdx.x; i < SW * SH; i += 256) o[i] = s[i];
}

// per slice Gaussian: gather its global index and count its still-open tiles (SAT query)
__global__ __launch_bounds__(256) void slice_counts_kernel(int n_slice, SliceDesc sd, int N, int tiles_x, int tiles_y,
                                                           const unsigned* __restrict__ sorted_gi,
                                                           const float* __restrict__ records,
                                                           const int* __restrict__ sat,   // null: nothing done yet
                                                           unsigned* __restrict__ slice_gi,
                                                           unsigned* __restrict__ counts) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_slice) return;
  unsigned gi = sorted_gi[slice_rank(sd, j)];
  const float* rec = records + (size_t)gi * kRecFloats;
  unsigned lo = (unsigned)__float_as_int(rec[10]), hi = (unsigned)__float_as_int(rec[11]);
  int x0 = lo & 0xFFFF, y0 = lo >> 16, x1 = hi & 0xFFFF, y1 = hi >> 16;
  unsigned c = (unsigned)((x1 - x0) * (y1 - y0));
  if (sat && c) {
    const int SW = tiles_x + 1;
    const int* s = sat + (size_t)(gi / (unsigned)N) * SW * (tiles_y + 1);
    c = (unsigned)(s[y1 * SW + x1] - s[y0 * SW + x1] - s[y1 * SW + x0] + s[y0 * SW + x0]);
  }
  slice_gi[j] = gi;
  counts[j] = c;
}

// Exact per-Gaussian counts: tiles of the box that are still open AND intersect the alpha >= 1/255
// ellipse (tile_hit).  With exact counts the emission is compact — no culled pairs reach the tile sort,
// the bin edges, the gradient tuples or their flags.  A wave owns 64 slice Gaussians; the summed-area
// table rejects Gaussians without open tiles with four loads, small boxes are walked by their own
// lane, large boxes (the nearest Gaussians cover hundreds of tiles) by the whole wave, 64 tiles per step.
constexpr int kCountSolo = 12;
// A wave with SEVERAL mid-sized boxes (13..64 tiles: one mask word) walks them in their own lanes too: taking them
// one after the other through the wave-cooperative path costs ~200 cycles each, and a scene of uniformly small
// splats (bench.py --scene trained: ~4x4 boxes) has dozens per wave — that path made the count of its later slices
// 3x slower than the rest of the binning together.
constexpr int kCountSoloMax = 64;
constexpr int kCountSoloMany = 4;
constexpr int kCountMidMax = 256;   // up to four mask words: still one lane per box when its rows fit a word (w <= 64)
constexpr int kMaskWords = 512;     // box-local hit-bit string assembled in LDS: boxes of up to 32768 tiles (4K: 32400)

// w (<= 64) bits of a tile row's open mask starting at column x0
__device__ __forceinline__ unsigned long long row_window(const unsigned long long* __restrict__ row, int x0, int w) {
  const int wi = x0 >> 6, sh = x0 & 63;
  unsigned long long v = row[wi] >> sh;
  if (sh && sh + w > 64) v |= row[wi + 1] << (64 - sh);
  return w == 64 ? v : (v & ((1ull << w) - 1ull));
}

// WAVE_PER_G: one Gaussian per wave (lane 0 owns it) — for slices of few, large Gaussians, where 64 big
// boxes per wave would serialise ~25k tile tests in each of only a few hundred waves.
// SWEEP (round 6): the pixel-velocity compositors move every splat by (sample time + row time) * pixel velocity; the
// hit test then covers every position the centre takes while a pixel of the tile row is exposed (row_span<true>), so
// the lists of those frames are culled by the swept ellipse instead of holding their whole swept bounding boxes.
struct SweepP { const float* pix_vel; float t_min, t_max, rs_time; };

template <bool WAVE_PER_G, bool SWEEP = false>
__global__ __launch_bounds__(256) void slice_counts_exact_kernel(int n_slice, SliceDesc sd, int N, int tiles_x,
                                                                 int tiles_y, const unsigned* __restrict__ sorted_gi,
                                                                 const float* __restrict__ records,
                                                                 const int* __restrict__ sat,            // nullable
                                                                 const unsigned char* __restrict__ done, // nullable
                                                                 int W, int H, unsigned* __restrict__ slice_gi,
                                                                 unsigned* __restrict__ counts,
                                                                 const unsigned* __restrict__ cum_rank,   // nullable
                                                                 unsigned long long* __restrict__ masks,  // nullable
                                                                 unsigned* __restrict__ mask_off,
                                                                 const unsigned long long* __restrict__ open_bits,
                                                                 const int* __restrict__ gate, SweepP sw) {
  __shared__ unsigned long long s_words[4][kMaskWords];
  __shared__ unsigned long long s_mid[WAVE_PER_G ? 1 : 4][256];      // lane-private bit strings of mid-sized boxes
  const int lane = lane_id();
  const int j = WAVE_PER_G ? (lane == 0 ? (int)(blockIdx.x * 4 + (threadIdx.x >> 6)) : n_slice)
                           : (int)(blockIdx.x * 256 + threadIdx.x);
  if (gate && *gate == 0) {
    // the previous slice closed the last open tile: this slice was launched without anybody waiting for that
    // answer; all it has to do is say "nothing" (everything downstream works off the counts)
    if (j < n_slice) counts[j] = 0u;
    return;
  }
  unsigned gi = 0, lo = 0, hi = 0, moff = 0;
  int area = 0;
  Ellipse el = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f, 0.f, 0.f, 0.f};
  if (j < n_slice) {
    const int rank = slice_rank(sd, j);
    gi = sorted_gi[rank];
    // hit-mask words of this Gaussian (one bit per box tile, 64 tiles per word): cum_rank is the exclusive
    // prefix of the BOX areas in rank order, so floor(cum/64) + j leaves every Gaussian ceil(area/64) words
    if (masks) { moff = (cum_rank[rank] >> 6) + (unsigned)j; mask_off[j] = moff; }
    const float* rec = records + (size_t)gi * kRecFloats;
    lo = (unsigned)__float_as_int(rec[10]); hi = (unsigned)__float_as_int(rec[11]);
    const int x0 = lo & 0xFFFF, y0 = lo >> 16, x1 = hi & 0xFFFF, y1 = hi >> 16;
    area = (x1 - x0) * (y1 - y0);
    if (area > 0 && sat) {
      const int SW = tiles_x + 1;
      const int* s = sat + (size_t)(gi / (unsigned)N) * SW * (tiles_y + 1);
      if (s[y1 * SW + x1] - s[y0 * SW + x1] - s[y1 * SW + x0] + s[y0 * SW + x0] == 0) area = 0;
    }
    if (area > 0) el = make_ellipse(rec);
    if (el.tau < 0.f) area = 0;
    if (SWEEP && area > 0) {
      const unsigned g = gi % (unsigned)N;
      el.pvx = sw.pix_vel[2 * (size_t)g]; el.pvy = sw.pix_vel[2 * (size_t)g + 1];
      el.st0 = sw.t_min; el.st1 = sw.t_max; el.srs = sw.rs_time;
    }
  }
  const unsigned T = (unsigned)(tiles_x * tiles_y);
  const int W64 = (tiles_x + 63) >> 6;
  unsigned cnt = 0;
  const int solo_limit =
      (!WAVE_PER_G && __popcll(__ballot(area > kCountSolo && area <= kCountSoloMax)) >= kCountSoloMany) ? kCountSoloMax
                                                                                                       : kCountSolo;
  // boxes of up to kCountMidMax tiles (up to four mask words) whose rows fit one 64-bit word: still ONE lane per box,
  // the box-local bit string assembled in lane-private LDS words.  Taking them through the wave-cooperative path
  // below costs ~1 us of a whole wave EACH, and a fitted-model-like scene has dozens of 65..256-tile boxes per wave
  // (mean box 56 tiles): that path was most of this kernel's 0.86 ms per frame there.
  const int box_w = (int)(hi & 0xFFFF) - (int)(lo & 0xFFFF);
  const bool mid_box = !WAVE_PER_G && area > solo_limit && area <= kCountMidMax && box_w <= 64;
  const bool mid = mid_box && __popcll(__ballot(mid_box)) >= kCountSoloMany;
  if (area > 0 && (area <= solo_limit || mid)) {
    // per tile ROW, the columns the ellipse reaches AND the open tiles, as bits
    const int x0 = lo & 0xFFFF, y0 = lo >> 16, x1 = hi & 0xFFFF, y1 = hi >> 16;
    const int w = x1 - x0;
    const unsigned pidx = gi / (unsigned)N;
    unsigned long long m = 0ull;                  // the one word of a small box
    unsigned long long* mw = &s_mid[0][threadIdx.x];          // word k of this lane: mw[k * 256]
    if (mid) { mw[0] = 0ull; mw[256] = 0ull; mw[512] = 0ull; mw[768] = 0ull; }
    auto put = [&](int y, unsigned long long rowbits) {
      cnt += (unsigned)__popcll(rowbits);
      const int bit0 = (y - y0) * w;
      if (!mid) { m |= rowbits << bit0; return; }
      const int wi = bit0 >> 6, sh = bit0 & 63;
      mw[wi * 256] |= rowbits << sh;
      if (sh && (rowbits >> (64 - sh))) mw[(wi + 1) * 256] |= rowbits >> (64 - sh);
    };
    if (open_bits) {
      // four tile rows per step, their open-tile words fetched TOGETHER and before any of them is used: one lane per
      // Gaussian walks ~8 rows, and a load per row inside the loop was one full memory round trip per row — the kernel
      // sat on ~15 serialized latencies per wave (0.2 - 0.6 ms per slice of the fitted-model-like scene)
      const int wi = x0 >> 6, sh = x0 & 63, wi1 = min(wi + 1, W64 - 1);
      const unsigned long long wmask = w == 64 ? ~0ull : ((1ull << w) - 1ull);
      const unsigned long long* base = open_bits + (size_t)pidx * tiles_y * W64;
      for (int yb = y0; yb < y1; yb += 4) {
        unsigned long long a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long* row = base + (size_t)min(yb + k, y1 - 1) * W64;
          a[k] = row[wi]; b[k] = row[wi1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int y = yb + k;
          if (y >= y1) break;
          int t0, t1;
          span_tiles(el, row_span<SWEEP>(el, y, H), x0, x1, t0, t1);
          if (t1 <= t0) continue;
          unsigned long long rowbits = (t1 - t0 >= 64 ? ~0ull : ((1ull << (t1 - t0)) - 1ull)) << (t0 - x0);   // w <= 64
          unsigned long long win = a[k] >> sh;                     // row_window(row, x0, w) on the prefetched words
          if (sh && sh + w > 64) win |= b[k] << (64 - sh);
          rowbits &= win & wmask;
          if (rowbits) put(y, rowbits);
        }
      }
    } else {
      for (int y = y0; y < y1; ++y) {
        int t0, t1;
        span_tiles(el, row_span<SWEEP>(el, y, H), x0, x1, t0, t1);
        if (t1 <= t0) continue;
        put(y, (t1 - t0 >= 64 ? ~0ull : ((1ull << (t1 - t0)) - 1ull)) << (t0 - x0));
      }
    }
    if (masks) {
      if (!mid) masks[moff] = m;                 // area <= 64: one word
      else {
        const int nwords = (area + 63) >> 6;
        for (int k = 0; k < nwords; ++k) masks[moff + (unsigned)k] = mw[k * 256];
      }
    }
  }
  unsigned long long big = __ballot(area > solo_limit && !mid);
  unsigned long long* words = s_words[threadIdx.x >> 6];
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const unsigned g = (unsigned)readlane_i((int)gi, src);
    const unsigned l = (unsigned)readlane_i((int)lo, src), h = (unsigned)readlane_i((int)hi, src);
    Ellipse eg;
    eg.gx = readlane_f(el.gx, src); eg.gy = readlane_f(el.gy, src); eg.a = readlane_f(el.a, src);
    eg.b = readlane_f(el.b, src); eg.c = readlane_f(el.c, src); eg.tau = readlane_f(el.tau, src);
    ellipse_derive(eg);
    if (SWEEP) {
      eg.pvx = readlane_f(el.pvx, src); eg.pvy = readlane_f(el.pvy, src);
      eg.st0 = sw.t_min; eg.st1 = sw.t_max; eg.srs = sw.rs_time;
    }
    const int x0 = l & 0xFFFF, y0 = l >> 16, x1 = h & 0xFFFF, y1 = h >> 16;
    const int w = x1 - x0, rows = y1 - y0, a = w * rows;
    const unsigned pidx = g / (unsigned)N;
    const unsigned mo = (unsigned)readlane_i((int)moff, src);
    const int nwords = (a + 63) >> 6;
    unsigned c = 0;
    if (nwords <= kMaskWords) {
      // one lane per tile ROW of the box: the row's hit bits (ellipse span AND open tiles, up to w bits) are OR-ed
      // into the box-local bit string (bit (y-y0)*w + (x-x0)) assembled in wave-private LDS, then written out
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < nwords; k += 64) words[k] = 0ull;
      __builtin_amdgcn_wave_barrier();
      for (int r = lane; r < rows; r += 64) {
        int t0, t1;
        span_tiles(eg, row_span<SWEEP>(eg, y0 + r, H), x0, x1, t0, t1);
        const unsigned long long* orow = open_bits ? open_bits + ((size_t)pidx * tiles_y + (y0 + r)) * W64 : nullptr;
        // the row's local bits in chunks of 64 columns
        for (int c0 = 0; c0 < w; c0 += 64) {
          const int cw = min(64, w - c0);
          const int lo_c = max(t0 - x0 - c0, 0), hi_c = min(t1 - x0 - c0, cw);
          if (hi_c <= lo_c) continue;
          unsigned long long bits = (hi_c - lo_c == 64 ? ~0ull : ((1ull << (hi_c - lo_c)) - 1ull)) << lo_c;
          if (orow) bits &= row_window(orow, x0 + c0, cw);
          if (!bits) continue;
          const int bit0 = r * w + c0;                       // position of the chunk in the box-local bit string
          const int wi = bit0 >> 6, sh = bit0 & 63;
          atomicOr(&words[wi], bits << sh);
          if (sh && (bits >> (64 - sh))) atomicOr(&words[wi + 1], bits >> (64 - sh));
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < nwords; k += 64) {
        const unsigned long long m = words[k];
        if (masks) masks[mo + (unsigned)k] = m;
        c += (unsigned)__popcll(m);
      }
      c = (unsigned)wave_sum_i((int)c);
    } else {
      // boxes beyond the LDS bit string (> 32768 tiles): one tile per lane and step
      const float rw = 1.0f / (float)w;
      const unsigned pbase = pidx * T;
      for (int base = 0; base < a; base += 64) {
        const int t = base + lane;
        bool ok = false;
        if (t < a) {
          const int q = (int)(((float)t + 0.5f) * rw);
          const int tx = x0 + (t - q * w), ty = y0 + q;
          ok = tile_hit<SWEEP>(eg, tx, ty, W, H) && (!done || done[pbase + (unsigned)(ty * tiles_x + tx)] == 0);
        }
        const unsigned long long m = __ballot(ok);
        if (masks && lane == 0) masks[mo + (unsigned)(base >> 6)] = m;
        c += (unsigned)__popcll(m);
      }
    }
    if (lane == src) cnt = c;
  }
  if (j < n_slice) { slice_gi[j] = gi; counts[j] = cnt; }
}

// emission with holes, wave-cooperative: a wave owns 64 slice Gaussians; every Gaussian that still has
// open tiles is expanded by the whole wave (64 tiles of its box per step, ballot-compacted), so one
// huge box does not serialise a lane while the other 63 idle.  Order inside a Gaussian = (y, x).
template <bool WAVE_PER_G>
__global__ __launch_bounds__(256) void emit_open_kernel(int n_slice, int N, int T, int tiles_x,
                                                        const unsigned* __restrict__ slice_gi,
                                                        const unsigned* __restrict__ counts,
                                                        const unsigned* __restrict__ cum,
                                                        const float* __restrict__ records,
                                                        const unsigned char* __restrict__ done,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ vals,
                                                        int W, int H, unsigned invalid_key, int compact,
                                                        const unsigned long long* __restrict__ masks,   // nullable
                                                        const unsigned* __restrict__ mask_off,
                                                        unsigned char* __restrict__ tile_hot) {         // nullable
  const int lane = lane_id();
  const int j = WAVE_PER_G ? (lane == 0 ? (int)(blockIdx.x * 4 + (threadIdx.x >> 6)) : n_slice)
                           : (int)(blockIdx.x * 256 + threadIdx.x);
  unsigned cnt = 0, gi = 0, e0 = 0, lo = 0, hi = 0, moff = 0;
  int hot = 0;             // opacity above the alpha clamp: the tiles it lands on need the clamping compositor loop
  Ellipse el = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 1.f, 1.f, 0.f, 0.f, 0.f};
  if (j < n_slice) {
    cnt = counts[j];
    if (cnt) {
      gi = slice_gi[j];
      e0 = cum[j];
      const float* rec = records + (size_t)gi * kRecFloats;
      lo = (unsigned)__float_as_int(rec[10]);
      hi = (unsigned)__float_as_int(rec[11]);
      hot = (tile_hot != nullptr && rec[5] > K::kAlphaMax) ? 1 : 0;
      if (masks) moff = mask_off[j];
      else if (invalid_key) el = make_ellipse(rec);
    }
  }
  if (masks && !WAVE_PER_G) {
    // boxes of one mask word (<= 64 tiles), when the wave holds several: every lane expands its own word — up to 64
    // Gaussians in ~cnt steps instead of one wave-wide step each (a scene of uniformly small splats has dozens per
    // wave).  Same (y, x) order inside a Gaussian; the writes of neighbouring lanes land in neighbouring ranges.
    // (Boxes of two to four words stay with the wave-wide expansion below: per-lane expansion of those was measured
    // slower, 0.33 -> 0.41 ms per frame on the fitted-model-like scene, run r3_run5.)
    const int x0 = lo & 0xFFFF, y0 = lo >> 16, x1 = hi & 0xFFFF, y1 = hi >> 16;
    const int w = x1 - x0, area = w * (y1 - y0);
    const bool small = cnt != 0 && area <= kCountSoloMax;
    if (__popcll(__ballot(small)) >= kCountSoloMany) {
      if (small) {
        unsigned long long m = masks[moff];
        const float rw = 1.0f / (float)w;
        const unsigned pbase = (gi / (unsigned)N) * (unsigned)T;
        unsigned e = e0;
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int q = (int)(((float)b + 0.5f) * rw);
          const unsigned k = pbase + (unsigned)((y0 + q) * tiles_x + x0 + (b - q * w));
          keys[e] = k;
          vals[e] = gi;
          if (hot) tile_hot[k] = 1;
          ++e;
        }
        cnt = 0;
      }
    }
  }
  unsigned long long todo = __ballot(cnt != 0);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const unsigned g = (unsigned)readlane_i((int)gi, src);
    unsigned e = (unsigned)readlane_i((int)e0, src);
    const unsigned l = (unsigned)readlane_i((int)lo, src), h = (unsigned)readlane_i((int)hi, src);
    Ellipse eg;
    eg.gx = readlane_f(el.gx, src); eg.gy = readlane_f(el.gy, src); eg.a = readlane_f(el.a, src);
    eg.b = readlane_f(el.b, src); eg.c = readlane_f(el.c, src); eg.tau = readlane_f(el.tau, src);
    ellipse_derive(eg);
    const int x0 = l & 0xFFFF, y0 = l >> 16, x1 = h & 0xFFFF, y1 = h >> 16;
    const int w = x1 - x0, area = w * (y1 - y0);
    const float rw = 1.0f / (float)w;
    const unsigned pbase = (g / (unsigned)N) * (unsigned)T;
    const bool hot_g = readlane_i(hot, src) != 0;          // rare
    if (masks) {
      // the exact-count pass left one bit per box tile (open AND inside the ellipse): no second ellipse test,
      // no tile_done reads — a word per 64 tiles drives the compaction directly
      const unsigned mo = (unsigned)readlane_i((int)moff, src);
      for (int base = 0; base < area; base += 64) {
        const unsigned long long m = masks[mo + (unsigned)(base >> 6)];
        if ((m >> lane) & 1ull) {
          const int t = base + lane;
          const int q = (int)(((float)t + 0.5f) * rw);
          const int tx = x0 + (t - q * w), ty = y0 + q;
          const unsigned dst = e + (unsigned)__popcll(m & lt_mask);
          const unsigned k = pbase + (unsigned)(ty * tiles_x + tx);
          keys[dst] = k;
          vals[dst] = g;
          if (hot_g) tile_hot[k] = 1;
        }
        e += (unsigned)__popcll(m);
      }
      continue;
    }
    for (int base = 0; base < area; base += 64) {
      const int t = base + lane;
      bool open = false;
      unsigned k = 0;
      if (t < area) {
        const int q = (int)(((float)t + 0.5f) * rw);
        const int tx = x0 + (t - q * w), ty = y0 + q;
        k = pbase + (unsigned)(ty * tiles_x + tx);
        open = !done || done[k] == 0;
        if (open && invalid_key && !tile_hit(eg, tx, ty, W, H)) {
          if (compact) open = false;      // counts are exact: culled pairs take no slot
          else k = invalid_key;           // counts are box counts: park the slot behind every tile
        }
      }
      const unsigned long long m = __ballot(open);
      if (open) {
        const unsigned dst = e + (unsigned)__popcll(m & lt_mask);
        keys[dst] = k;
        vals[dst] = g;
        if (hot_g && (invalid_key == 0u || k != invalid_key)) tile_hot[k] = 1;
      }
      e += (unsigned)__popcll(m);
    }
  }
}

// ---------------------------------------------------------------------------
// Nearest-first selection (round 6).  A frame whose tiles saturate early composites a few percent of its visible
// (sub-pose, Gaussian) pairs — the benchmark scene's one depth slice holds 55 k of 3.95 M — yet the depth pre-sort
// ordered all of them (0.21 ms + a 5 M-rank count scan).  Here a two-level radix SELECT over the 31 key bits of the
// visible pairs (bits 30..20, then 19..9: a float's exponent and 14 mantissa bits) finds, per sub-pose, the key bound
// thr such that the pairs with key < thr hold at least `budget` bounding-box pairs (weights = num_tiles_hit) and no
// whole 512-ulp bucket could be left out; the compacting sort then keeps and orders ONLY those (SegDev key_hi), and the
// pairs behind the bound are sorted later, if the frame's first slice leaves a tile open (SegDev key_lo).  Where the
// bound falls changes nothing in the images: a slice boundary never does.
// Three launches, ordered by the stream alone: level-0 histogram (LDS, 64-bit sums, flushed with device-scope atomics),
// level-1 histogram (every block first finds level 0's crossing for itself from the 2048 totals), and one block per
// segment that walks both histograms and writes the bound.  (First built with a ticket counter and __threadfence() so
// that a segment's last block did the walk: on this 8-XCD part an agent-scope fence is an L2 write-back, and 2560 waves
// issuing one made each level a 50 us kernel; measured, profiles/r06_depth_select_fence.txt.)
// ---------------------------------------------------------------------------
constexpr int kSelBits = 11, kSelBins = 1 << kSelBits;
struct SelSeg {                                // per segment, zeroed by the caller
  unsigned long long hist[2][kSelBins];
};

__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long& total,
                                                               unsigned long long* lds /*[8]*/) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  const unsigned long long w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
  const unsigned long long woff = wave == 0 ? 0ull : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
  total = w0 + w1 + w2 + w3;
  __syncthreads();
  return woff + inc - v;
}

// where the cumulative weight of a 2048-bin histogram reaches `target` (all 256 threads of the block): the bin, or
// kSelBins when the whole histogram carries less; *before = weight in front of that bin, *total = weight of all bins
__device__ __forceinline__ unsigned sel_crossing(const unsigned long long* __restrict__ hist, unsigned long long target,
                                                 unsigned long long* s_scan /*[8]*/, unsigned long long* s_res /*[2]*/,
                                                 unsigned long long& before, unsigned long long& total) {
  constexpr int PER = kSelBins / 256;
  if (threadIdx.x == 0) { s_res[0] = (unsigned long long)kSelBins; s_res[1] = 0ull; }
  unsigned long long c[PER], sum = 0ull;
#pragma unroll
  for (int j = 0; j < PER; ++j) { c[j] = hist[threadIdx.x * PER + j]; sum += c[j]; }
  unsigned long long run = block_excl_scan64(sum, total, s_scan);      // (its barriers also publish s_res)
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (run < target && run + c[j] >= target) { s_res[0] = (unsigned long long)(threadIdx.x * PER + j); s_res[1] = run; }
    run += c[j];                                                       // (at most one thread: the crossing is unique)
  }
  __syncthreads();
  before = s_res[1];
  const unsigned found = (unsigned)s_res[0];
  __syncthreads();
  return found;
}

constexpr int kSelRounds = 8;                  // keys per thread and chunk: 2048-key chunks
template <int LEVEL>
__global__ __launch_bounds__(256) void depth_hist_kernel(size_t seg_len, const unsigned* __restrict__ keys,
                                                         const unsigned* __restrict__ weights,
                                                         unsigned long long budget, SelSeg* __restrict__ ws) {
  __shared__ unsigned long long h[kSelBins];
  __shared__ unsigned long long s_scan[8], s_res[2];
  const unsigned seg = blockIdx.y;
  SelSeg& S = ws[seg];
  unsigned b1 = 0;
  if (LEVEL == 1) {
    unsigned long long before, total;
    b1 = sel_crossing(S.hist[0], budget, s_scan, s_res, before, total);
    if (b1 >= (unsigned)kSelBins) return;      // the segment carries less than the budget: everything is selected
  }
  for (int d = threadIdx.x; d < kSelBins; d += 256) h[d] = 0ull;
  __syncthreads();
  const size_t seg_base = (size_t)seg * seg_len, limit = seg_base + seg_len;
  for (size_t base = seg_base + (size_t)blockIdx.x * (256 * kSelRounds); base < limit;
       base += (size_t)gridDim.x * (256 * kSelRounds)) {
    unsigned k[kSelRounds], w[kSelRounds];
#pragma unroll
    for (int r = 0; r < kSelRounds; ++r) {
      const size_t i = base + (size_t)r * 256 + threadIdx.x;
      k[r] = keys[min(i, limit - 1)];
      if (i >= limit) k[r] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int r = 0; r < kSelRounds; ++r) {
      const size_t i = base + (size_t)r * 256 + threadIdx.x;
      // (visible keys are positive floats: bit 31 marks a culled pair)
      const bool ok = (k[r] >> 31) == 0u && (LEVEL == 0 || (k[r] >> 20) == b1);
      w[r] = ok ? weights[min(i, limit - 1)] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kSelRounds; ++r)
      if (w[r]) atomicAdd(&h[LEVEL == 0 ? (k[r] >> 20) : ((k[r] >> 9) & (unsigned)(kSelBins - 1))], (unsigned long long)w[r]);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kSelBins; d += 256)
    if (h[d]) atomicAdd(&S.hist[LEVEL][d], h[d]);
}

// one block per segment: the bound from the two histograms
__global__ __launch_bounds__(256) void depth_find_kernel(unsigned long long budget, const SelSeg* __restrict__ ws,
                                                         unsigned* __restrict__ thr_out,
                                                         unsigned* __restrict__ grand_out) {
  __shared__ unsigned long long s_scan[8], s_res[2];
  const unsigned seg = blockIdx.x;
  const SelSeg& S = ws[seg];
  unsigned long long before0, total0, before1, total1;
  const unsigned b1 = sel_crossing(S.hist[0], budget, s_scan, s_res, before0, total0);
  unsigned thr = 0x80000000u;                  // the segment holds less than the budget: all of it
  if (b1 < (unsigned)kSelBins) {
    // level 0 guarantees the crossing lies inside bucket b1; (b1, b2) + 1 is the exclusive bound in 512-ulp buckets
    unsigned b2 = sel_crossing(S.hist[1], budget - before0, s_scan, s_res, before1, total1);
    if (b2 >= (unsigned)kSelBins) b2 = (unsigned)(kSelBins - 1);
    thr = (((b1 << kSelBits) | b2) + 1u) << 9;
  }
  if (threadIdx.x == 0) {
    thr_out[seg] = thr;
    if (grand_out) atomicAdd(grand_out, (unsigned)total0);            // frame total of bounding-box pairs (mod 2^32)
  }
}

// upstream-compatible 64-bit intersection ids (one thread per Gaussian; API-parity path)
__global__ __launch_bounds__(256) void map_isect_kernel(int N, const float* __restrict__ xys,
                                                        const float* __restrict__ depths,
                                                        const int* __restrict__ radii,
                                                        const int* __restrict__ cum_tiles_hit,  // inclusive
                                                        int tiles_x, int tiles_y, long long* __restrict__ isect_ids,
                                                        int* __restrict__ gaussian_ids) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int rad = radii[i];
  if (rad <= 0) return;
  const float inv_tile = 1.0f / (float)K::kTile;
  float radf = (float)rad;
  float tcx = xys[2 * i] * inv_tile, tcy = xys[2 * i + 1] * inv_tile, tr = radf * inv_tile;
  int x0 = (int)(tcx - tr), x1 = (int)(tcx + tr + 1.0f);
  int y0 = (int)(tcy - tr), y1 = (int)(tcy + tr + 1.0f);
  x0 = x0 < 0 ? 0 : (x0 > tiles_x ? tiles_x : x0);
  x1 = x1 < 0 ? 0 : (x1 > tiles_x ? tiles_x : x1);
  y0 = y0 < 0 ? 0 : (y0 > tiles_y ? tiles_y : y0);
  y1 = y1 < 0 ? 0 : (y1 > tiles_y ? tiles_y : y1);
  long long depth_bits = (long long)(unsigned)__float_as_int(depths[i]);
  int cur = i == 0 ? 0 : cum_tiles_hit[i - 1];
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      long long tile = (long long)(y * tiles_x + x);
      isect_ids[cur] = (tile << 32) | depth_bits;
      gaussian_ids[cur] = i;
      ++cur;
    }
}

__global__ __launch_bounds__(256) void depth_keys64_kernel(size_t n, int N, const unsigned* __restrict__ dk,
                                                           unsigned long long* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = ((unsigned long long)(i / (size_t)N) << 32) | (unsigned long long)dk[i];
}

}  // namespace gs

using namespace gs;

// C ABI -------------------------------------------------------------------------
GS_EXPORT long long gs_scan_workspace_bytes(long long n) { return (long long)scan_ws_bytes((size_t)n) + 256; }
GS_EXPORT long long gs_radix_sort_workspace_bytes(long long n, int begin_bit, int end_bit) {
  if (n <= 0 || end_bit <= begin_bit) return 0;
  // sized for the larger (u64) layout so one query serves both key widths
  size_t a = radix_ws_bytes<unsigned>((size_t)n, 0, end_bit - begin_bit);
  size_t b = radix_ws_bytes<unsigned long long>((size_t)n, 0, end_bit - begin_bit);
  return (long long)(a > b ? a : b);
}

// out[i] = sum_{j<i} in[j]; *total_out (device u32, nullable) = sum of all.  in == out allowed.
GS_EXPORT int gs_exclusive_scan_u32(long long n, const unsigned* in, unsigned* out, unsigned* total_out, void* ws,
                                    long long ws_bytes, void* stream) {
  if (n <= 0) return GS_ERR_INVALID;
  if ((size_t)ws_bytes < scan_ws_bytes((size_t)n)) return GS_ERR_WORKSPACE;
  return run_scan((size_t)n, in, out, total_out, ws, (hipStream_t)stream);
}

// Stable LSD radix sort of (key, u32 value) pairs over key bits [begin_bit, end_bit).
// Buffers 0 are the input (clobbered), buffers 1 scratch of the same size; *result_buf (host int)
// receives which pair holds the sorted output.  vals0_is_iota!=0: vals0 is not read, values = index.
GS_EXPORT int gs_radix_sort_pairs_u32(long long n, unsigned* keys0, unsigned* vals0, unsigned* keys1,
                                      unsigned* vals1, int vals0_is_iota, int begin_bit, int end_bit, void* ws,
                                      long long ws_bytes, int* result_buf, void* stream) {
  if (n <= 0 || begin_bit < 0 || end_bit > 32) return GS_ERR_INVALID;
  return radix_sort<unsigned>((size_t)n, 0, keys0, vals0, keys1, vals1, vals0_is_iota, begin_bit, end_bit, ws,
                              (size_t)ws_bytes, result_buf, (hipStream_t)stream);
}

// gs_radix_sort_pairs_u32 whose final pass also writes gather_out[i] = gather_src[sorted value i] (gather_out: n
// ints; the tile sort of a depth slice sorts emission indices and leaves the record index of every sorted entry)
GS_EXPORT int gs_radix_sort_pairs_gather_u32(long long n, unsigned* keys0, unsigned* vals0, unsigned* keys1,
                                             unsigned* vals1, int vals0_is_iota, int begin_bit, int end_bit,
                                             void* ws, long long ws_bytes, int* result_buf,
                                             const unsigned* gather_src, unsigned* gather_out,
                                             const unsigned* n_dev, void* stream) {
  if (n <= 0 || begin_bit < 0 || end_bit > 32 || !gather_src || !gather_out) return GS_ERR_INVALID;
  return radix_sort<unsigned>((size_t)n, 0, keys0, vals0, keys1, vals1, vals0_is_iota, begin_bit, end_bit, ws,
                              (size_t)ws_bytes, result_buf, (hipStream_t)stream, 11, gather_src, gather_out, n_dev);
}

// gs_radix_sort_pairs_u32 with a SECOND payload per key: payload2_in[i] belongs to input element i (not clobbered);
// the passes ping-pong it through payload2_a / payload2_b (n ints each), *result_p2 (host) = 0 / 1 names the one
// that holds it in sorted order.  The tile sort of a depth slice carries the record index of every entry this way
// (8 sequential bytes per entry and pass instead of a random 4-byte gather per entry after the sort).
GS_EXPORT int gs_radix_sort_pairs_carry_u32(long long n, unsigned* keys0, unsigned* vals0, unsigned* keys1,
                                            unsigned* vals1, int vals0_is_iota, int begin_bit, int end_bit, void* ws,
                                            long long ws_bytes, int* result_buf, const unsigned* payload2_in,
                                            unsigned* payload2_a, unsigned* payload2_b, int* result_p2,
                                            const unsigned* n_dev, void* stream) {
  if (n <= 0 || begin_bit < 0 || end_bit > 32 || !payload2_in || !payload2_a || !payload2_b || !result_p2)
    return GS_ERR_INVALID;
  return radix_sort<unsigned>((size_t)n, 0, keys0, vals0, keys1, vals1, vals0_is_iota, begin_bit, end_bit, ws,
                              (size_t)ws_bytes, result_buf, (hipStream_t)stream, 11, nullptr, nullptr, n_dev, nullptr,
                              0ull, payload2_in, payload2_a, payload2_b, result_p2);
}

GS_EXPORT int gs_radix_sort_pairs_u64(long long n, unsigned long long* keys0, unsigned* vals0,
                                      unsigned long long* keys1, unsigned* vals1, int vals0_is_iota, int begin_bit,
                                      int end_bit, void* ws, long long ws_bytes, int* result_buf, void* stream) {
  if (n <= 0 || begin_bit < 0 || end_bit > 64) return GS_ERR_INVALID;
  return radix_sort<unsigned long long>((size_t)n, 0, keys0, vals0, keys1, vals1, vals0_is_iota, begin_bit, end_bit,
                                        ws, (size_t)ws_bytes, result_buf, (hipStream_t)stream);
}

// Segmented variant: n = num_segments * seg_len keys, every segment of seg_len keys sorted independently
// (stable) in one set of launches.  Used for the per-sub-pose depth pre-sort: 32-bit depth keys, 3 passes,
// instead of 35-bit (sub-pose, depth) keys in u64.
GS_EXPORT long long gs_segmented_sort_workspace_bytes(long long n, long long seg_len, int begin_bit, int end_bit) {
  if (n <= 0 || seg_len <= 0 || end_bit <= begin_bit) return 0;
  return (long long)radix_ws_bytes<unsigned>((size_t)n, (size_t)seg_len, end_bit - begin_bit, 8);
}

GS_EXPORT int gs_segmented_sort_pairs_u32(long long n, long long seg_len, unsigned* keys0, unsigned* vals0,
                                          unsigned* keys1, unsigned* vals1, int vals0_is_iota, int begin_bit,
                                          int end_bit, void* ws, long long ws_bytes, int* result_buf, void* stream) {
  if (n <= 0 || seg_len <= 0 || n % seg_len != 0 || begin_bit < 0 || end_bit > 32) return GS_ERR_INVALID;
  // 8-bit digits: at a few million keys the light 4-wave-per-SIMD 8-bit passes beat three 11-bit passes
  // (250 VGPRs, 80 KB LDS per block) — measured 0.51 ms vs 0.37 ms for the 64-bit route at 5M keys
  return radix_sort<unsigned>((size_t)n, (size_t)seg_len, keys0, vals0, keys1, vals1, vals0_is_iota, begin_bit,
                              end_bit, ws, (size_t)ws_bytes, result_buf, (hipStream_t)stream, 8);
}

// Compacting form for the depth pre-sort: keys equal to skip_key (culled Gaussians) are dropped by the first pass.
// On return segment s holds its seg_counts[s] surviving keys sorted at [s*seg_len, s*seg_len + seg_counts[s]); what
// lies behind them is unspecified.  Payload = global index (iota).  gather_src / gather_out (nullable together):
// gather_out[slot] = gather_src[payload] for every sorted survivor, written by the last pass.
GS_EXPORT long long gs_segmented_sort_compact_workspace_bytes(long long n, long long seg_len, int begin_bit,
                                                              int end_bit, int max_digit_bits) {
  if (n <= 0 || seg_len <= 0 || end_bit <= begin_bit) return 0;
  return (long long)radix_ws_bytes<unsigned>((size_t)n, (size_t)seg_len, end_bit - begin_bit, max_digit_bits);
}

GS_EXPORT int gs_segmented_sort_compact_u32(long long n, long long seg_len, unsigned* keys0, unsigned* vals0,
                                            unsigned* keys1, unsigned* vals1, int begin_bit, int end_bit,
                                            int max_digit_bits, unsigned skip_key, unsigned* seg_counts,
                                            const unsigned* gather_src, unsigned* gather_out, void* ws,
                                            long long ws_bytes, int* result_buf, void* stream) {
  if (n <= 0 || seg_len <= 0 || n % seg_len != 0 || begin_bit < 0 || end_bit > 32 || !seg_counts)
    return GS_ERR_INVALID;
  if ((gather_src != nullptr) != (gather_out != nullptr)) return GS_ERR_INVALID;
  return radix_sort<unsigned>((size_t)n, (size_t)seg_len, keys0, vals0, keys1, vals1, 1, begin_bit, end_bit, ws,
                              (size_t)ws_bytes, result_buf, (hipStream_t)stream, max_digit_bits, gather_src,
                              gather_out, nullptr, seg_counts, (unsigned long long)skip_key);
}

// Exclusive scan over n = k*seg_len values of which only the first seg_counts[s] of every segment are live: the rest
// count as zero and are not read.  out is defined for the live ranks and at every segment's first rank; what lies
// behind a segment's live ranks is unspecified (round 6: blocks without a live element write nothing else).
GS_EXPORT int gs_exclusive_scan_segments_u32(long long n, long long seg_len, const unsigned* seg_counts,
                                             const unsigned* in, unsigned* out, unsigned* total_out, void* ws,
                                             long long ws_bytes, void* stream) {
  if (n <= 0 || seg_len <= 0 || n % seg_len != 0 || !seg_counts) return GS_ERR_INVALID;
  if ((size_t)ws_bytes < scan_ws_bytes((size_t)n)) return GS_ERR_WORKSPACE;
  return run_scan((size_t)n, in, out, total_out, ws, (hipStream_t)stream, DevLen{nullptr, 1u, 1u},
                  SegMask{seg_counts, (size_t)seg_len});
}

// (sub-pose, depth) keys for the N-sized pre-sort: out[i] = (i / N) << 32 | depth_keys[i]
GS_EXPORT int gs_make_depth_keys64(long long n, int N, const unsigned* depth_keys, unsigned long long* out,
                                   void* stream) {
  if (n <= 0 || N <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(depth_keys64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (size_t)n, N, depth_keys, out);
  return gs_launch_status();
}

// counts_out[r] = num_tiles_hit[sorted_gi[r]]
GS_EXPORT int gs_gather_counts(long long n, const unsigned* sorted_gi, const int* num_tiles_hit, unsigned* counts_out,
                               void* stream) {
  if (n <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (size_t)n, sorted_gi, num_tiles_hit, counts_out);
  return gs_launch_status();
}

// Emit the tile intersections of the depth-ranked Gaussians: keys[e] = p*T + tile, vals[e] = p*N + g.
GS_EXPORT int gs_emit_intersects(long long n_ranked, int N, int H, int W, const unsigned* sorted_gi,
                                 const unsigned* cum_excl, const float* records, long long n_isect, unsigned* keys,
                                 unsigned* vals, unsigned invalid_key, void* stream) {
  if (n_ranked <= 0 || N <= 0) return GS_ERR_INVALID;
  if (n_isect <= 0) return GS_OK;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((n_isect + kEmitChunk - 1) / kEmitChunk)), dim3(256), 0,
                     (hipStream_t)stream, (size_t)n_ranked, N, tiles_x * tiles_y, tiles_x, sorted_gi, cum_excl, records,
                     (size_t)n_isect, keys, vals, W, H, invalid_key);
  return gs_launch_status();
}

// bins[t] = [start,end) of key t in the sorted u32 keys ((0, 0) for a key that does not occur; every bin is written).
GS_EXPORT int gs_tile_bin_edges_u32(long long n, const unsigned* sorted_keys, int num_bins, int* bins,
                                    const unsigned* n_dev, void* stream) {
  if (num_bins <= 0) return GS_ERR_INVALID;
  if (n <= 0) {       // no key at all: every bin is empty (the kernel below zeroes the key-less bins itself)
    hipError_t e = hipMemsetAsync(bins, 0, (size_t)num_bins * 2 * sizeof(int), (hipStream_t)stream);
    return e == hipSuccess ? GS_OK : 1000 + (int)e;
  }
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (n_dev && blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL((bin_edges_kernel<unsigned, 0>), dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (size_t)n, sorted_keys, reinterpret_cast<int2*>(bins), (unsigned)num_bins, n_dev);
  return gs_launch_status();
}

// gs_tile_bin_edges_u32 plus ids_out[i] = gi_of_e[sorted_vals[i]] for i < n and ids_out[n .. n+8) = 0
// (ids_out holds n + 8 ints): the record index of every sorted entry, what gs_rasterize_*_slice take as sorted_ids.
GS_EXPORT int gs_tile_bin_edges_ids_u32(long long n, const unsigned* sorted_keys, int num_bins, int* bins,
                                        const unsigned* sorted_vals, const unsigned* gi_of_e, unsigned* ids_out,
                                        void* stream) {
  if (num_bins <= 0 || !sorted_vals || !gi_of_e || !ids_out) return GS_ERR_INVALID;
  hipError_t e = hipMemsetAsync(bins, 0, (size_t)num_bins * 2 * sizeof(int), (hipStream_t)stream);
  if (e != hipSuccess) return 1000 + (int)e;
  if (n < 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(bin_edges_ids_kernel, dim3((unsigned)((n + 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (size_t)n, sorted_keys, reinterpret_cast<int2*>(bins), sorted_vals, gi_of_e, ids_out);
  return gs_launch_status();
}

// gsplat.get_tile_bin_edges: tile id = isect_id >> 32
GS_EXPORT int gs_tile_bin_edges_u64(long long n, const unsigned long long* sorted_ids, int num_bins, int* bins,
                                    void* stream) {
  if (num_bins <= 0) return GS_ERR_INVALID;
  if (n <= 0) {       // no key at all: every bin is empty (the kernel below zeroes the key-less bins itself)
    hipError_t e = hipMemsetAsync(bins, 0, (size_t)num_bins * 2 * sizeof(int), (hipStream_t)stream);
    return e == hipSuccess ? GS_OK : 1000 + (int)e;
  }
  hipLaunchKernelGGL((bin_edges_kernel<unsigned long long, 32>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (size_t)n, sorted_ids, reinterpret_cast<int2*>(bins), (unsigned)num_bins);
  return gs_launch_status();
}

// gsplat.map_gaussian_to_intersects (device side): cum_tiles_hit is the INCLUSIVE cumsum like upstream.
GS_EXPORT int gs_map_gaussian_to_intersects(int N, const float* xys, const float* depths, const int* radii,
                                            const int* cum_tiles_hit, int H, int W, long long* isect_ids,
                                            int* gaussian_ids, void* stream) {
  if (N <= 0) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  hipLaunchKernelGGL(map_isect_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, xys, depths,
                     radii, cum_tiles_hit, tiles_x, tiles_y, isect_ids, gaussian_ids);
  return gs_launch_status();
}

// ---- depth-sliced binning -------------------------------------------------------------------
// bounds [P*K]: first depth rank of each sub-pose at which the cumulative intersection count reaches base<<k
GS_EXPORT int gs_slice_plan(int P, int N, int K, const unsigned* cum_excl, const unsigned* total, long long base,
                            int* bounds, unsigned* rels, unsigned* seg_totals, const unsigned* n_live,
                            unsigned* tail, void* stream) {
  if (P <= 0 || N <= 0 || K <= 0 || K > 32 || base <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(slice_plan_kernel, dim3((P * K + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, N, K, cum_excl,
                     total, (unsigned long long)base, bounds, rels, seg_totals, n_live, tail, (const unsigned*)nullptr);
  return gs_launch_status();
}

// gs_slice_plan for a nearest-first selection (gs_depth_select + gs_segmented_sort_select_u32): the ranked pairs are the
// selection only, so every boundary lies behind them (one slice = all of them) and the frame's total of bounding-box
// pairs comes from the selection's own count (`grand`, device, = gs_depth_select's grand_out) instead of the scan's.
GS_EXPORT int gs_slice_plan_select(int P, int N, int K, const unsigned* cum_excl, const unsigned* total, int* bounds,
                                   unsigned* rels, unsigned* seg_totals, const unsigned* n_live, unsigned* tail,
                                   const unsigned* grand, void* stream) {
  if (P <= 0 || N <= 0 || K <= 0 || K > 32 || !n_live || !grand) return GS_ERR_INVALID;
  hipLaunchKernelGGL(slice_plan_kernel, dim3((P * K + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, N, K, cum_excl,
                     total, 1ull << 40, bounds, rels, seg_totals, n_live, tail, grand);
  return gs_launch_status();
}

// ---- nearest-first selection ---------------------------------------------------------------------------
GS_EXPORT long long gs_depth_select_workspace_bytes(int segments) {
  return segments > 0 ? (long long)segments * (long long)sizeof(SelSeg) + 256 : 0;
}

// Per segment s of seg_len keys (n = segments * seg_len; keys with bit 31 set are culled pairs): thr_out[s] = the
// smallest multiple of 512 such that the pairs with key < thr_out[s] carry at least `budget` of the weights (0x80000000:
// the whole segment carries less, everything is selected).  *grand_out (device u32, nullable, ZEROED by the caller) +=
// the weight of every visible pair.  ws: gs_depth_select_workspace_bytes(segments) bytes, ZEROED by the caller.
GS_EXPORT int gs_depth_select(long long n, long long seg_len, const unsigned* keys, const unsigned* weights,
                              long long budget, unsigned* thr_out, unsigned* grand_out, void* ws, long long ws_bytes,
                              void* stream) {
  if (n <= 0 || seg_len <= 0 || n % seg_len != 0 || budget <= 0 || !keys || !weights || !thr_out || !ws)
    return GS_ERR_INVALID;
  const long long segs = n / seg_len;
  if (segs > 65535) return GS_ERR_INVALID;
  if (ws_bytes < gs_depth_select_workspace_bytes((int)segs)) return GS_ERR_WORKSPACE;
  const long long chunks = (seg_len + 256 * kSelRounds - 1) / (256 * kSelRounds);
  // about four blocks per CU over all segments, at least two chunks per block
  const unsigned gx = (unsigned)std::max(1ll, std::min((chunks + 1) / 2, std::max(1ll, 1024 / segs)));
  SelSeg* S = reinterpret_cast<SelSeg*>(ws);
  hipLaunchKernelGGL(depth_hist_kernel<0>, dim3(gx, (unsigned)segs), dim3(256), 0, (hipStream_t)stream, (size_t)seg_len,
                     keys, weights, (unsigned long long)budget, S);
  hipLaunchKernelGGL(depth_hist_kernel<1>, dim3(gx, (unsigned)segs), dim3(256), 0, (hipStream_t)stream, (size_t)seg_len,
                     keys, weights, (unsigned long long)budget, S);
  hipLaunchKernelGGL(depth_find_kernel, dim3((unsigned)segs), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)budget, S, thr_out, grand_out);
  return gs_launch_status();
}

// gs_segmented_sort_compact_u32 restricted to a key range per segment: only keys in [key_lo[s], key_hi[s]) (device
// arrays, either nullable; the skip key is dropped as well) are kept and sorted.  keys_src is read by the first pass and
// left INTACT; keys0 / keys1 are scratch (n keys each).  Everything else as gs_segmented_sort_compact_u32.
// tail_cap > 0: the caller PROMISES that no segment keeps more than tail_cap keys — the passes behind the compacting one
// are then sized for that many (launch-latency bound otherwise); a segment that keeps more comes out WRONG beyond
// tail_cap: the caller must check seg_counts against it (gs_frame_forward does, and sorts again without the promise).
GS_EXPORT int gs_segmented_sort_select_u32(long long n, long long seg_len, const unsigned* keys_src, unsigned* keys0,
                                           unsigned* vals0, unsigned* keys1, unsigned* vals1, int begin_bit, int end_bit,
                                           int max_digit_bits, unsigned skip_key, const unsigned* key_lo,
                                           const unsigned* key_hi, unsigned* seg_counts, const unsigned* gather_src,
                                           unsigned* gather_out, void* ws, long long ws_bytes, int* result_buf,
                                           long long tail_cap, void* stream) {
  if (n <= 0 || seg_len <= 0 || n % seg_len != 0 || begin_bit < 0 || end_bit > 32 || !seg_counts || !keys_src ||
      tail_cap < 0)
    return GS_ERR_INVALID;
  if ((gather_src != nullptr) != (gather_out != nullptr)) return GS_ERR_INVALID;
  return radix_sort<unsigned>((size_t)n, (size_t)seg_len, keys0, vals0, keys1, vals1, 1, begin_bit, end_bit, ws,
                              (size_t)ws_bytes, result_buf, (hipStream_t)stream, max_digit_bits, gather_src,
                              gather_out, nullptr, seg_counts, (unsigned long long)skip_key, nullptr, nullptr, nullptr,
                              nullptr, keys_src, key_lo, key_hi,
                              tail_cap > 0 ? (unsigned)((tail_cap + sort_block_keys<unsigned>() - 1) / sort_block_keys<unsigned>()) : 0u);
}

// sat [P*(tiles_y+1)*(tiles_x+1)]: summed-area table of tiles that are NOT done.
GS_EXPORT int gs_tile_open_sat(int P, int H, int W, const unsigned char* tile_done, int* sat,
                               unsigned long long* open_bits, const int* gate, void* stream) {
  if (P <= 0) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  size_t lds = (size_t)(tiles_x + 1) * (tiles_y + 1) * sizeof(int);
  if (lds > 160 * 1024) return GS_ERR_INVALID;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_sat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(tile_sat_kernel, dim3(P), dim3(256), lds, (hipStream_t)stream, tiles_x, tiles_y, tile_done, sat,
                     open_bits, gate);
  return gs_launch_status();
}

// For the n_slice = slice_prefix[P] Gaussians of a depth slice (sub-pose p contributes the depth ranks
// slice_begin[p] .. of sorted_gi): slice_gi[j] = their global index, counts[j] = tiles still open
// (sat == NULL: all tiles open).
GS_EXPORT int gs_slice_counts(int n_slice, int P, int N, const int* slice_begin, const int* slice_prefix,
                              const unsigned* sorted_gi, const float* records, const int* sat, int H, int W,
                              unsigned* slice_gi, unsigned* counts, void* stream) {
  SliceDesc sd;
  if (n_slice <= 0 || !make_slice_desc(P, slice_begin, slice_prefix, sd)) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  hipLaunchKernelGGL(slice_counts_kernel, dim3((n_slice + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_slice, sd,
                     N, tiles_x, tiles_y, sorted_gi, records, sat, slice_gi, counts);
  return gs_launch_status();
}

// Exact counts: open tiles that also pass the ellipse test (sat / tile_done NULL: every tile open).
GS_EXPORT int gs_slice_counts_exact(int n_slice, int P, int N, const int* slice_begin, const int* slice_prefix,
                                    const unsigned* sorted_gi, const float* records, const int* sat,
                                    const unsigned char* tile_done, int H, int W, unsigned* slice_gi,
                                    unsigned* counts, int wave_per_gaussian, const unsigned* cum_rank,
                                    unsigned long long* hit_masks, unsigned* mask_off,
                                    const unsigned long long* open_bits, const int* gate, void* stream) {
  SliceDesc sd;
  if (n_slice <= 0 || !make_slice_desc(P, slice_begin, slice_prefix, sd)) return GS_ERR_INVALID;
  if (hit_masks && (!cum_rank || !mask_off)) return GS_ERR_INVALID;
  if ((tile_done != nullptr) != (open_bits != nullptr)) return GS_ERR_INVALID;   // both describe the same closed tiles
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  const SweepP none{nullptr, 0.f, 0.f, 0.f};
  if (wave_per_gaussian)
    hipLaunchKernelGGL(slice_counts_exact_kernel<true>, dim3((n_slice + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       n_slice, sd, N, tiles_x, tiles_y, sorted_gi, records, sat, tile_done, W, H, slice_gi, counts,
                       cum_rank, hit_masks, mask_off, open_bits, gate, none);
  else
    hipLaunchKernelGGL(slice_counts_exact_kernel<false>, dim3((n_slice + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, n_slice, sd, N, tiles_x, tiles_y, sorted_gi, records, sat, tile_done, W,
                       H, slice_gi, counts, cum_rank, hit_masks, mask_off, open_bits, gate, none);
  return gs_launch_status();
}

// gs_slice_counts_exact for the lists of the pixel-velocity compositors (gs_rasterize_*_rs_slice): a splat's centre is
// records[gi].xy + t * pix_vel[gi % N], t = a sample time in [t_min, t_max] plus the row time
// ((y + 0.5) / H - 0.5) * rolling_shutter_time of the pixel that looks at it; a tile is counted (and its hit-mask bit
// set: hit_masks is required) when the alpha >= 1/255 ellipse reaches it at ANY such time.  The records' tile boxes
// must already be swept (gs_project_pixvel_fwd).  Per-sample lists: t_min = t_max = 0.
GS_EXPORT int gs_slice_counts_exact_swept(int n_slice, int P, int N, const int* slice_begin, const int* slice_prefix,
                                          const unsigned* sorted_gi, const float* records, const int* sat,
                                          const unsigned char* tile_done, int H, int W, unsigned* slice_gi,
                                          unsigned* counts, int wave_per_gaussian, const unsigned* cum_rank,
                                          unsigned long long* hit_masks, unsigned* mask_off,
                                          const unsigned long long* open_bits, const float* pix_vel, float t_min,
                                          float t_max, float rolling_shutter_time, void* stream) {
  SliceDesc sd;
  if (n_slice <= 0 || N <= 0 || !pix_vel || !hit_masks || !cum_rank || !mask_off ||
      !make_slice_desc(P, slice_begin, slice_prefix, sd))
    return GS_ERR_INVALID;
  if ((tile_done != nullptr) != (open_bits != nullptr)) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  const SweepP sw{pix_vel, fminf(t_min, t_max), fmaxf(t_min, t_max), rolling_shutter_time};
  if (wave_per_gaussian)
    hipLaunchKernelGGL((slice_counts_exact_kernel<true, true>), dim3((n_slice + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       n_slice, sd, N, tiles_x, tiles_y, sorted_gi, records, sat, tile_done, W, H, slice_gi, counts,
                       cum_rank, hit_masks, mask_off, open_bits, (const int*)nullptr, sw);
  else
    hipLaunchKernelGGL((slice_counts_exact_kernel<false, true>), dim3((n_slice + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, n_slice, sd, N, tiles_x, tiles_y, sorted_gi, records, sat, tile_done, W,
                       H, slice_gi, counts, cum_rank, hit_masks, mask_off, open_bits, (const int*)nullptr, sw);
  return gs_launch_status();
}

// Emit the intersections of a slice with the tiles that are still open (depth order preserved).
GS_EXPORT int gs_emit_open_intersects(int n_slice, int N, int H, int W, const unsigned* slice_gi,
                                      const unsigned* counts, const unsigned* cum_excl, const float* records,
                                      const unsigned char* tile_done, unsigned* keys, unsigned* vals,
                                      unsigned invalid_key, int compact, int wave_per_gaussian,
                                      const unsigned long long* hit_masks, const unsigned* mask_off,
                                      unsigned char* tile_hot, void* stream) {
  if (n_slice <= 0) return GS_ERR_INVALID;
  if (hit_masks && (!mask_off || !compact)) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  if (wave_per_gaussian)
    hipLaunchKernelGGL(emit_open_kernel<true>, dim3((n_slice + 3) / 4), dim3(256), 0, (hipStream_t)stream, n_slice, N,
                       tiles_x * tiles_y, tiles_x, slice_gi, counts, cum_excl, records, tile_done, keys, vals, W, H,
                       invalid_key, compact, hit_masks, mask_off, tile_hot);
  else
    hipLaunchKernelGGL(emit_open_kernel<false>, dim3((n_slice + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_slice,
                       N, tiles_x * tiles_y, tiles_x, slice_gi, counts, cum_excl, records, tile_done, keys, vals, W, H,
                       invalid_key, compact, hit_masks, mask_off, tile_hot);
  return gs_launch_status();
}

GS_EXPORT const char* gs_version(void) { return "gsdeblur-hip 0.1 (gfx950, wave64)"; }
