// raster.hip — per-pixel front-to-back alpha compositing (forward) and the sub-frame averaging, written
// for gfx950 / wave64.  The reverse-order backward lives in raster_bwd.hip.
//
// Restates (absent fork sources) gsplat's rasterize_forward as recollected in SURVEY.md
// App. A "Blend"; constants in gs::K (gs_math.h).
//
// MI355X design (not the CUDA 256-thread-block/shared-memory tiling):
//   * one wave64 owns one 16x16 tile; each lane owns a 1x4 pixel column segment
//     (x = lane&15, y = 4*(lane>>4)+k).  No LDS, no __syncthreads: the 64 Gaussians of a batch live
//     one-per-lane in VGPRs and are broadcast with v_readlane_b32 into SGPRs, so the per-pair math reads
//     uniform operands from the scalar file; dx is shared by a lane's 4 pixels.
//   * tile header and both loops run on the scalar unit (readfirstlane); the inner loop is branch-free
//     ("stopped" == T = 0, every update a select, exp2 on a pre-scaled exponent); early termination and the
//     skip of (Gaussian, tile) pairs that touch no pixel are wave ballots.
//   * entry ids / Gaussian ids / records are software-prefetched 3 / 2 / 1 batches ahead.
//   * per-pixel state persists in HBM between depth slices (see binning.hip "depth-sliced binning").
#include "raster_common.h"

namespace gs {

// ---------------------------------------------------------------------------
// forward compositor (single pass and depth-sliced): the per-pixel state (colour without
// background, final T, live T) persists in HBM between slices and a tile whose pixels have all
// stopped is flagged `done` (it gets its background term then, is skipped by later slices and
// receives no further intersections from the binning).  first && last reproduces the unsliced pass.
// ---------------------------------------------------------------------------
struct SliceState {
  unsigned char* tile_done;   // [P*T]
  float* live_T;              // [S,H,W]  0 once a pixel has stopped
  int first, last;
  int* open_flag;             // nullable (zeroed by the caller): += 1 for every tile this launch leaves open
};

// The round-1 compositor (records broadcast with v_readlane; alpha / sigma / T tests as three compares) is TEST
// infrastructure since round 4: compiled only into tests/libgsdeblur_round1.so (-DGS_ROUND1_KERNELS=1, _build.py
// build_round1_library) as the kernel side of the equivalence tests' "plain path" and of the lane-utilisation counters;
// the product library answers GS_ERR_INVALID to variant != 0 and to gs_rasterize_fwd_slice_stats.
#ifndef GS_ROUND1_KERNELS
#define GS_ROUND1_KERNELS 0
#endif
#if GS_ROUND1_KERNELS
// STATS (debug, gs_rasterize_fwd_slice_stats): lane-utilisation counters of the walk, see kLaneStat* below
constexpr int kLaneStats = 13;
template <bool SKIP_EMPTY, bool STATS = false>
__global__ __launch_bounds__(256) void raster_fwd_slice_kernel(RasterParams prm, SliceState st,
                                                               float* __restrict__ out_img,
                                                               float* __restrict__ out_T,
                                                               int* __restrict__ final_idx, unsigned n_blocks,
                                                               unsigned long long* __restrict__ stats = nullptr) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  // wave-uniform tile index in an SGPR: the tile header loads become scalar loads and both loops
  // run on the scalar unit
  const unsigned work = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6)));
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  const size_t tkey = (size_t)p * T + t;
  if (!st.first && st.tile_done[tkey]) return;
  int2 range = prm.tile_bins[tkey];
  range.x = __builtin_amdgcn_readfirstlane(range.x);
  range.y = __builtin_amdgcn_readfirstlane(range.y);
  if (!st.first && !st.last && range.y <= range.x) {         // nothing for this tile in this slice: it stays open
    if (st.open_flag && lane == 0) atomicAdd(st.open_flag, 1);
    return;
  }

  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  float Tk[4], Tf[4], Cr[4], Cg[4], Cb[4], pyf[4];
  int last[4];
  bool inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    inside[k] = px < prm.W && (py0 + k) < prm.H;
    Tk[k] = inside[k] ? 1.f : 0.f; Tf[k] = 1.f; Cr[k] = Cg[k] = Cb[k] = 0.f; last[k] = range.x;
    pyf[k] = (float)(py0 + k) + 0.5f;
    if (!st.first && inside[k]) {
      size_t pix = ((size_t)s * prm.H + (py0 + k)) * prm.W + px;
      Cr[k] = out_img[pix * 3 + 0]; Cg[k] = out_img[pix * 3 + 1]; Cb[k] = out_img[pix * 3 + 2];
      Tf[k] = out_T[pix];
      Tk[k] = st.live_T[pix];
    }
  }
  const int* __restrict__ vals = prm.sorted_vals;
  const int* __restrict__ gi_of_e = prm.gi_of_e;
  // software pipeline over batches of 64 sorted entries: entry ids 3 batches ahead, Gaussian ids 2
  // ahead (one more dependent gather when the list stores emission indices), records 1 ahead
  auto load_id = [&](int i) -> int { return i < range.y ? vals[i] : 0; };
  auto to_gi = [&](int id, int i) -> int { return (gi_of_e && i < range.y) ? gi_of_e[id] : id; };
  int gi_next = to_gi(load_id(range.x + lane), range.x + lane);
  Rec9 rec_next = load_rec(prm.records, gi_next, (range.x + lane) < range.y);
  gi_next = to_gi(load_id(range.x + 64 + lane), range.x + 64 + lane);
  int id_next = load_id(range.x + 128 + lane);
  const float kL2E = -1.4426950408889634f;
  // STATS: [0] entries walked, [1] live pixel hits, [2] geometric pixel hits (alpha >= 1/255, live or not),
  // [3]/[4] 4x4 blocks / 8x8 quadrants with a geometric hit, [5]/[6] steps a 64-entry chunk would take if every 4x4
  // block / 8x8 quadrant walked only its own entries in lock-step (max over the blocks of their entry counts),
  // [7] chunks, [8] entries with a live hit, [9]/[10] blocks / quadrants with a live hit, [11]/[12] as [5]/[6] for
  // live hits
  unsigned long long sc[kLaneStats] = {0};
  unsigned c4g[16], c8g[4], c4l[16], c8l[4];

  for (int batch = range.x; batch < range.y; batch += 64) {
    if (__ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull) break;
    if (STATS) {
#pragma unroll
      for (int b = 0; b < 16; ++b) { c4g[b] = 0; c4l[b] = 0; }
#pragma unroll
      for (int b = 0; b < 4; ++b) { c8g[b] = 0; c8l[b] = 0; }
    }
    Rec9 rec = rec_next;
    rec_next = load_rec(prm.records, gi_next, (batch + 64 + lane) < range.y);
    gi_next = to_gi(id_next, batch + 128 + lane);
    id_next = load_id(batch + 192 + lane);
    rec.cx *= 0.5f * kL2E; rec.cy *= kL2E; rec.cz *= 0.5f * kL2E;
    const int n = min(64, range.y - batch);
    for (int j = 0; j < n; ++j) {
      if ((j & 15) == 15 && __ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull) break;
      const float gx = readlane_f(rec.x, j), gy = readlane_f(rec.y, j);
      const float cx = readlane_f(rec.cx, j), cy = readlane_f(rec.cy, j), cz = readlane_f(rec.cz, j);
      const float op = readlane_f(rec.op, j);
      const float dx = gx - pxf;
      const float hx = cx * dx * dx;
      const float bx = cy * dx;
      float alpha[4];
      bool valid[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = gy - pyf[k];
        const float s2 = hx + dy * (bx + cz * dy);
        alpha[k] = fminf(K::kAlphaMax, op * __builtin_amdgcn_exp2f(s2));
        valid[k] = (s2 <= 0.f) && (alpha[k] >= K::kAlphaMin);
      }
      if (STATS) {
        const bool l0 = valid[0] && Tk[0] > 0.f, l1 = valid[1] && Tk[1] > 0.f, l2 = valid[2] && Tk[2] > 0.f,
                   l3 = valid[3] && Tk[3] > 0.f;
        unsigned long long mg = __ballot(valid[0] || valid[1] || valid[2] || valid[3]), ml = __ballot(l0 || l1 || l2 || l3);
        sc[0] += 1;
        sc[1] += (unsigned)(__popcll(__ballot(l0)) + __popcll(__ballot(l1)) + __popcll(__ballot(l2)) + __popcll(__ballot(l3)));
        sc[2] += (unsigned)(__popcll(__ballot(valid[0])) + __popcll(__ballot(valid[1])) + __popcll(__ballot(valid[2])) +
                            __popcll(__ballot(valid[3])));
        sc[8] += ml != 0ull;
        // lanes 4b .. 4b+3 are the 4x4 pixel block b = (lane >> 4) * 4 + ((lane & 15) >> 2): fold every nibble to bit 0
        mg |= mg >> 1; mg |= mg >> 2; mg &= 0x1111111111111111ull;
        ml |= ml >> 1; ml |= ml >> 2; ml &= 0x1111111111111111ull;
        sc[3] += (unsigned)__popcll(mg);
        sc[9] += (unsigned)__popcll(ml);
#pragma unroll
        for (int b = 0; b < 16; ++b) { c4g[b] += (unsigned)((mg >> (4 * b)) & 1ull); c4l[b] += (unsigned)((ml >> (4 * b)) & 1ull); }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          // quadrant (qx, qy) = blocks (2qx, 2qy) .. (2qx+1, 2qy+1), block index by*4 + bx
          const int b00 = (qd >> 1) * 8 + (qd & 1) * 2;
          const unsigned long long sel = (1ull << (4 * b00)) | (1ull << (4 * (b00 + 1))) | (1ull << (4 * (b00 + 4))) |
                                         (1ull << (4 * (b00 + 5)));
          const unsigned hg = (mg & sel) != 0ull, hl = (ml & sel) != 0ull;
          c8g[qd] += hg; c8l[qd] += hl;
          sc[4] += hg; sc[10] += hl;
        }
      }
      // the tile list comes from a bounding BOX: many (Gaussian, tile) pairs touch no pixel at all
      if (SKIP_EMPTY && __ballot(valid[0] || valid[1] || valid[2] || valid[3]) == 0ull) continue;
      const float cr = readlane_f(rec.r, j), cg = readlane_f(rec.g, j), cb = readlane_f(rec.b, j);
      const int idx1 = batch + j + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float nT = Tk[k] - Tk[k] * alpha[k];
        const bool upd = valid[k] && (nT > K::kTMin);
        const float w = upd ? alpha[k] * Tk[k] : 0.f;
        Cr[k] += w * cr; Cg[k] += w * cg; Cb[k] += w * cb;
        Tf[k] = upd ? nT : Tf[k];
        Tk[k] = upd ? nT : (valid[k] ? 0.f : Tk[k]);
        last[k] = upd ? idx1 : last[k];
      }
    }
    if (STATS) {
      unsigned m4g = 0, m8g = 0, m4l = 0, m8l = 0;
#pragma unroll
      for (int b = 0; b < 16; ++b) { m4g = max(m4g, c4g[b]); m4l = max(m4l, c4l[b]); }
#pragma unroll
      for (int b = 0; b < 4; ++b) { m8g = max(m8g, c8g[b]); m8l = max(m8l, c8l[b]); }
      sc[5] += m4g; sc[6] += m8g; sc[11] += m4l; sc[12] += m8l; sc[7] += 1;
    }
  }
  if (STATS && stats && lane == 0) {
#pragma unroll
    for (int i = 0; i < kLaneStats; ++i) atomicAdd(stats + i, sc[i]);
  }
  const bool all_stopped = __ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull;
  const bool finalize = all_stopped || st.last;
  const float bgr = finalize ? prm.background[0] : 0.f, bgg = finalize ? prm.background[1] : 0.f,
              bgb = finalize ? prm.background[2] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside[k]) {
      size_t pix = ((size_t)s * prm.H + (py0 + k)) * prm.W + px;
      out_img[pix * 3 + 0] = Cr[k] + Tf[k] * bgr;
      out_img[pix * 3 + 1] = Cg[k] + Tf[k] * bgg;
      out_img[pix * 3 + 2] = Cb[k] + Tf[k] * bgb;
      out_T[pix] = Tf[k];
      final_idx[pix] = last[k];
      if (!st.last) st.live_T[pix] = Tk[k];
    }
  }
  if (!st.last && lane == 0) {
    if (all_stopped) st.tile_done[tkey] = 1;
    else if (st.open_flag) atomicAdd(st.open_flag, 1);      // the word counts the tiles left open
  }
}

#endif  // GS_ROUND1_KERNELS

// ---------------------------------------------------------------------------
// Forward compositor, scalar-cache variant (round 2; inner loop rewritten in round 4).  tools/valu_bench*.hip measured
// what instructions cost on gfx950 (wall ns per wave-instruction and SIMD at 4 waves): plain mul / add / mov / int 1.15,
// v_fma 1.36, v_pk_*_f32 2.15, **v_cmp / v_cndmask / v_min / v_max / any SGPR operand 1.9-2.0**, v_exp / v_rcp 3.8,
// **v_readlane_b32 4.6**.  So:
//   * the wave fetches each entry's record with SCALAR loads (s_load_dwordx8 + x2 + x4 off the wave-uniform record
//     index): no VGPR gather, no v_readlane, no VALU slot at all.  The list is walked in aligned groups of four
//     entries (one s_load_dwordx4 of record indices per group); two register sets of two records each: while one pair
//     is blended the loads of the other are in flight (SMEM returns out of order, so every wait is lgkmcnt(0): the
//     `asm volatile` fences pin "wait for the pair, THEN issue the next loads, then blend");
//   * the four pixels of a lane are two hand-packed float2 pairs (v_pk_add/fma/mul_f32);
//   * round 4 — compares and selects were half of the loop's issue time (profiles/r03_valu_mix.txt: 3 compares and 4
//     selects per pixel and entry).  Now ONE compare decides `sigma >= 0 and alpha >= 1/255` (|u| <= nmid on the
//     shifted exponent u = s2 + nmid, constants from the record, gs_math.h rec_aux), alpha = kmul * 2^u needs no
//     separate opacity product, the transmittance update is T -= w (w already gated), and everything a STOPPING
//     pixel needs — its stop index and its removal from the walk — happens in a wave-uniform branch that only runs
//     for entries at which some pixel of the tile stops: a stopped pixel's row coordinate becomes NaN, so it fails
//     the one compare for the rest of the list without any per-entry bookkeeping.  Per pixel and entry: 2 compares,
//     1 select (was 3 + 4).  final_idx now holds, per pixel, the list position at which it stopped (exclusive end of
//     its contributing entries), or the end of the tile's list when it never did: the backward re-tests validity
//     per entry anyway, so `idx < final_idx and valid` selects exactly the entries that were blended.
// Arithmetic differs from the round-1 kernel (raster_fwd_slice_kernel) in rounding only: the exponent carries the
// shift nmid (|nmid| <= 4: 2.4e-7 absolute on u, 1.7e-7 relative on alpha) and T(1 - alpha) is T - alpha T.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

struct RecS { float x, y, cy, r, g, b, d, nmid, kmul, qx, qz; };

template <bool DEPTH>
__device__ __forceinline__ RecS load_rec_s(const float* __restrict__ records, unsigned gi) {
  const float* p = records + (size_t)gi * kRecFloats;
  RecS o;
  o.x = p[0]; o.y = p[1]; o.cy = p[3]; o.r = p[6]; o.g = p[7]; o.b = p[8];
  o.d = DEPTH ? p[9] : 0.f;             // camera-space depth of the splat (record float 9)
  o.nmid = p[kRecNmid]; o.kmul = p[kRecKmul]; o.qx = p[kRecQx]; o.qz = p[kRecQz];
  return o;
}

// py: pixel-centre row coordinate, NaN once the pixel has stopped (or lies outside the image); T: transmittance
// (of a stopped pixel: its final value); fin: see above
struct PixPair { f2 T, Cr, Cg, Cb, Cd, py; };

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// CLAMP=false: no Gaussian of this tile's list has an opacity above 0.999 (the binning flags the tiles that do, see
// tile_hot), so alpha = op * exp(..) <= op can never reach the clamp on a pixel that is blended (those have
// s2 <= 0): min(0.999, .) is the identity there and is not issued.  The choice is made once per tile, outside the
// loop: a per-entry branch costs more in merge copies than the two v_min it saves (measured, round 2 run 22).
template <bool DEPTH, bool CLAMP>
__device__ __forceinline__ void blend_entry(const RecS& rc, float pxf, int idx, PixPair (&pp)[2], int* __restrict__ fin_out,
                                            unsigned fin_off, unsigned fin_row) {
  const float dx = rc.x - pxf;
  const float hxm = fmaf(rc.qx * dx, dx, rc.nmid);          // exponent terms, pre-scaled by -log2(e), + the shift
  const float bx = (rc.cy * kNegLog2e) * dx;
  const f2 hx2 = {hxm, hxm}, bx2 = {bx, bx}, qz2 = {rc.qz, rc.qz}, gy2 = {rc.y, rc.y}, km2 = {rc.kmul, rc.kmul};
  const f2 cr2 = {rc.r, rc.r}, cg2 = {rc.g, rc.g}, cb2 = {rc.b, rc.b};
  f2 w[2], nT[2];
  bool c[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    PixPair& q = pp[h];
    const f2 dy = gy2 - q.py;
    const f2 u = fma2(dy, fma2(qz2, dy, bx2), hx2);
    const f2 ov = km2 * f2{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)};
    const f2 alpha = CLAMP ? f2{fminf(K::kAlphaMax, ov.x), fminf(K::kAlphaMax, ov.y)} : ov;
    // sigma >= 0 and alpha >= 1/255 (false for a stopped pixel: u is NaN)
    const bool v0 = fabsf(u.x) <= rc.nmid, v1 = fabsf(u.y) <= rc.nmid;
    const f2 ag = {v0 ? alpha.x : 0.f, v1 ? alpha.y : 0.f};
    w[h] = ag * q.T;
    nT[h] = q.T - w[h];
    // T never falls to 1e-4 or below while a pixel is live, so "not greater" can only come from this entry's hit
    c[2 * h] = nT[h].x > K::kTMin; c[2 * h + 1] = nT[h].y > K::kTMin;
  }
  if (!(c[0] && c[1] && c[2] && c[3])) {
    // some pixel of this lane stops at this entry (the block is skipped when no lane of the wave has one): it does not
    // blend the entry, keeps its T, leaves the walk, and its stop index goes straight to final_idx — kept in a register
    // and merged after this rarely taken block it cost eight register copies per entry on the path that skips it.
    // In-place selects and an exec-masked store through inline assembly for the same reason (written as C++ the
    // compiler also evaluates the negated compares on the path that skips the block).
    const int idxv = idx;
    static_assert(K::kTMin == 1e-4f, "the literal 0x38d1b717 below is 1e-4f");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      PixPair& q = pp[k >> 1];
      float wk = (k & 1) ? w[k >> 1].y : w[k >> 1].x, pyk = (k & 1) ? q.py.y : q.py.x;
      const float nTk = (k & 1) ? nT[k >> 1].y : nT[k >> 1].x;
      unsigned long long saved;
      asm volatile("v_cmp_lt_f32_e32 vcc, 0x38d1b717, %3\n\t"           // vcc: pixel k goes on (nT > 1e-4)
                   "s_nop 1\n\t"       // gfx950: VALU write of an SGPR -> VALU read needs two wait states, and the
                                        // hazard recogniser cannot see inside inline assembly
                   "v_cndmask_b32_e32 %0, 0, %0, vcc\n\t"
                   "v_cndmask_b32_e32 %1, -1, %1, vcc\n\t"              // all ones: a quiet NaN
                   "s_andn1_saveexec_b64 %2, vcc\n\t"                   // exec := the lanes whose pixel k stops
                   "global_store_dword %4, %5, %6\n\t"
                   "s_mov_b64 exec, %2"
                   : "+v"(wk), "+v"(pyk), "=&s"(saved)
                   : "v"(nTk), "v"((fin_off + (unsigned)k * fin_row) * 4u) /*byte offset*/, "v"(idxv), "s"(fin_out)
                   : "memory", "vcc");
      if (k & 1) { w[k >> 1].y = wk; q.py.y = pyk; } else { w[k >> 1].x = wk; q.py.x = pyk; }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    PixPair& q = pp[h];
    q.Cr = fma2(w[h], cr2, q.Cr); q.Cg = fma2(w[h], cg2, q.Cg); q.Cb = fma2(w[h], cb2, q.Cb);
    if (DEPTH) q.Cd = fma2(w[h], f2{rc.d, rc.d}, q.Cd);     // sum of weight * depth (expected depth = this / alpha)
    q.T -= w[h];                                            // (a stopping pixel's w is zero by now: it keeps its T)
  }
}

// the walk asks "is any pixel of the tile still live?" after EVERY group of four entries (asking every 8 / 16 entries
// was measured at 0.406 / 0.420 ms against 0.395 on the headline: a tile keeps walking past its last stop, and that
// costs more than the five instructions)
__device__ __forceinline__ bool any_live(const PixPair (&pp)[2]) {
  // v_max returns the operand that is not NaN: the maximum is NaN only when all four rows are
  const float m = fmaxf(fmaxf(pp[0].py.x, pp[0].py.y), fmaxf(pp[1].py.x, pp[1].py.y));
  return __builtin_amdgcn_ballot_w64(m == m) != 0ull;
}

// (A quadrant mapping of the lane's four pixels — one per 8x8 quadrant of the tile, quadrants nobody touches skipped
//  by a scalar branch — is what the BACKWARD uses, raster_bwd.hip.  Built for the forward too and measured, visit r4_v5:
//  0.346 vs 0.328 ms on the benchmark scene, 2.10 vs 2.12 ms on the fitted-model-like one — the four branches cost the
//  forward's short per-pixel chain more than the skipped quadrants save.  The two kernels meet in per-pixel arrays, so
//  each keeps the mapping that suits it; both evaluate the validity test on bit-identical operands.)
// the tile's list, front to back (n = range.y - range.x > 0 entries); `idx` handed to an entry = its list position
template <bool DEPTH, bool CLAMP, class State>
__device__ __forceinline__ void fwd_walk(const int* __restrict__ ids, const float* __restrict__ records, unsigned max_id,
                                         int2 range, unsigned n, float pxf, State& pp, int* __restrict__ fin_out,
                                         unsigned fin_off, unsigned fin_row) {
  int b = range.x & ~3;
  const int4* __restrict__ ids4 = reinterpret_cast<const int4*>(ids);
  int4 idv = ids4[b >> 2];
  // indices read in front of / behind the tile's own range belong to other tiles (or to the padding): clamp, the
  // record is loaded but never blended
  RecS a0 = load_rec_s<DEPTH>(records, min((unsigned)idv.x, max_id)), a1 = load_rec_s<DEPTH>(records, min((unsigned)idv.y, max_id));
  for (;;) {
    // pair A (entries b, b+1) is ready; put pair B (b+2, b+3) and the indices of the next group in flight
    asm volatile("" :: "s"(a0.x), "s"(a1.x) : "memory");
    const RecS b0 = load_rec_s<DEPTH>(records, min((unsigned)idv.z, max_id)), b1 = load_rec_s<DEPTH>(records, min((unsigned)idv.w, max_id));
    idv = ids4[(b >> 2) + 1];
    asm volatile("" ::: "memory");
    if ((unsigned)(b - range.x) < n) blend_entry<DEPTH, CLAMP>(a0, pxf, b, pp, fin_out, fin_off, fin_row);
    if ((unsigned)(b + 1 - range.x) < n) blend_entry<DEPTH, CLAMP>(a1, pxf, b + 1, pp, fin_out, fin_off, fin_row);
    // pair B is ready; refill pair A from the next group
    asm volatile("" :: "s"(b0.x), "s"(b1.x), "s"(idv.x) : "memory");
    a0 = load_rec_s<DEPTH>(records, min((unsigned)idv.x, max_id)); a1 = load_rec_s<DEPTH>(records, min((unsigned)idv.y, max_id));
    asm volatile("" ::: "memory");
    if ((unsigned)(b + 2 - range.x) < n) blend_entry<DEPTH, CLAMP>(b0, pxf, b + 2, pp, fin_out, fin_off, fin_row);
    if ((unsigned)(b + 3 - range.x) < n) blend_entry<DEPTH, CLAMP>(b1, pxf, b + 3, pp, fin_out, fin_off, fin_row);
    b += 4;
    if (b >= range.y) break;
    if (!any_live(pp)) break;
  }
}

// per-pixel state accessors shared by the two lane -> pixel mappings
struct PixCols { PixPair p[2]; };                // column mapping: four consecutive rows of one column, as two float2 pairs
__device__ __forceinline__ void pix_set(PixCols& st, int k, float T, float cr, float cg, float cb, float cd, float py) {
  PixPair& q = st.p[k >> 1];
  if (k & 1) { q.T.y = T; q.Cr.y = cr; q.Cg.y = cg; q.Cb.y = cb; q.Cd.y = cd; q.py.y = py; }
  else       { q.T.x = T; q.Cr.x = cr; q.Cg.x = cg; q.Cb.x = cb; q.Cd.x = cd; q.py.x = py; }
}
__device__ __forceinline__ void pix_get(const PixCols& st, int k, float& T, float& cr, float& cg, float& cb, float& cd, float& py) {
  const PixPair& q = st.p[k >> 1];
  if (k & 1) { T = q.T.y; cr = q.Cr.y; cg = q.Cg.y; cb = q.Cb.y; cd = q.Cd.y; py = q.py.y; }
  else       { T = q.T.x; cr = q.Cr.x; cg = q.Cg.x; cb = q.Cb.x; cd = q.Cd.x; py = q.py.x; }
}
template <bool DEPTH, bool CLAMP>
__device__ __forceinline__ void blend_entry(const RecS& rc, float pxf, int idx, PixCols& st, int* __restrict__ fin_out,
                                            unsigned fin_off, unsigned fin_row) {
  blend_entry<DEPTH, CLAMP>(rc, pxf, idx, st.p, fin_out, fin_off, fin_row);
}
__device__ __forceinline__ bool any_live(const PixCols& st) { return any_live(st.p); }

#ifndef GS_FWD_SLOAD_WAVES
#define GS_FWD_SLOAD_WAVES 1      // no occupancy request: 65 VGPRs, 7 waves per SIMD (8 would need 64)
#endif
template <bool DEPTH>
__global__ __launch_bounds__(256, GS_FWD_SLOAD_WAVES) void raster_fwd_sload_kernel(RasterParams prm, SliceState st,
                                                               const int* __restrict__ ids,       // padded, see ABI
                                                               const float* __restrict__ records, unsigned max_id,
                                                               float* __restrict__ out_img,
                                                               float* __restrict__ out_T,
                                                               int* __restrict__ final_idx, unsigned n_blocks,
                                                               float* __restrict__ out_depth,
                                                               const unsigned char* __restrict__ tile_hot) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6)));
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  const size_t tkey = (size_t)p * T + t;
  if (!st.first && st.tile_done[tkey]) return;
  int2 range = prm.tile_bins[tkey];
  range.x = __builtin_amdgcn_readfirstlane(range.x);
  range.y = __builtin_amdgcn_readfirstlane(range.y);
  if (!st.first && !st.last && range.y <= range.x) {         // nothing for this tile in this slice: it stays open
    if (st.open_flag && lane == 0) atomicAdd(st.open_flag, 1);
    return;
  }

  // lane -> its four pixels: pixel k at (px0 + kx(k), py0 + ky(k))
  const int px0 = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  auto kx = [](int) { return 0; };
  auto ky = [](int k) { return k; };
  const float pxf = (float)px0 + 0.5f;
  const float qnan = __builtin_nanf("");
  // element offsets into final_idx [S,H,W] of the lane's first pixel, and of a row (32 bit: S*H*W < 2^31 is checked
  // by the caller's buffer sizes)
  const unsigned fin_off = ((unsigned)s * (unsigned)prm.H + (unsigned)py0) * (unsigned)prm.W + (unsigned)px0;
  const unsigned fin_row = (unsigned)prm.W;
  int* __restrict__ fin_out = final_idx;
  PixCols pp;
  bool inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = px0 + kx(k), py = py0 + ky(k);
    inside[k] = px < prm.W && py < prm.H;
    float Tk = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f;
    bool live = inside[k];
    if (!st.first && inside[k]) {
      size_t pix = ((size_t)s * prm.H + py) * prm.W + px;
      cr = out_img[pix * 3 + 0]; cg = out_img[pix * 3 + 1]; cb = out_img[pix * 3 + 2];
      if (DEPTH) cd = out_depth[pix];
      const float Tf = out_T[pix], lv = st.live_T[pix];
      live = lv > 0.f;
      Tk = live ? lv : Tf;
    }
    const float pyk = live ? (float)py + 0.5f : qnan;
    // final_idx of a pixel that stopped in an earlier slice: it blends nothing of this one.  (A pixel that stops in
    // this slice writes its stop index at that moment, one that stays live gets the end of the list below.)
    if (inside[k] && !live) final_idx[fin_off + (unsigned)ky(k) * fin_row + (unsigned)kx(k)] = range.x;
    pix_set(pp, k, Tk, cr, cg, cb, cd, pyk);
  }
  const unsigned n = (unsigned)(range.y - range.x);
  if (n != 0u) {
    const bool hot = tile_hot == nullptr || __builtin_amdgcn_readfirstlane((int)tile_hot[tkey]) != 0;
    if (hot) fwd_walk<DEPTH, true>(ids, records, max_id, range, n, pxf, pp, fin_out, fin_off, fin_row);
    else fwd_walk<DEPTH, false>(ids, records, max_id, range, n, pxf, pp, fin_out, fin_off, fin_row);
  }
  const bool all_stopped = !any_live(pp);
  const bool finalize = all_stopped || st.last;
  const float bgr = finalize ? prm.background[0] : 0.f, bgg = finalize ? prm.background[1] : 0.f,
              bgb = finalize ? prm.background[2] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside[k]) {
      float Tf, cr, cg, cb, cd, pyk;
      pix_get(pp, k, Tf, cr, cg, cb, cd, pyk);
      size_t pix = ((size_t)s * prm.H + (py0 + ky(k))) * prm.W + (px0 + kx(k));
      out_img[pix * 3 + 0] = cr + Tf * bgr;
      out_img[pix * 3 + 1] = cg + Tf * bgg;
      out_img[pix * 3 + 2] = cb + Tf * bgb;
      out_T[pix] = Tf;
      if (DEPTH) out_depth[pix] = cd;
      if (pyk == pyk) final_idx[pix] = range.y;
      if (!st.last) st.live_T[pix] = pyk == pyk ? Tf : 0.f;
    }
  }
  if (!st.last && lane == 0) {
    if (all_stopped) st.tile_done[tkey] = 1;
    else if (st.open_flag) atomicAdd(st.open_flag, 1);      // the word counts the tiles left open
  }
}

// ---------------------------------------------------------------------------
// sub-frame averaging in linearised colour:
//   out = ( mean_k max(C_k, m)^gamma )^(1/gamma),  m = min_rgb_level/255
// ---------------------------------------------------------------------------
// 4 values per thread (16-byte loads/stores); n4 = n/4 vectors, scalar tail handled by the last threads
__global__ __launch_bounds__(256) void combine_fwd_kernel(int S, size_t n, const float* __restrict__ samples,
                                                          float gamma, float m, float* __restrict__ out) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float invS = 1.f / (float)S, ig = 1.f / gamma;
  if (i + 3 < n) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S; ++k) {
      float4 c = *reinterpret_cast<const float4*>(samples + (size_t)k * n + i);
      acc.x += combine_lin(c.x, gamma, m); acc.y += combine_lin(c.y, gamma, m);
      acc.z += combine_lin(c.z, gamma, m); acc.w += combine_lin(c.w, gamma, m);
    }
    acc.x *= invS; acc.y *= invS; acc.z *= invS; acc.w *= invS;
    if (gamma != 1.f) { acc.x = fast_pow(acc.x, ig); acc.y = fast_pow(acc.y, ig); acc.z = fast_pow(acc.z, ig); acc.w = fast_pow(acc.w, ig); }
    *reinterpret_cast<float4*>(out + i) = acc;
  } else {
    for (size_t j = i; j < n; ++j) {
      float acc = 0.f;
      for (int k = 0; k < S; ++k) acc += combine_lin(samples[(size_t)k * n + j], gamma, m);
      acc *= invS;
      out[j] = gamma != 1.f ? fast_pow(acc, ig) : acc;
    }
  }
}

__global__ __launch_bounds__(256) void combine_bwd_kernel(int S, size_t n, const float* __restrict__ samples,
                                                          float gamma, float m, const float* __restrict__ out,
                                                          const float* __restrict__ v_out,
                                                          float* __restrict__ v_samples) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float invS = 1.f / (float)S;
  // d out / d mean = (1/gamma) mean^(1/gamma - 1) = out^(1-gamma) / gamma
  if (i + 3 < n) {
    float4 o = *reinterpret_cast<const float4*>(out + i);
    float4 vo = *reinterpret_cast<const float4*>(v_out + i);
    float4 d = make_float4(combine_scale(o.x, vo.x, invS, gamma), combine_scale(o.y, vo.y, invS, gamma),
                           combine_scale(o.z, vo.z, invS, gamma), combine_scale(o.w, vo.w, invS, gamma));
    for (int k = 0; k < S; ++k) {
      float4 c = *reinterpret_cast<const float4*>(samples + (size_t)k * n + i);
      float4 g = make_float4(combine_grad(c.x, d.x, gamma, m), combine_grad(c.y, d.y, gamma, m),
                             combine_grad(c.z, d.z, gamma, m), combine_grad(c.w, d.w, gamma, m));
      *reinterpret_cast<float4*>(v_samples + (size_t)k * n + i) = g;
    }
  } else {
    for (size_t j = i; j < n; ++j) {
      float dm = combine_scale(out[j], v_out[j], invS, gamma);
      for (int k = 0; k < S; ++k) v_samples[(size_t)k * n + j] = combine_grad(samples[(size_t)k * n + j], dm, gamma, m);
    }
  }
}

// the sample-independent factor of combine_bwd alone (consumed by the compositor's backward prologue)
__global__ __launch_bounds__(256) void combine_scale_kernel(int S, size_t n, float gamma, const float* __restrict__ out,
                                                            const float* __restrict__ v_out, float* __restrict__ scale,
                                                            bool vec /*all three pointers 16-byte aligned*/) {
  // four elements per thread, 16-byte accesses (one element per thread ran at 3.5 TB/s: 21 us for a 1080p frame)
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float invS = 1.f / (float)S;
  if (vec && i + 3 < n) {
    const float4 o = *reinterpret_cast<const float4*>(out + i), v = *reinterpret_cast<const float4*>(v_out + i);
    *reinterpret_cast<float4*>(scale + i) = make_float4(combine_scale(o.x, v.x, invS, gamma), combine_scale(o.y, v.y, invS, gamma),
                                                        combine_scale(o.z, v.z, invS, gamma), combine_scale(o.w, v.w, invS, gamma));
  } else {
    for (size_t j = i; j < min(n, i + 4); ++j) scale[j] = combine_scale(out[j], v_out[j], invS, gamma);
  }
}

}  // namespace gs

using namespace gs;

// C ABI -----------------------------------------------------------------------
// Replaces the device side of gsplat.rasterize_gaussians' forward
// (_C.rasterize_forward in the absent fork; SURVEY.md §8 a7, boundary §8b).
static int launch_fwd(const RasterParams& prm, const SliceState& st, const int* ids, int n_records, float* out_img,
                      float* out_T, int* final_idx, int variant, hipStream_t stream, float* out_depth = nullptr,
                      const unsigned char* tile_hot = nullptr, unsigned long long* stats = nullptr) {
  unsigned work = (unsigned)(prm.S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
#if GS_ROUND1_KERNELS
  if (stats) {
    if (out_depth) return GS_ERR_INVALID;
    hipLaunchKernelGGL((raster_fwd_slice_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, prm, st, out_img,
                       out_T, final_idx, blocks, stats);
    return GS_OK;
  }
#else
  if (stats) return GS_ERR_INVALID;
#endif
  // n_records == 0: the caller says there is not a single list entry — every tile's range is empty and `ids` is never
  // read (background only)
  const bool sload = variant == 0 && (ids != nullptr || n_records == 0);
  if (out_depth && !sload) return GS_ERR_INVALID;                   // no depth channel in the round-1 kernel
  if (sload && out_depth)
    hipLaunchKernelGGL(raster_fwd_sload_kernel<true>, dim3(blocks), dim3(256), 0, stream, prm, st, ids, prm.records,
                       (unsigned)(n_records > 0 ? n_records - 1 : 0), out_img, out_T, final_idx, blocks, out_depth,
                       tile_hot);
  else if (sload)
    hipLaunchKernelGGL(raster_fwd_sload_kernel<false>, dim3(blocks), dim3(256), 0, stream, prm, st, ids, prm.records,
                       (unsigned)(n_records > 0 ? n_records - 1 : 0), out_img, out_T, final_idx, blocks,
                       (float*)nullptr, tile_hot);
#if GS_ROUND1_KERNELS
  else if (variant == 1)
    hipLaunchKernelGGL(raster_fwd_slice_kernel<false>, dim3(blocks), dim3(256), 0, stream, prm, st, out_img, out_T,
                       final_idx, blocks);
  else
    hipLaunchKernelGGL(raster_fwd_slice_kernel<true>, dim3(blocks), dim3(256), 0, stream, prm, st, out_img, out_T,
                       final_idx, blocks);
#else
  else
    return GS_ERR_INVALID;      // the round-1 compositors (variant 1 / 2, or no record-index list) are not in this build
#endif
  return GS_OK;
}

GS_EXPORT int gs_rasterize_fwd(const float* records, const int* sorted_vals, const int* tile_bins,
                               const int* band_edges, const float* background, int S, int R, int H, int W,
                               float* out_img, float* out_T, int* final_idx, int n_records, int variant,
                               void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, R, H, W);
  // one pass over the complete tile lists = the sliced kernel with first == last (no persistent state)
  SliceState st; st.tile_done = nullptr; st.live_T = nullptr; st.first = 1; st.last = 1; st.open_flag = nullptr;
  int rc = launch_fwd(prm, st, n_records > 0 ? sorted_vals : nullptr, n_records, out_img, out_T, final_idx, variant,
                      (hipStream_t)stream);
  if (rc != GS_OK) return rc;
  return gs_launch_status();
}

// ---- depth-sliced variants (see binning.hip "depth-sliced binning") ---------------------------------
// One forward launch per slice, slices front to back.  out_img / out_T / live_T carry the per-pixel
// state between launches; tile_done [S*R*T] (zeroed by the caller before the first slice) flags tiles
// whose pixels have all stopped.  first/last mark the first and the final slice (first && last ==
// the unsliced pass).  final_idx is per slice (the backward needs one per slice).
// tile_hot (nullable) [S*R*T] u8, from gs_emit_open_intersects: non-zero where the tile's list of THIS slice holds
// a Gaussian with opacity > 0.999; the other tiles run the loop version without the alpha clamp (NULL: all clamp).
// open_flag (nullable, one int zeroed by the caller): the number of tiles still open after this slice — the
// one word the host reads to decide whether the next planned slice has anything to do.
GS_EXPORT int gs_rasterize_fwd_slice(const float* records, const int* sorted_vals, const int* tile_bins,
                                     const int* band_edges, const float* background, int S, int R, int H, int W,
                                     float* out_img, float* out_T, float* live_T, int* final_idx,
                                     unsigned char* tile_done, int first, int last, const int* gi_of_e,
                                     const int* sorted_ids, int n_records, float* out_depth,
                                     const unsigned char* tile_hot, int* open_flag, int variant, void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  // the stop-index store of the scalar-cache compositor forms a 32-bit BYTE offset into final_idx [S,H,W]
  if ((long long)S * H * W >= (1ll << 30)) return GS_ERR_INVALID;
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, R, H, W);
  prm.gi_of_e = gi_of_e;
  SliceState st; st.tile_done = tile_done; st.live_T = live_T; st.first = first; st.last = last; st.open_flag = open_flag;
  const int* ids = sorted_ids ? sorted_ids : (gi_of_e ? nullptr : sorted_vals);
  int rc = launch_fwd(prm, st, n_records > 0 ? ids : nullptr, n_records, out_img, out_T, final_idx, variant,
                      (hipStream_t)stream, out_depth, tile_hot);
  if (rc != GS_OK) return rc;
  return gs_launch_status();
}

// Debug twin of gs_rasterize_fwd_slice (same results through the round-1 compositor) that also accumulates the
// lane-utilisation counters of its walk into stats[13] (u64, caller zeroes; see raster_fwd_slice_kernel STATS).
GS_EXPORT int gs_rasterize_fwd_slice_stats(const float* records, const int* sorted_vals, const int* tile_bins,
                                           const int* band_edges, const float* background, int S, int R, int H, int W,
                                           float* out_img, float* out_T, float* live_T, int* final_idx,
                                           unsigned char* tile_done, int first, int last, const int* gi_of_e,
                                           int* open_flag, unsigned long long* stats, void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0 || !stats) return GS_ERR_INVALID;
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, R, H, W);
  prm.gi_of_e = gi_of_e;
  SliceState st; st.tile_done = tile_done; st.live_T = live_T; st.first = first; st.last = last; st.open_flag = open_flag;
  int rc = launch_fwd(prm, st, nullptr, 0, out_img, out_T, final_idx, 1, (hipStream_t)stream, nullptr, nullptr, stats);
  if (rc != GS_OK) return rc;
  return gs_launch_status();
}

GS_EXPORT int gs_combine_fwd(int S, long long n, const float* samples, float gamma, float min_level,
                             float* out, void* stream) {
  if (S <= 0 || n <= 0) return GS_ERR_INVALID;
  unsigned blocks = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(combine_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (size_t)n, samples,
                     gamma, min_level, out);
  return gs_launch_status();
}

GS_EXPORT int gs_combine_bwd(int S, long long n, const float* samples, float gamma, float min_level,
                             const float* out, const float* v_out, float* v_samples, void* stream) {
  if (S <= 0 || n <= 0) return GS_ERR_INVALID;
  unsigned blocks = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(combine_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (size_t)n, samples,
                     gamma, min_level, out, v_out, v_samples);
  return gs_launch_status();
}

// scale [n] = (1/S) * d out / d mean * v_out: what gs_rasterize_bwd_slice(cmb_scale=...) consumes to derive the
// per-sample gradients itself instead of reading the [S,n] tensor gs_combine_bwd would write.
GS_EXPORT int gs_combine_bwd_scale(int S, long long n, float gamma, const float* out, const float* v_out,
                                   float* scale, void* stream) {
  if (S <= 0 || n <= 0) return GS_ERR_INVALID;
  // float4 accesses when the three pointers allow it (a gradient that is a view into a larger tensor may not)
  const bool vec = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(v_out) |
                     reinterpret_cast<uintptr_t>(scale)) & 15u) == 0;
  unsigned blocks = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(combine_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (size_t)n, gamma, out,
                     v_out, scale, vec);
  return gs_launch_status();
}
