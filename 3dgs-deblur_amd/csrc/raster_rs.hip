// raster_rs.hip — compositors of the pixel-velocity model with EXACT per-row rolling-shutter time (round 3).
//
// The paper's model (SURVEY.md App. A "Paper's blur/RS model"; the fork's flag /root/reference/train.py:56
// `rolling-shutter-compensation`, its data field /root/reference/render_video.py:242-243 `rolling_shutter_time`, its
// changelog /root/reference/README.md:200 "pixel velocity formulas"): a splat's centre moves linearly in pixels,
// mu'(t) = mu' + t * v', and pixel ROW y is exposed at t_s + tau(y), tau(y) = ((y + 0.5)/H - 0.5) * T_ro.  Rounds 1-2
// discretised tau into R tile-row bands (R projections / sorts / lists per blur sample).  Here the row term lives in
// the compositor: ONE record per (blur sample, Gaussian) — centred at mu' + t_s v' by gs_project_pixvel_fwd, which
// also widens its tile box by the sweep and hands out v' [N,2] — and every pixel evaluates the splat at
//     d = (mu' + t_s v' + tau(y) v') - pixel,
// so projection, binning and sorting cost what a frame WITHOUT rolling shutter costs.  The backward returns, beside the
// nine gradients of the other compositors, d loss / d v' (tuple slots 9 and 10: sum over the pixels of tau(y) * d/d mu').
//
// Round 6: brought to the kernel generation of raster.hip / raster_bwd.hip (rounds 3-5 were "written for clarity"):
//   * records AND pixel velocities through the scalar cache in aligned groups of four list entries, two register sets
//     of two entries each (one pair's loads in flight while the other pair is blended);
//   * ONE compare |u| <= nmid on the shifted exponent decides sigma >= 0 and alpha >= 1/255 (gs_math.h rec_aux; the
//     same expression, operand for operand, in both directions: `rs_u`), alpha = kmul * 2^u;
//   * forward: the lane's four pixels as two hand-packed float2 pairs, a stopped pixel's row coordinate is NaN and
//     fails the compare for the rest of the list, the stop index is stored by the rarely taken stop block;
//   * backward: quadrant mapping (pixel k of a lane in 8x8 quadrant k: a quadrant nobody's pixel blends is skipped by
//     a scalar branch), the geometric gradients from eight moments of v_sigma in (dx, dy, tau) — dx is no longer a
//     lane constant here, so the three-moment form of raster_bwd.hip does not apply —, eleven partial sums pair-added
//     by DPP, 44 rows x 36 floats of wave-private LDS, lanes 0..43 store row totals straight into the entry's tuple.
#include "raster_common.h"

namespace gs {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct RsParams {
  const float* pix_vel;   // [N,2] pixel velocity of every Gaussian (gs_project_pixvel_fwd)
  int N;
  float rs_time;          // readout time T_ro (same unit as the sub-pose times)
  // round 4 — ONE list for all blur samples (the form the paper describes, SURVEY.md App. A: "depth order and
  // covariance fixed across samples", /root/reference/README.md:196-200): the records are the mid-exposure splats (one
  // per Gaussian, tile boxes swept over the whole exposure + readout), binned and sorted ONCE; sample s walks the same
  // tile list and evaluates every splat at mu' + (times[s] + tau(y)) v'.  times == NULL: one record set and one list
  // per sample (records already centred at mu' + t_s v'; round 3).
  const float* times;     // [S] sample times, or NULL
};

struct RsSliceState {
  unsigned char* tile_done;
  float* live_T;
  int first, last;
  int* open_flag;
};

// exposure time of pixel-centre row pyf relative to the record's centre time: the SAME expression in both kernels
__device__ __forceinline__ float rs_tau(float pyf, float H, float rs_time, float t_s) {
  return fmaf(pyf / H - 0.5f, rs_time, t_s);
}

// shifted exponent u = nmid - log2(e) * sigma at a pixel exposed tau after the record's centre time (scalar form; the
// forward evaluates the packed twin below on identical operands, so both directions take the same decision)
__device__ __forceinline__ float rs_u(float dx0, float dy0, float tau, float pvx, float pvy, float qx, float qy, float qz,
                                      float nmid, float& dx, float& dy) {
  dx = fmaf(tau, pvx, dx0);
  dy = fmaf(tau, pvy, dy0);
  return fmaf(dx, fmaf(qx, dx, qy * dy), fmaf(qz * dy, dy, nmid));
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct RsRecF { float x, y, qy, r, g, b, d, nmid, kmul, qx, qz, pvx, pvy; };

template <bool DEPTH>
__device__ __forceinline__ RsRecF load_rs_f(const float* __restrict__ records, const float* __restrict__ pix_vel,
                                            unsigned gi, unsigned g) {
  const float* p = records + (size_t)gi * kRecFloats;
  RsRecF o;
  o.x = p[0]; o.y = p[1]; o.qy = p[3] * kNegLog2e; o.r = p[6]; o.g = p[7]; o.b = p[8];
  o.d = DEPTH ? p[9] : 0.f;
  o.nmid = p[kRecNmid]; o.kmul = p[kRecKmul]; o.qx = p[kRecQx]; o.qz = p[kRecQz];
  o.pvx = pix_vel[2 * (size_t)g]; o.pvy = pix_vel[2 * (size_t)g + 1];
  return o;
}

// py: pixel-centre row coordinate, NaN once the pixel has stopped (or lies outside the image); tau: its exposure time
struct RsPair { f2 T, Cr, Cg, Cb, Cd, py, tau; };

template <bool DEPTH>
__device__ __forceinline__ void rs_blend(const RsRecF& rc, float pxf, int idx, RsPair (&pp)[2], int* __restrict__ fin_out,
                                         unsigned fin_off, unsigned fin_row) {
  const float dx0 = rc.x - pxf;
  const f2 dx02 = {dx0, dx0}, gy2 = {rc.y, rc.y}, pvx2 = {rc.pvx, rc.pvx}, pvy2 = {rc.pvy, rc.pvy};
  const f2 qx2 = {rc.qx, rc.qx}, qy2 = {rc.qy, rc.qy}, qz2 = {rc.qz, rc.qz}, nm2 = {rc.nmid, rc.nmid}, km2 = {rc.kmul, rc.kmul};
  const f2 cr2 = {rc.r, rc.r}, cg2 = {rc.g, rc.g}, cb2 = {rc.b, rc.b};
  f2 w[2], nT[2];
  bool c[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    RsPair& q = pp[h];
    const f2 dx = fma2(q.tau, pvx2, dx02);
    const f2 dy = fma2(q.tau, pvy2, gy2 - q.py);                 // NaN for a stopped pixel
    const f2 u = fma2(dx, fma2(qx2, dx, qy2 * dy), fma2(qz2 * dy, dy, nm2));
    const f2 ov = km2 * f2{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)};
    const f2 alpha = {fminf(K::kAlphaMax, ov.x), fminf(K::kAlphaMax, ov.y)};
    const bool v0 = fabsf(u.x) <= rc.nmid, v1 = fabsf(u.y) <= rc.nmid;   // sigma >= 0 and alpha >= 1/255
    const f2 ag = {v0 ? alpha.x : 0.f, v1 ? alpha.y : 0.f};
    w[h] = ag * q.T;
    nT[h] = q.T - w[h];
    c[2 * h] = nT[h].x > K::kTMin; c[2 * h + 1] = nT[h].y > K::kTMin;
  }
  if (!(c[0] && c[1] && c[2] && c[3])) {
    // some pixel of this lane stops at this entry (rare; wave-uniformly skipped otherwise): it does not blend the
    // entry, keeps its T, leaves the walk (row coordinate := NaN) and its stop index goes straight to final_idx
    // (raster.hip blend_entry has the reasons for the inline assembly)
    const int idxv = idx;
    static_assert(K::kTMin == 1e-4f, "the literal 0x38d1b717 below is 1e-4f");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      RsPair& q = pp[k >> 1];
      float wk = (k & 1) ? w[k >> 1].y : w[k >> 1].x, pyk = (k & 1) ? q.py.y : q.py.x;
      const float nTk = (k & 1) ? nT[k >> 1].y : nT[k >> 1].x;
      unsigned long long saved;
      asm volatile("v_cmp_lt_f32_e32 vcc, 0x38d1b717, %3\n\t"
                   "s_nop 1\n\t"
                   "v_cndmask_b32_e32 %0, 0, %0, vcc\n\t"
                   "v_cndmask_b32_e32 %1, -1, %1, vcc\n\t"
                   "s_andn1_saveexec_b64 %2, vcc\n\t"
                   "global_store_dword %4, %5, %6\n\t"
                   "s_mov_b64 exec, %2"
                   : "+v"(wk), "+v"(pyk), "=&s"(saved)
                   : "v"(nTk), "v"((fin_off + (unsigned)k * fin_row) * 4u), "v"(idxv), "s"(fin_out)
                   : "memory", "vcc");
      if (k & 1) { w[k >> 1].y = wk; q.py.y = pyk; } else { w[k >> 1].x = wk; q.py.x = pyk; }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    RsPair& q = pp[h];
    q.Cr = fma2(w[h], cr2, q.Cr); q.Cg = fma2(w[h], cg2, q.Cg); q.Cb = fma2(w[h], cb2, q.Cb);
    if (DEPTH) q.Cd = fma2(w[h], f2{rc.d, rc.d}, q.Cd);
    q.T -= w[h];
  }
}

__device__ __forceinline__ bool rs_any_live(const RsPair (&pp)[2]) {
  const float m = fmaxf(fmaxf(pp[0].py.x, pp[0].py.y), fmaxf(pp[1].py.x, pp[1].py.y));   // NaN only when all four are
  return __builtin_amdgcn_ballot_w64(m == m) != 0ull;
}

// the tile's list, front to back, in aligned groups of four entries (ids padded by the caller, see the ABI)
template <bool DEPTH>
__device__ __forceinline__ void rs_fwd_walk(const int* __restrict__ ids, const float* __restrict__ records,
                                            const float* __restrict__ pix_vel, unsigned s_base, unsigned max_id,
                                            unsigned max_g, int2 range, unsigned n, float pxf, RsPair (&pp)[2],
                                            int* __restrict__ fin_out, unsigned fin_off, unsigned fin_row) {
  int b = range.x & ~3;
  const int4* __restrict__ ids4 = reinterpret_cast<const int4*>(ids);
  int4 idv = ids4[b >> 2];
  // indices read in front of / behind the tile's own range belong to other tiles (or to the padding): clamped, the
  // record is loaded but never blended
  auto rec = [&](int id) {
    const unsigned gi = min((unsigned)id, max_id);
    return load_rs_f<DEPTH>(records, pix_vel, gi, min(gi - s_base, max_g));
  };
  RsRecF a0 = rec(idv.x), a1 = rec(idv.y);
  for (;;) {
    asm volatile("" :: "s"(a0.x), "s"(a1.x), "s"(a0.pvx), "s"(a1.pvx) : "memory");
    const RsRecF b0 = rec(idv.z), b1 = rec(idv.w);
    idv = ids4[(b >> 2) + 1];
    asm volatile("" ::: "memory");
    if ((unsigned)(b - range.x) < n) rs_blend<DEPTH>(a0, pxf, b, pp, fin_out, fin_off, fin_row);
    if ((unsigned)(b + 1 - range.x) < n) rs_blend<DEPTH>(a1, pxf, b + 1, pp, fin_out, fin_off, fin_row);
    asm volatile("" :: "s"(b0.x), "s"(b1.x), "s"(b0.pvx), "s"(b1.pvx), "s"(idv.x) : "memory");
    a0 = rec(idv.x); a1 = rec(idv.y);
    asm volatile("" ::: "memory");
    if ((unsigned)(b + 2 - range.x) < n) rs_blend<DEPTH>(b0, pxf, b + 2, pp, fin_out, fin_off, fin_row);
    if ((unsigned)(b + 3 - range.x) < n) rs_blend<DEPTH>(b1, pxf, b + 3, pp, fin_out, fin_off, fin_row);
    b += 4;
    if (b >= range.y) break;
    if (!rs_any_live(pp)) break;
  }
}

template <bool DEPTH>
__global__ __launch_bounds__(256) void raster_fwd_rs_kernel(RasterParams prm, RsParams rs, RsSliceState st,
                                                            const int* __restrict__ ids, const float* __restrict__ records,
                                                            unsigned max_id, float* __restrict__ out_img,
                                                            float* __restrict__ out_T, int* __restrict__ final_idx,
                                                            unsigned n_blocks, float* __restrict__ out_depth) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6)));
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const size_t tkey = (size_t)s * T + t;                      // R == 1: sub-pose = sample
  const bool shared = rs.times != nullptr;                    // all samples walk sub-pose 0's list
  int2 range = prm.tile_bins[shared ? (size_t)t : tkey];
  range.x = __builtin_amdgcn_readfirstlane(range.x);
  range.y = __builtin_amdgcn_readfirstlane(range.y);
  const int px0 = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const unsigned fin_off = ((unsigned)s * (unsigned)prm.H + (unsigned)py0) * (unsigned)prm.W + (unsigned)px0;
  const unsigned fin_row = (unsigned)prm.W;
  if (!st.first && st.tile_done[tkey]) {
    // A finished (sample, tile) of a SHARED list: the tile's list still receives entries while any other sample's tile
    // is open (the binning sees the AND over the samples), so this slice's backward walks a non-empty range for this
    // sample too and reads this slice's stop indices — which are fresh, never memset arena memory.
    // Every pixel stopped before this slice: its stop index is the slice's first entry.
    if (shared) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (px0 < prm.W && (py0 + k) < prm.H) final_idx[fin_off + (unsigned)k * fin_row] = range.x;
    }
    return;
  }
  const float t_s = shared ? rs.times[s] : 0.f;
  if (!st.first && !st.last && range.y <= range.x) {
    if (st.open_flag && lane == 0) atomicAdd(st.open_flag, 1);
    return;
  }
  const float pxf = (float)px0 + 0.5f;
  const float qnan = __builtin_nanf("");
  RsPair pp[2];
  bool inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int py = py0 + k;
    inside[k] = px0 < prm.W && py < prm.H;
    float Tk = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f;
    bool live = inside[k];
    if (!st.first && inside[k]) {
      const size_t pix = ((size_t)s * prm.H + py) * prm.W + px0;
      cr = out_img[pix * 3 + 0]; cg = out_img[pix * 3 + 1]; cb = out_img[pix * 3 + 2];
      if (DEPTH) cd = out_depth[pix];
      const float Tf = out_T[pix], lv = st.live_T[pix];
      live = lv > 0.f;
      Tk = live ? lv : Tf;
    }
    const float pyc = (float)py + 0.5f;
    const float pyk = live ? pyc : qnan;
    const float tk = rs_tau(pyc, (float)prm.H, rs.rs_time, t_s);
    // a pixel that stopped in an earlier slice blends nothing of this one (one that stops in this slice writes its stop
    // index at that moment, one that stays live gets the end of the list below)
    if (inside[k] && !live) final_idx[fin_off + (unsigned)k * fin_row] = range.x;
    RsPair& q = pp[k >> 1];
    if (k & 1) { q.T.y = Tk; q.Cr.y = cr; q.Cg.y = cg; q.Cb.y = cb; q.Cd.y = cd; q.py.y = pyk; q.tau.y = tk; }
    else       { q.T.x = Tk; q.Cr.x = cr; q.Cg.x = cg; q.Cb.x = cb; q.Cd.x = cd; q.py.x = pyk; q.tau.x = tk; }
  }
  const unsigned n = (unsigned)(range.y - range.x);
  const unsigned s_base = shared ? 0u : (unsigned)s * (unsigned)rs.N;
  if (n != 0u)
    rs_fwd_walk<DEPTH>(ids, records, rs.pix_vel, s_base, max_id, (unsigned)(rs.N - 1), range, n, pxf, pp, final_idx,
                       fin_off, fin_row);
  const bool all_stopped = !rs_any_live(pp);
  const bool finalize = all_stopped || st.last;
  const float bgr = finalize ? prm.background[0] : 0.f, bgg = finalize ? prm.background[1] : 0.f,
              bgb = finalize ? prm.background[2] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside[k]) {
      const RsPair& q = pp[k >> 1];
      const float Tf = (k & 1) ? q.T.y : q.T.x, cr = (k & 1) ? q.Cr.y : q.Cr.x, cg = (k & 1) ? q.Cg.y : q.Cg.x;
      const float cb = (k & 1) ? q.Cb.y : q.Cb.x, cd = (k & 1) ? q.Cd.y : q.Cd.x, pyk = (k & 1) ? q.py.y : q.py.x;
      const size_t pix = ((size_t)s * prm.H + (py0 + k)) * prm.W + px0;
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

// ---------------------------------------------------------------------------------------------------------------
// backward: reverse walk, 11 partial sums per (entry, lane) — x, y, conic (3), opacity, colour (3), pixel velocity
// (2) — pair-added by DPP, transposed through wave-private LDS in groups of four entries, row totals straight into the
// entry's tuple
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRsComp = 11;
constexpr int kRsGroup = 4;
constexpr int kRsCols = 32, kRsStride = 36;          // 32 pair sums per row + 4: rows 16-byte aligned, b128 reads conflict-free
constexpr int kRsFloats = kRsGroup * kRsComp * kRsStride;

struct RsRecB { float x, y, cx, cy, cz, r, g, b, nmid, kmul, qx, qz, pvx, pvy; };

__device__ __forceinline__ RsRecB load_rs_b(const float* __restrict__ records, const float* __restrict__ pix_vel,
                                            unsigned gi, unsigned g) {
  const float* p = records + (size_t)gi * kRecFloats;
  RsRecB o;
  o.x = p[0]; o.y = p[1]; o.cx = p[2]; o.cy = p[3]; o.cz = p[4]; o.r = p[6]; o.g = p[7]; o.b = p[8];
  o.nmid = p[kRecNmid]; o.kmul = p[kRecKmul]; o.qx = p[kRecQx]; o.qz = p[kRecQz];
  o.pvx = pix_vel[2 * (size_t)g]; o.pvy = pix_vel[2 * (size_t)g + 1];
  return o;
}

// w[i] = v[i](lane) + v[i](lane ^ 1) for eleven values (see raster_bwd.hip pair_sum9 for the s_nop)
__device__ __forceinline__ void pair_sum11(float (&v)[11]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
        "+v"(v[9]), "+v"(v[10]));
}

// pixel k of a lane lies in the 8x8 quadrant (k & 1, k >> 1) of the tile; its row decides tau: two values per lane
struct RsQuad { float T[4], Dv[4], vr[4], vg[4], vb[4], py[4], tau[2]; int fin[4]; };

// One list entry against the lane's four pixels (returns whether any lane of the wave was hit; then the lane pair's 11
// partial sums are in LDS rows slot*11 .. slot*11+10, column lane >> 1).  Slot 5 carries the plain sum of v_sigma (tuple
// flag 2: the tuple reduce divides by -opacity once per Gaussian).
__device__ __forceinline__ bool rs_bwd_entry(const RsRecB& rc, float pxf, int idx, RsQuad& pp, float* __restrict__ red,
                                             int slot, int lane, float agm) {
  const float dxa = rc.x - pxf, dxb = rc.x - (pxf + 8.0f);
  const float qy = rc.cy * kNegLog2e;
  float M00 = 0.f, M10 = 0.f, M01 = 0.f, M20 = 0.f, M11 = 0.f, M02 = 0.f, T10 = 0.f, T01 = 0.f;
  float q_r = 0.f, q_g = 0.f, q_b = 0.f;
  bool any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float tk = pp.tau[k >> 1];
    float dx, dy;
    const float u = rs_u((k & 1) ? dxb : dxa, rc.y - pp.py[k], tk, rc.pvx, rc.pvy, rc.qx, qy, rc.qz, rc.nmid, dx, dy);
    const bool hit = (idx < pp.fin[k]) && (fabsf(u) <= rc.nmid);
    if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;           // nobody's quadrant-k pixel blended this entry
    any = true;
    const float ov = rc.kmul * __builtin_amdgcn_exp2f(u);
    // pixels that are not hit are neutralised by SELECTING alpha = 0 (1/(1-0) = 1 exactly, every term an exact zero)
    const float alpha = hit ? fminf(K::kAlphaMax, ov) : 0.f;
    const float ovm = (hit && ov <= agm) ? ov : 0.f;                  // d min(0.999, o*vis) = 0 when clamped
    const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
    pp.T[k] *= ra;                                                    // transmittance in front of this entry
    const float fac = alpha * pp.T[k];
    q_r = fmaf(fac, pp.vr[k], q_r); q_g = fmaf(fac, pp.vg[k], q_g); q_b = fmaf(fac, pp.vb[k], q_b);
    const float cv = fmaf(rc.b, pp.vb[k], fmaf(rc.g, pp.vg[k], rc.r * pp.vr[k]));
    const float v_al = fmaf(pp.T[k], cv, -(ra * pp.Dv[k]));
    pp.Dv[k] = fmaf(fac, cv, pp.Dv[k]);
    const float v_sigma = -ovm * v_al;
    const float vsx = v_sigma * dx, vsy = v_sigma * dy;
    M00 += v_sigma; M10 += vsx; M01 += vsy;
    M20 = fmaf(vsx, dx, M20); M11 = fmaf(vsx, dy, M11); M02 = fmaf(vsy, dy, M02);
    T10 = fmaf(tk, vsx, T10); T01 = fmaf(tk, vsy, T01);               // centre = mu' + (t_s + tau) v'
  }
  if (!any) return false;
  float w[11];
  w[0] = fmaf(rc.cx, M10, rc.cy * M01);                               // d sigma / d centre
  w[1] = fmaf(rc.cy, M10, rc.cz * M01);
  w[2] = 0.5f * M20; w[3] = M11; w[4] = 0.5f * M02;
  w[5] = M00;
  w[6] = q_r; w[7] = q_g; w[8] = q_b;
  w[9] = fmaf(rc.cx, T10, rc.cy * T01);                               // d / d pixel velocity
  w[10] = fmaf(rc.cy, T10, rc.cz * T01);
  pair_sum11(w);
  if ((lane & 1) == 0) {
    float* r0 = red + slot * (kRsComp * kRsStride) + (lane >> 1);
#pragma unroll
    for (int c = 0; c < kRsComp; ++c) r0[c * kRsStride] = w[c];
  }
  return true;
}

template <bool STATE>
__global__ __launch_bounds__(256) void raster_bwd_rs_kernel(
    RasterParams prm, RsParams rs, const int* __restrict__ ids, const int* __restrict__ eids,
    const float* __restrict__ records, unsigned max_id, const float* __restrict__ out_T,
    const int* __restrict__ final_idx, const float* __restrict__ v_img, const float* __restrict__ v_alpha,
    unsigned n_blocks, float* __restrict__ bwd_T, float* __restrict__ bwd_B, float* __restrict__ tuples,
    unsigned char* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) float lds_all[4 * kRsFloats];
  const int lane = lane_id();
  float* red = lds_all + (threadIdx.x >> 6) * kRsFloats;
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6)));
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const bool shared = rs.times != nullptr;
  const float t_s = shared ? rs.times[s] : 0.f;
  int2 range = prm.tile_bins[shared ? (size_t)t : (size_t)s * T + t];
  range.x = __builtin_amdgcn_readfirstlane(range.x);
  range.y = __builtin_amdgcn_readfirstlane(range.y);
  if (range.y <= range.x) return;
  // lane -> its four pixels, one per 8x8 quadrant: pixel k at (px0 + 8 (k & 1), py0 + 8 (k >> 1))
  const int px0 = tx * K::kTile + (lane & 7);
  const int py0 = ty * K::kTile + (lane >> 3);
  const float pxf = (float)px0 + 0.5f;
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];
  RsQuad pp;
  int my_end = range.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
    float Tk = 1.f, Dv = 0.f, vr = 0.f, vg = 0.f, vb = 0.f;
    int fin = range.x;
    if (x < prm.W && y < prm.H) {
      const size_t pix = ((size_t)s * prm.H + y) * prm.W + x;
      const float Tfin = out_T[pix];
      fin = min(max(final_idx[pix], range.x), range.y);      // a stop index never leaves the tile's list
      vr = v_img[pix * 3 + 0]; vg = v_img[pix * 3 + 1]; vb = v_img[pix * 3 + 2];
      if (prm.cmb_scale) {
        const size_t q = ((size_t)y * prm.W + x) * 3;
        vr = combine_grad(vr, prm.cmb_scale[q + 0], prm.cmb_gamma, prm.cmb_min);
        vg = combine_grad(vg, prm.cmb_scale[q + 1], prm.cmb_gamma, prm.cmb_min);
        vb = combine_grad(vb, prm.cmb_scale[q + 2], prm.cmb_gamma, prm.cmb_min);
      }
      const float va_out = v_alpha ? v_alpha[pix] : 0.f;
      const float va = Tfin * (va_out - (bgr * vr + bgg * vg + bgb * vb));
      Tk = Tfin;
      Dv = -va;
      if (STATE) { Tk = bwd_T[pix]; Dv = bwd_B[pix] - va; }
    }
    my_end = max(my_end, fin);
    pp.T[k] = Tk; pp.Dv[k] = Dv; pp.vr[k] = vr; pp.vg[k] = vg; pp.vb[k] = vb; pp.py[k] = (float)y + 0.5f; pp.fin[k] = fin;
  }
  pp.tau[0] = rs_tau((float)py0 + 0.5f, (float)prm.H, rs.rs_time, t_s);
  pp.tau[1] = rs_tau((float)(py0 + 8) + 0.5f, (float)prm.H, rs.rs_time, t_s);
  const int wave_end = __builtin_amdgcn_readfirstlane(wave_max_i(my_end));
  const float agm = prm.alpha_grad_max;
  const unsigned s_base = shared ? 0u : (unsigned)s * (unsigned)rs.N;
  const unsigned max_g = (unsigned)(rs.N - 1);
  // shared list: the S samples' waves write the SAME entry's gradients — every (entry, sample) pair gets a tuple of its
  // own, e * S + s (the S tuples of an entry are adjacent, a Gaussian's tuples stay one contiguous range)
  const unsigned tmul = shared ? (unsigned)prm.S : 1u, tofs = shared ? (unsigned)s : 0u;
  const unsigned n = (unsigned)(wave_end - range.x);
  if (n != 0u) {
    const int row = lane;                                    // row-sum role: lanes 0..43
    const int row_g = row / kRsComp, row_c = row - row_g * kRsComp;
    const int4* __restrict__ ids4 = reinterpret_cast<const int4*>(ids);
    const int4* __restrict__ eids4 = reinterpret_cast<const int4*>(eids);
    auto rec = [&](int id) {
      const unsigned gi = min((unsigned)id, max_id);
      return load_rs_b(records, rs.pix_vel, gi, min(gi - s_base, max_g));
    };
    int b = (wave_end - 1) & ~3;
    const int b_last = range.x & ~3;
    int4 idv = ids4[b >> 2];
    RsRecB a0 = rec(idv.w), a1 = rec(idv.z);
    for (;;) {
      // pair A (entries b+3, b+2) is ready; put pair B (b+1, b) and the indices of the next (lower) group in flight
      asm volatile("" :: "s"(a0.x), "s"(a1.x), "s"(a0.pvx), "s"(a1.pvx) : "memory");
      const RsRecB b0 = rec(idv.y), b1 = rec(idv.x);
      const int4 ev = eids4[b >> 2];
      idv = ids4[max(b - 4, 0) >> 2];
      asm volatile("" ::: "memory");
      unsigned filled = 0;
      if ((unsigned)(b + 3 - range.x) < n && rs_bwd_entry(a0, pxf, b + 3, pp, red, 3, lane, agm)) filled |= 8u;
      if ((unsigned)(b + 2 - range.x) < n && rs_bwd_entry(a1, pxf, b + 2, pp, red, 2, lane, agm)) filled |= 4u;
      asm volatile("" :: "s"(b0.x), "s"(b1.x), "s"(b0.pvx), "s"(b1.pvx), "s"(idv.x), "s"(ev.x) : "memory");
      a0 = rec(idv.w); a1 = rec(idv.z);
      asm volatile("" ::: "memory");
      if ((unsigned)(b + 1 - range.x) < n && rs_bwd_entry(b0, pxf, b + 1, pp, red, 1, lane, agm)) filled |= 2u;
      if ((unsigned)(b - range.x) < n && rs_bwd_entry(b1, pxf, b, pp, red, 0, lane, agm)) filled |= 1u;
      if (filled) {
        __builtin_amdgcn_wave_barrier();
        if (row < kRsGroup * kRsComp && ((filled >> row_g) & 1u)) {
          const f4* rp = reinterpret_cast<const f4*>(red + row * kRsStride);
          f4 s0 = rp[0], s1 = rp[1], s2 = rp[2], s3 = rp[3];
          s0 += rp[4]; s1 += rp[5]; s2 += rp[6]; s3 += rp[7];
          const f4 v = (s0 + s1) + (s2 + s3);
          const float sum = (v.x + v.y) + (v.z + v.w);
          const int id_e = row_g == 0 ? ev.x : (row_g == 1 ? ev.y : (row_g == 2 ? ev.z : ev.w));
          const size_t e = (size_t)(unsigned)id_e * tmul + tofs;
          tuples[e * kGradFloats + row_c] = sum;
          if (row_c == 0) flags[e] = 2;                      // 2: slot 5 is the plain sum of v_sigma
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (b <= b_last) break;
      b -= 4;
    }
  }
  if (STATE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = px0 + 8 * (k & 1), y = py0 + 8 * (k >> 1);
      if (x < prm.W && y < prm.H) {
        const size_t pix = ((size_t)s * prm.H + y) * prm.W + x;
        const float Tfin = out_T[pix];
        const float va_out = v_alpha ? v_alpha[pix] : 0.f;
        const float va = Tfin * (va_out - (bgr * pp.vr[k] + bgg * pp.vg[k] + bgb * pp.vb[k]));
        bwd_T[pix] = pp.T[k];
        bwd_B[pix] = pp.Dv[k] + va;
      }
    }
  }
}

}  // namespace gs

using namespace gs;

// Forward of one depth slice with exact per-row rolling-shutter time (pixel-velocity model; R = 1: sub-pose = blur
// sample).  Same state protocol as gs_rasterize_fwd_slice; sorted_ids [I+8] = record index of every sorted entry;
// pix_vel [N,2] from gs_project_pixvel_fwd(rolling_shutter_time != 0).  No upstream counterpart in the tree: the
// fork's kernels with the paper's in-kernel row-time loop are in the un-vendored gsplat submodule.
GS_EXPORT int gs_rasterize_fwd_rs_slice(const float* records, const int* tile_bins, const int* band_edges,
                                        const float* background, int S, int H, int W, float* out_img, float* out_T,
                                        float* live_T, int* final_idx, unsigned char* tile_done, int first, int last,
                                        const int* sorted_ids, int n_records, float* out_depth, int* open_flag,
                                        const float* pix_vel, int N, float rolling_shutter_time,
                                        const float* shared_list_times, void* stream) {
  if (S <= 0 || H <= 0 || W <= 0 || N <= 0 || !pix_vel || !sorted_ids || n_records <= 0) return GS_ERR_INVALID;
  RasterParams prm = make_raster_params(records, sorted_ids, tile_bins, band_edges, background, S, 1, H, W);
  RsParams rs; rs.pix_vel = pix_vel; rs.N = N; rs.rs_time = rolling_shutter_time; rs.times = shared_list_times;
  RsSliceState st; st.tile_done = tile_done; st.live_T = live_T; st.first = first; st.last = last; st.open_flag = open_flag;
  const unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y), blocks = (work + 3) / 4;
  if (out_depth)
    hipLaunchKernelGGL(raster_fwd_rs_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, rs, st,
                       sorted_ids, records, (unsigned)(n_records - 1), out_img, out_T, final_idx, blocks, out_depth);
  else
    hipLaunchKernelGGL(raster_fwd_rs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, rs, st,
                       sorted_ids, records, (unsigned)(n_records - 1), out_img, out_T, final_idx, blocks, (float*)nullptr);
  return gs_launch_status();
}

// Backward of the same slice: tuples [I*12] (slots 0..8 as gs_rasterize_bwd_slice, 9..10 = d loss / d pixel velocity),
// flags [I] zeroed by the caller — shared_list_times != NULL: tuples [I*S*12], flags [I*S], entry e / sample s at e*S+s; sorted_vals = emission index of every sorted entry; bwd_T / bwd_B as in
// gs_rasterize_bwd_slice (both NULL for a one-slice frame); variant: + 256 = upstream alpha-clamp gradient.
GS_EXPORT int gs_rasterize_bwd_rs_slice(const float* records, const int* sorted_vals, const int* tile_bins,
                                        const int* band_edges, const float* background, int S, int H, int W,
                                        const float* out_T, const int* final_idx, const float* v_img,
                                        const float* v_alpha, float* bwd_T, float* bwd_B, float* tuples,
                                        unsigned char* flags, const int* sorted_ids, int n_records, int variant,
                                        const float* cmb_scale, float cmb_gamma, float cmb_min_level,
                                        const float* pix_vel, int N, float rolling_shutter_time,
                                        const float* shared_list_times, void* stream) {
  if (S <= 0 || H <= 0 || W <= 0 || N <= 0 || !pix_vel || !sorted_ids || !tuples || !flags || n_records <= 0)
    return GS_ERR_INVALID;
  if ((bwd_T == nullptr) != (bwd_B == nullptr)) return GS_ERR_INVALID;
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, 1, H, W);
  prm.cmb_scale = cmb_scale; prm.cmb_gamma = cmb_gamma; prm.cmb_min = cmb_min_level;
  if (variant & 256) prm.alpha_grad_max = 3.0e38f;
  RsParams rs; rs.pix_vel = pix_vel; rs.N = N; rs.rs_time = rolling_shutter_time; rs.times = shared_list_times;
  const unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y), blocks = (work + 3) / 4;
  if (bwd_T)
    hipLaunchKernelGGL(raster_bwd_rs_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, rs, sorted_ids,
                       sorted_vals, records, (unsigned)(n_records - 1), out_T, final_idx, v_img, v_alpha, blocks, bwd_T,
                       bwd_B, tuples, flags);
  else
    hipLaunchKernelGGL(raster_bwd_rs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, rs, sorted_ids,
                       sorted_vals, records, (unsigned)(n_records - 1), out_T, final_idx, v_img, v_alpha, blocks,
                       (float*)nullptr, (float*)nullptr, tuples, flags);
  return gs_launch_status();
}
