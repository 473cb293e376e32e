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
// Written for clarity over speed (scalar-cache record fetch like raster.hip, plain per-pixel arithmetic: dx is no
// longer a lane constant, so the packed / moment formulations of the other kernels do not apply).
#include "raster_common.h"

namespace gs {

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

struct RsRec { float x, y, qx, qy, qz, cx, cy, cz, op, r, g, b, d, pvx, pvy; };

template <bool DEPTH>
__device__ __forceinline__ RsRec load_rs_rec(const float* __restrict__ records, const float* __restrict__ pix_vel,
                                             unsigned gi, unsigned g) {
  const float kL2E = -1.4426950408889634f;
  const float* p = records + (size_t)gi * kRecFloats;
  RsRec o;
  o.x = p[0]; o.y = p[1]; o.cx = p[2]; o.cy = p[3]; o.cz = p[4]; o.op = p[5]; o.r = p[6]; o.g = p[7]; o.b = p[8];
  o.d = DEPTH ? p[9] : 0.f;
  o.qx = o.cx * (0.5f * kL2E); o.qy = o.cy * kL2E; o.qz = o.cz * (0.5f * kL2E);
  o.pvx = pix_vel[2 * (size_t)g]; o.pvy = pix_vel[2 * (size_t)g + 1];
  return o;
}

// -log2(e) * sigma at a pixel whose row is exposed tau after the sample time: the SAME expression in both kernels
__device__ __forceinline__ float rs_exponent(const RsRec& rc, float dx0, float dy0, float tau, float& dx, float& dy) {
  dx = dx0 + tau * rc.pvx;
  dy = dy0 + tau * rc.pvy;
  return dx * (rc.qx * dx + rc.qy * dy) + rc.qz * (dy * dy);
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
  if (!st.first && st.tile_done[tkey]) {
    // A finished (sample, tile) of a SHARED list: the tile's list still receives entries while any other sample's tile
    // is open (the binning sees the AND over the samples), so this slice's backward walks a non-empty range for this
    // sample too and reads this slice's stop indices — which are fresh, never memset arena memory.
    // Every pixel stopped before this slice: its stop index is the slice's first entry.
    if (shared) {
      const int qx = tx * K::kTile + (lane & 15), qy0 = ty * K::kTile + (lane >> 4) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (qx < prm.W && (qy0 + k) < prm.H) final_idx[((size_t)s * prm.H + (qy0 + k)) * prm.W + qx] = range.x;
    }
    return;
  }
  const float t_s = shared ? rs.times[s] : 0.f;
  if (!st.first && !st.last && range.y <= range.x) {
    if (st.open_flag && lane == 0) atomicAdd(st.open_flag, 1);
    return;
  }
  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  float Tk[4], Cr[4], Cg[4], Cb[4], Cd[4], pyf[4], tau[4];
  int last[4];
  bool inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    inside[k] = px < prm.W && (py0 + k) < prm.H;
    pyf[k] = (float)(py0 + k) + 0.5f;
    tau[k] = (pyf[k] / (float)prm.H - 0.5f) * rs.rs_time + t_s;
    Tk[k] = inside[k] ? 1.f : -1.f; Cr[k] = Cg[k] = Cb[k] = Cd[k] = 0.f; last[k] = range.x;
    if (!st.first && inside[k]) {
      const size_t pix = ((size_t)s * prm.H + (py0 + k)) * prm.W + px;
      Cr[k] = out_img[pix * 3 + 0]; Cg[k] = out_img[pix * 3 + 1]; Cb[k] = out_img[pix * 3 + 2];
      if (DEPTH) Cd[k] = out_depth[pix];
      const float Tf = out_T[pix], lv = st.live_T[pix];
      Tk[k] = lv > 0.f ? lv : -Tf;                         // a stopped pixel keeps its final T as a negative value
    }
  }
  auto any_live = [&]() { return __ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) != 0ull; };
  const unsigned s_base = shared ? 0u : (unsigned)s * (unsigned)rs.N;
  for (int i = range.x; i < range.y; ++i) {
    if (((i - range.x) & 3) == 0 && !any_live()) break;
    const unsigned gi = min((unsigned)ids[i], max_id);
    const RsRec rc = load_rs_rec<DEPTH>(records, rs.pix_vel, gi, gi - s_base);
    const float dx0 = rc.x - pxf;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float dx, dy;
      const float s2 = rs_exponent(rc, dx0, rc.y - pyf[k], tau[k], dx, dy);
      const float alpha = fminf(K::kAlphaMax, rc.op * __builtin_amdgcn_exp2f(s2));
      const bool v = (s2 <= 0.f) && (alpha >= K::kAlphaMin);
      const float w0 = alpha * Tk[k];
      const float nT = Tk[k] - Tk[k] * alpha;
      const bool u = v && (nT > K::kTMin);
      const float w = u ? w0 : 0.f;
      Cr[k] += w * rc.r; Cg[k] += w * rc.g; Cb[k] += w * rc.b;
      if (DEPTH) Cd[k] += w * rc.d;
      Tk[k] = u ? nT : (v ? -fabsf(Tk[k]) : Tk[k]);
      last[k] = u ? i + 1 : last[k];
    }
  }
  const bool all_stopped = !any_live();
  const bool finalize = all_stopped || st.last;
  const float bgr = finalize ? prm.background[0] : 0.f, bgg = finalize ? prm.background[1] : 0.f,
              bgb = finalize ? prm.background[2] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside[k]) {
      const float Tf = fabsf(Tk[k]);
      const size_t pix = ((size_t)s * prm.H + (py0 + k)) * prm.W + px;
      out_img[pix * 3 + 0] = Cr[k] + Tf * bgr;
      out_img[pix * 3 + 1] = Cg[k] + Tf * bgg;
      out_img[pix * 3 + 2] = Cb[k] + Tf * bgb;
      out_T[pix] = Tf;
      if (DEPTH) out_depth[pix] = Cd[k];
      final_idx[pix] = last[k];
      if (!st.last) st.live_T[pix] = fmaxf(Tk[k], 0.f);
    }
  }
  if (!st.last && lane == 0) {
    if (all_stopped) st.tile_done[tkey] = 1;
    else if (st.open_flag) atomicAdd(st.open_flag, 1);      // the word counts the tiles left open
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: reverse walk, 11 partial sums per (entry, lane) — x, y, conic (3), opacity, colour (3), pixel velocity
// (2) — transposed through wave-private LDS in groups of four entries, row totals straight into the entry's tuple
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRsComp = 11;
constexpr int kRsGroup = 4;
constexpr int kRsStride = 68;                        // 64 columns + 4: rows 16-byte aligned, b128 reads conflict-free
constexpr int kRsFloats = kRsGroup * kRsComp * kRsStride;
typedef float f4 __attribute__((ext_vector_type(4)));

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
  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];
  float Tk[4], Dv[4], vr[4], vg[4], vb[4], pyf[4], tau[4];
  int fin[4];
  int my_end = range.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = py0 + k;
    pyf[k] = (float)y + 0.5f;
    tau[k] = (pyf[k] / (float)prm.H - 0.5f) * rs.rs_time + t_s;
    Tk[k] = 1.f; Dv[k] = 0.f; fin[k] = range.x; vr[k] = vg[k] = vb[k] = 0.f;
    if (px < prm.W && y < prm.H) {
      const size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      const float Tfin = out_T[pix];
      fin[k] = min(max(final_idx[pix], range.x), range.y);   // a stop index never leaves the tile's list
      vr[k] = v_img[pix * 3 + 0]; vg[k] = v_img[pix * 3 + 1]; vb[k] = v_img[pix * 3 + 2];
      if (prm.cmb_scale) {
        const size_t q = ((size_t)y * prm.W + px) * 3;
        vr[k] = combine_grad(vr[k], prm.cmb_scale[q + 0], prm.cmb_gamma, prm.cmb_min);
        vg[k] = combine_grad(vg[k], prm.cmb_scale[q + 1], prm.cmb_gamma, prm.cmb_min);
        vb[k] = combine_grad(vb[k], prm.cmb_scale[q + 2], prm.cmb_gamma, prm.cmb_min);
      }
      const float va_out = v_alpha ? v_alpha[pix] : 0.f;
      const float va = Tfin * (va_out - (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]));
      Tk[k] = Tfin;
      Dv[k] = -va;
      if (STATE) { Tk[k] = bwd_T[pix]; Dv[k] = bwd_B[pix] - va; }
    }
    my_end = max(my_end, fin[k]);
  }
  const int wave_end = __builtin_amdgcn_readfirstlane(wave_max_i(my_end));
  const float agm = prm.alpha_grad_max;
  const unsigned s_base = shared ? 0u : (unsigned)s * (unsigned)rs.N;
  // shared list: the S samples' waves write the SAME entry's gradients — every (entry, sample) pair gets a tuple of its
  // own, e * S + s (the S tuples of an entry are adjacent, a Gaussian's tuples stay one contiguous range)
  const unsigned tmul = shared ? (unsigned)prm.S : 1u, tofs = shared ? (unsigned)s : 0u;
  const int row = lane;
  const int row_g = row / kRsComp, row_c = row - row_g * kRsComp;
  if (wave_end > range.x) {
    for (int b = (wave_end - 1) & ~3; b >= (range.x & ~3); b -= 4) {
      unsigned filled = 0;
#pragma unroll
      for (int slot = 3; slot >= 0; --slot) {
        const int i = b + slot;
        if (i < range.x || i >= wave_end) continue;
        const unsigned gi = min((unsigned)ids[i], max_id);
        const RsRec rc = load_rs_rec<false>(records, rs.pix_vel, gi, gi - s_base);
        const float dx0 = rc.x - pxf;
        float dxk[4], dyk[4], vis[4], ov[4];
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float s2 = rs_exponent(rc, dx0, rc.y - pyf[k], tau[k], dxk[k], dyk[k]);
          vis[k] = __builtin_amdgcn_exp2f(s2);
          ov[k] = rc.op * vis[k];
          hit[k] = (i < fin[k]) && (s2 <= 0.f) && (fminf(K::kAlphaMax, ov[k]) >= K::kAlphaMin);
        }
        if (__ballot(hit[0] || hit[1] || hit[2] || hit[3]) == 0ull) continue;
        float p[kRsComp];
#pragma unroll
        for (int c = 0; c < kRsComp; ++c) p[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float alpha = hit[k] ? fminf(K::kAlphaMax, ov[k]) : 0.f;      // alpha = 0: every term below is an exact zero
          const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
          Tk[k] *= ra;                                                         // transmittance in front of this entry
          const float fac = alpha * Tk[k];
          p[6] += fac * vr[k]; p[7] += fac * vg[k]; p[8] += fac * vb[k];
          const float cv = rc.r * vr[k] + rc.g * vg[k] + rc.b * vb[k];
          const float v_al = Tk[k] * cv - ra * Dv[k];
          Dv[k] += fac * cv;
          const bool free_ = hit[k] && ov[k] <= agm;                           // d min(0.999, o*vis) = 0 when clamped
          const float vism = free_ ? vis[k] : 0.f;
          const float v_sigma = -(rc.op * vism) * v_al;
          p[5] += vism * v_al;
          const float gdx = v_sigma * (rc.cx * dxk[k] + rc.cy * dyk[k]);       // d sigma / d dx
          const float gdy = v_sigma * (rc.cy * dxk[k] + rc.cz * dyk[k]);
          p[0] += gdx; p[1] += gdy;
          p[9] += tau[k] * gdx; p[10] += tau[k] * gdy;                          // centre = mu' + (t_s + tau) v'
          p[2] += 0.5f * v_sigma * dxk[k] * dxk[k];
          p[3] += v_sigma * dxk[k] * dyk[k];
          p[4] += 0.5f * v_sigma * dyk[k] * dyk[k];
        }
        filled |= 1u << slot;
        float* r0 = red + slot * (kRsComp * kRsStride) + lane;
#pragma unroll
        for (int c = 0; c < kRsComp; ++c) r0[c * kRsStride] = p[c];
      }
      if (filled) {
        __builtin_amdgcn_wave_barrier();
        if (row < kRsGroup * kRsComp && ((filled >> row_g) & 1u)) {
          const f4* rp = reinterpret_cast<const f4*>(red + row * kRsStride);
          f4 a0 = rp[0], a1 = rp[1], a2 = rp[2], a3 = rp[3];
#pragma unroll
          for (int q = 4; q < 16; q += 4) { a0 += rp[q]; a1 += rp[q + 1]; a2 += rp[q + 2]; a3 += rp[q + 3]; }
          const f4 v = (a0 + a1) + (a2 + a3);
          const float sum = (v.x + v.y) + (v.z + v.w);
          const size_t e = (size_t)(unsigned)eids[b + row_g] * tmul + tofs;
          tuples[e * kGradFloats + row_c] = sum;
          if (row_c == 0) flags[e] = 1;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (STATE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = py0 + k;
      if (px < prm.W && y < prm.H) {
        const size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
        const float Tfin = out_T[pix];
        const float va_out = v_alpha ? v_alpha[pix] : 0.f;
        const float va = Tfin * (va_out - (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]));
        bwd_T[pix] = Tk[k];
        bwd_B[pix] = Dv[k] + va;
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
