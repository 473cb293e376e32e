// raster.hip — per-pixel front-to-back alpha compositing (forward) and its
// reverse-order backward, written for gfx950 / wave64.
//
// Restates (absent fork sources) gsplat's rasterize_forward /
// rasterize_backward_kernel as recollected in SURVEY.md App. A "Blend" and
// "Backward"; constants in gs::K (gs_math.h).
//
// MI355X design (not the CUDA 256-thread-block/shared-memory tiling):
//   * one wave64 owns one 16x16 tile; each lane owns a 1x4 pixel column segment
//     (x = lane&15, y = 4*(lane>>4)+k).  No LDS, no __syncthreads: the 64
//     Gaussians of a batch live one-per-lane in VGPRs and are broadcast with
//     v_readlane_b32 into SGPRs, so the per-pair math reads uniform operands
//     from the scalar file.
//   * dx is shared by a lane's 4 pixels.
//   * early termination is a wave ballot.
//   * backward: a lane pre-sums its 4 pixels, one 6-step DPP wave reduction per
//     gradient component per (Gaussian, tile), the total is parked in the lane
//     that owns the Gaussian, and every lane issues its own 9 fp32 atomics after
//     the batch (64 distinct addresses per instruction, no same-address storms).
#include "gs_common.h"

namespace gs {

struct RasterParams {
  const float* records;      // [P*N, 12]
  const int*   sorted_vals;  // [I]   p*N+g sorted by (p*T+tile, depth) -- or, when gi_of_e != null, the
                             //       EMISSION index e of each sorted entry (p*N+g = gi_of_e[e])
  const int*   gi_of_e;      // [I]   nullable
  const int2*  tile_bins;    // [P*T]
  const int*   band_edges;   // [R+1] tile-row edges of the rolling-shutter bands
  const float* background;   // [3]
  int S, R, H, W, tiles_x, tiles_y;
};

__device__ __forceinline__ int find_band(const int* __restrict__ edges, int R, int ty) {
  int r = 0;
  while (r + 1 < R && ty >= edges[r + 1]) ++r;
  return r;
}

struct Rec9 { float x, y, cx, cy, cz, op, r, g, b; };

__device__ __forceinline__ Rec9 load_rec(const float* __restrict__ records, int gid, bool valid) {
  Rec9 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const float4* p = reinterpret_cast<const float4*>(records + (size_t)gid * kRecFloats);
    float4 a = p[0], b = p[1], c = p[2];
    o.x = a.x; o.y = a.y; o.cx = a.z; o.cy = a.w;
    o.cz = b.x; o.op = b.y; o.r = b.z; o.g = b.w;
    o.b = c.x;
  }
  return o;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void raster_fwd_kernel(RasterParams prm, float* __restrict__ out_img,
                                                         float* __restrict__ out_T, int* __restrict__ final_idx,
                                                         unsigned n_blocks) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6);
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  const int2 range = prm.tile_bins[(size_t)p * T + t];

  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  float Tk[4], Cr[4], Cg[4], Cb[4];
  int last[4];
  bool done[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Tk[k] = 1.f; Cr[k] = Cg[k] = Cb[k] = 0.f; last[k] = range.x;
    done[k] = !(px < prm.W && (py0 + k) < prm.H);
  }

  const int* __restrict__ vals = prm.sorted_vals;
  // software pipeline: ids two batches ahead, records one batch ahead
  int id_next = 0;
  {
    int i0 = range.x + lane;
    id_next = i0 < range.y ? vals[i0] : 0;
  }
  Rec9 rec_next = load_rec(prm.records, id_next, (range.x + lane) < range.y);
  {
    int i1 = range.x + 64 + lane;
    id_next = i1 < range.y ? vals[i1] : 0;
  }

  for (int batch = range.x; batch < range.y; batch += 64) {
    if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
    Rec9 rec = rec_next;
    // prefetch the following batch
    rec_next = load_rec(prm.records, id_next, (batch + 64 + lane) < range.y);
    {
      int i2 = batch + 128 + lane;
      id_next = i2 < range.y ? vals[i2] : 0;
    }
    const int n = min(64, range.y - batch);
    for (int j = 0; j < n; ++j) {
      if ((j & 7) == 0 && j && __ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
      const float gx = readlane_f(rec.x, j), gy = readlane_f(rec.y, j);
      const float cx = readlane_f(rec.cx, j), cy = readlane_f(rec.cy, j), cz = readlane_f(rec.cz, j);
      const float op = readlane_f(rec.op, j);
      const float cr = readlane_f(rec.r, j), cg = readlane_f(rec.g, j), cb = readlane_f(rec.b, j);
      const float dx = gx - pxf;
      const float hx = 0.5f * cx * dx * dx;   // shared by the lane's 4 pixels
      const float bx = cy * dx;
      const float hz = 0.5f * cz;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!done[k]) {
          const float dy = gy - ((float)(py0 + k) + 0.5f);
          const float sigma = hx + dy * (bx + hz * dy);
          if (sigma >= 0.f) {
            const float alpha = fminf(K::kAlphaMax, op * __expf(-sigma));
            if (alpha >= K::kAlphaMin) {
              const float nT = Tk[k] * (1.f - alpha);
              if (nT <= K::kTMin) {
                done[k] = true;
              } else {
                const float w = alpha * Tk[k];
                Cr[k] += w * cr; Cg[k] += w * cg; Cb[k] += w * cb;
                Tk[k] = nT;
                last[k] = batch + j + 1;
              }
            }
          }
        }
      }
    }
  }
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = py0 + k;
    if (px < prm.W && y < prm.H) {
      size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      out_img[pix * 3 + 0] = Cr[k] + Tk[k] * bgr;
      out_img[pix * 3 + 1] = Cg[k] + Tk[k] * bgg;
      out_img[pix * 3 + 2] = Cb[k] + Tk[k] * bgb;
      out_T[pix] = Tk[k];
      final_idx[pix] = last[k];
    }
  }
}

// ---------------------------------------------------------------------------
// forward, variant 2: branch-free inner loop.  "Stopped" is encoded as T = 0 (so a stopped
// pixel can never pass nT > 1e-4 again), conics are pre-scaled by -log2(e) so the exponent feeds
// v_exp_f32 directly, and every update is a select: ~20 VALU per (pixel, Gaussian) with no
// exec-mask juggling.  Bit-for-bit the same decisions as variant 1 except for the exp argument
// rounding (inside the stated fp32 tolerance).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void raster_fwd_kernel_v2(RasterParams prm, float* __restrict__ out_img,
                                                            float* __restrict__ out_T, int* __restrict__ final_idx,
                                                            unsigned n_blocks) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6);
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  const int2 range = prm.tile_bins[(size_t)p * T + t];

  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  float Tk[4], Tf[4], Cr[4], Cg[4], Cb[4], pyf[4];
  int last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool inside = px < prm.W && (py0 + k) < prm.H;
    Tk[k] = inside ? 1.f : 0.f; Tf[k] = 1.f; Cr[k] = Cg[k] = Cb[k] = 0.f; last[k] = range.x;
    pyf[k] = (float)(py0 + k) + 0.5f;
  }
  const int* __restrict__ vals = prm.sorted_vals;
  int id_next = (range.x + lane) < range.y ? vals[range.x + lane] : 0;
  Rec9 rec_next = load_rec(prm.records, id_next, (range.x + lane) < range.y);
  id_next = (range.x + 64 + lane) < range.y ? vals[range.x + 64 + lane] : 0;
  const float kL2E = -1.4426950408889634f;

  for (int batch = range.x; batch < range.y; batch += 64) {
    if (__ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull) break;
    Rec9 rec = rec_next;
    rec_next = load_rec(prm.records, id_next, (batch + 64 + lane) < range.y);
    id_next = (batch + 128 + lane) < range.y ? vals[batch + 128 + lane] : 0;
    // pre-scale: sigma2 = -log2e * sigma = hx2 + dy*(bx2 + hz2*dy)
    rec.cx *= 0.5f * kL2E; rec.cy *= kL2E; rec.cz *= 0.5f * kL2E;
    const int n = min(64, range.y - batch);
    for (int j = 0; j < n; ++j) {
      if ((j & 15) == 15 && __ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull) break;
      const float gx = readlane_f(rec.x, j), gy = readlane_f(rec.y, j);
      const float cx = readlane_f(rec.cx, j), cy = readlane_f(rec.cy, j), cz = readlane_f(rec.cz, j);
      const float op = readlane_f(rec.op, j);
      const float cr = readlane_f(rec.r, j), cg = readlane_f(rec.g, j), cb = readlane_f(rec.b, j);
      const float dx = gx - pxf;
      const float hx = cx * dx * dx;
      const float bx = cy * dx;
      const int idx1 = batch + j + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = gy - pyf[k];
        const float s2 = hx + dy * (bx + cz * dy);                // = -log2e * sigma
        const float alpha = fminf(K::kAlphaMax, op * __builtin_amdgcn_exp2f(s2));
        const bool valid = (s2 <= 0.f) && (alpha >= K::kAlphaMin);
        const float nT = Tk[k] - Tk[k] * alpha;
        const bool upd = valid && (nT > K::kTMin);
        const float w = upd ? alpha * Tk[k] : 0.f;
        Cr[k] += w * cr; Cg[k] += w * cg; Cb[k] += w * cb;
        Tf[k] = upd ? nT : Tf[k];
        Tk[k] = upd ? nT : (valid ? 0.f : Tk[k]);
        last[k] = upd ? idx1 : last[k];
      }
    }
  }
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = py0 + k;
    if (px < prm.W && y < prm.H) {
      size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      out_img[pix * 3 + 0] = Cr[k] + Tf[k] * bgr;
      out_img[pix * 3 + 1] = Cg[k] + Tf[k] * bgg;
      out_img[pix * 3 + 2] = Cb[k] + Tf[k] * bgb;
      out_T[pix] = Tf[k];
      final_idx[pix] = last[k];
    }
  }
}

// ---------------------------------------------------------------------------
// forward, depth-sliced: same inner loop as variant 2, but the per-pixel state (colour without
// background, final T, live T) persists in HBM between slices and a tile whose pixels have all
// stopped is flagged `done` (it gets its background term then, is skipped by later slices and
// receives no further intersections from the binning).  first && last reproduces the unsliced pass.
// ---------------------------------------------------------------------------
struct SliceState {
  unsigned char* tile_done;   // [P*T]
  float* live_T;              // [S,H,W]  0 once a pixel has stopped
  int first, last;
};

template <bool SKIP_EMPTY>
__global__ __launch_bounds__(256) void raster_fwd_slice_kernel(RasterParams prm, SliceState st,
                                                               float* __restrict__ out_img,
                                                               float* __restrict__ out_T,
                                                               int* __restrict__ final_idx, unsigned n_blocks) {
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
  if (!st.first && !st.last && range.y <= range.x) return;   // nothing for this tile in this slice

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

  for (int batch = range.x; batch < range.y; batch += 64) {
    if (__ballot(fmaxf(fmaxf(Tk[0], Tk[1]), fmaxf(Tk[2], Tk[3])) > 0.f) == 0ull) break;
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
  if (all_stopped && !st.last && lane == 0) st.tile_done[tkey] = 1;
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// STATE = true: depth-sliced backward — the running transmittance and the colour accumulated from
// behind persist in bwd_T / bwd_B between slice launches (slices are visited back to front).
template <bool STATE>
__global__ __launch_bounds__(256) void raster_bwd_kernel(RasterParams prm, const float* __restrict__ out_T,
                                                         const int* __restrict__ final_idx,
                                                         const float* __restrict__ v_img,
                                                         const float* __restrict__ v_alpha,  // may be null
                                                         float* __restrict__ v_records, unsigned n_blocks,
                                                         float* __restrict__ bwd_T, float* __restrict__ bwd_B) {
  const int lane = lane_id();
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6);
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  const int2 range = prm.tile_bins[(size_t)p * T + t];
  if (range.y <= range.x) return;

  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];

  float Tk[4], Tfin[4], Br[4], Bg[4], Bb[4], vr[4], vg[4], vb[4], va[4];
  int fin[4];
  int my_end = range.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = py0 + k;
    Br[k] = Bg[k] = Bb[k] = 0.f;
    if (px < prm.W && y < prm.H) {
      size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      Tfin[k] = out_T[pix];
      fin[k] = final_idx[pix];
      vr[k] = v_img[pix * 3 + 0]; vg[k] = v_img[pix * 3 + 1]; vb[k] = v_img[pix * 3 + 2];
      float va_out = v_alpha ? v_alpha[pix] : 0.f;
      // d(out)/d(alpha_i) carries T_final/(1-alpha_i) * (v_alpha_out - sum_c bg_c v_c)
      va[k] = Tfin[k] * (va_out - (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]));
    } else {
      Tfin[k] = 1.f; fin[k] = range.x; vr[k] = vg[k] = vb[k] = 0.f; va[k] = 0.f;
    }
    Tk[k] = Tfin[k];
    if (STATE && px < prm.W && y < prm.H) {
      size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      Tk[k] = bwd_T[pix];
      Br[k] = bwd_B[pix * 3 + 0]; Bg[k] = bwd_B[pix * 3 + 1]; Bb[k] = bwd_B[pix * 3 + 2];
    }
    my_end = max(my_end, fin[k]);
  }
  const int wave_end = wave_max_i(my_end);
  const int* __restrict__ vals = prm.sorted_vals;

  for (int batch_end = wave_end; batch_end > range.x; batch_end -= 64) {
    const int idx = batch_end - 1 - lane;
    const bool valid = idx >= range.x;
    const int gid = valid ? vals[idx] : 0;
    const Rec9 rec = load_rec(prm.records, gid, valid);
    float a_x = 0.f, a_y = 0.f, a_cx = 0.f, a_cy = 0.f, a_cz = 0.f, a_op = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f;
    const int n = min(64, batch_end - range.x);
    for (int j = 0; j < n; ++j) {
      const int idx_j = batch_end - 1 - j;
      const float gx = readlane_f(rec.x, j), gy = readlane_f(rec.y, j);
      const float cx = readlane_f(rec.cx, j), cy = readlane_f(rec.cy, j), cz = readlane_f(rec.cz, j);
      const float op = readlane_f(rec.op, j);
      const float cr = readlane_f(rec.r, j), cg = readlane_f(rec.g, j), cb = readlane_f(rec.b, j);
      const float dx = gx - pxf;
      const float hx = 0.5f * cx * dx * dx;
      const float bx = cy * dx;
      const float hz = 0.5f * cz;
      float p_x = 0.f, p_y = 0.f, p_cx = 0.f, p_cy = 0.f, p_cz = 0.f, p_op = 0.f, p_r = 0.f, p_g = 0.f, p_b = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (idx_j < fin[k]) {
          const float dy = gy - ((float)(py0 + k) + 0.5f);
          const float sigma = hx + dy * (bx + hz * dy);
          if (sigma >= 0.f) {
            const float vis = __expf(-sigma);
            const float ov = op * vis;
            const float alpha = fminf(K::kAlphaMax, ov);
            if (alpha >= K::kAlphaMin) {
              any = true;
              const float ra = 1.f / (1.f - alpha);
              Tk[k] *= ra;                       // transmittance in front of this Gaussian
              const float fac = alpha * Tk[k];
              p_r += fac * vr[k]; p_g += fac * vg[k]; p_b += fac * vb[k];
              float v_al = (cr * Tk[k] - Br[k] * ra) * vr[k] + (cg * Tk[k] - Bg[k] * ra) * vg[k] +
                           (cb * Tk[k] - Bb[k] * ra) * vb[k] + va[k] * ra;
              Br[k] += cr * fac; Bg[k] += cg * fac; Bb[k] += cb * fac;
              if (ov <= K::kAlphaMax) {          // d min(0.999, o*vis) = 0 when clamped
                const float v_sigma = -ov * v_al;
                p_op += vis * v_al;
                p_cx += 0.5f * v_sigma * dx * dx;
                p_cy += v_sigma * dx * dy;
                p_cz += 0.5f * v_sigma * dy * dy;
                p_x += v_sigma * (cx * dx + cy * dy);
                p_y += v_sigma * (cy * dx + cz * dy);
              }
            }
          }
        }
      }
      if (__ballot(any) == 0ull) continue;
      const float t_x = wave_sum_uniform(p_x), t_y = wave_sum_uniform(p_y);
      const float t_cx = wave_sum_uniform(p_cx), t_cy = wave_sum_uniform(p_cy), t_cz = wave_sum_uniform(p_cz);
      const float t_op = wave_sum_uniform(p_op);
      const float t_r = wave_sum_uniform(p_r), t_g = wave_sum_uniform(p_g), t_b = wave_sum_uniform(p_b);
      if (lane == j) {
        a_x = t_x; a_y = t_y; a_cx = t_cx; a_cy = t_cy; a_cz = t_cz; a_op = t_op; a_r = t_r; a_g = t_g; a_b = t_b;
      }
    }
    if (valid) {
      float* dst = v_records + (size_t)gid * kRecFloats;
      if (a_x != 0.f) atomic_add_f32(dst + 0, a_x);
      if (a_y != 0.f) atomic_add_f32(dst + 1, a_y);
      if (a_cx != 0.f) atomic_add_f32(dst + 2, a_cx);
      if (a_cy != 0.f) atomic_add_f32(dst + 3, a_cy);
      if (a_cz != 0.f) atomic_add_f32(dst + 4, a_cz);
      if (a_op != 0.f) atomic_add_f32(dst + 5, a_op);
      if (a_r != 0.f) atomic_add_f32(dst + 6, a_r);
      if (a_g != 0.f) atomic_add_f32(dst + 7, a_g);
      if (a_b != 0.f) atomic_add_f32(dst + 8, a_b);
    }
  }
  if (STATE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = py0 + k;
      if (px < prm.W && y < prm.H) {
        size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
        bwd_T[pix] = Tk[k];
        bwd_B[pix * 3 + 0] = Br[k]; bwd_B[pix * 3 + 1] = Bg[k]; bwd_B[pix * 3 + 2] = Bb[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward, variant 2 (default for the depth-sliced path).  Same math as raster_bwd_kernel, cheaper
// instruction stream:
//   * one predicate per pixel instead of three nested exec-mask regions, exp2 on a pre-scaled
//     exponent, v_rcp_f32 for 1/(1-alpha) (the IEEE division expansion cost ~10 VALU per pixel);
//   * the 9 per-Gaussian wave reductions (54 DPP adds + 18 lane moves) are replaced by a transposed
//     reduction through wave-private LDS: every lane drops its 9 partials into row (g*9+c) of a
//     [36][68] tile (conflict-free ds_write_b32), after 4 Gaussians lanes 0..35 each sum one row
//     with 16 conflict-free ds_read_b128 and park the total in tot[j][c]; at the end of the batch
//     lane j picks up its 9 totals.  ~20 issue slots per Gaussian instead of ~80.
// ---------------------------------------------------------------------------
#ifndef GS_RED_G
#define GS_RED_G 3
#endif
constexpr int kRedG = GS_RED_G;             // Gaussians per transposed-reduction group
constexpr int kRedStride = 68;              // floats per row (64 + 4: 16-byte aligned, b128 conflict-free)
constexpr int kRedFloats = kRedG * 9 * kRedStride + 64 * 9;

// OUT = 0: 9 fp32 atomics per (Gaussian, tile) into v_records;  OUT = 2: timing ablation (plain stores);
// OUT = 1: no atomics at all — the entry's 9 gradients go to tuples[e] (48 B, e = emission index of the
// entry, so the tuples of one Gaussian are CONTIGUOUS) and flags[e] = 1; gs_reduce_grad_tuples then sums
// each Gaussian's segment.  At ~20 G atomic ops/s the atomics were 40 % of this kernel.
#ifndef GS_BWD_WAVES
#define GS_BWD_WAVES 4   // 128 VGPRs + 38 KB LDS per block -> 4 waves per SIMD (+3.5 % measured)
#endif
template <bool STATE, int OUT>
__global__ __launch_bounds__(256, GS_BWD_WAVES) void raster_bwd_kernel_v2(RasterParams prm, const float* __restrict__ out_T,
                                                            const int* __restrict__ final_idx,
                                                            const float* __restrict__ v_img,
                                                            const float* __restrict__ v_alpha,  // may be null
                                                            float* __restrict__ v_records, unsigned n_blocks,
                                                            float* __restrict__ bwd_T, float* __restrict__ bwd_B,
                                                            float* __restrict__ tuples,
                                                            unsigned char* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) float lds_all[4 * kRedFloats];
  const int lane = lane_id();
  float* red = lds_all + (threadIdx.x >> 6) * kRedFloats;   // wave-private
  float* tot = red + kRedG * 9 * kRedStride;
  const int T = prm.tiles_x * prm.tiles_y;
  const unsigned work = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(xcd_remap(blockIdx.x, n_blocks) * 4u + (threadIdx.x >> 6)));
  if (work >= (unsigned)(prm.S * T)) return;
  const int s = work / T, t = work % T;
  const int ty = t / prm.tiles_x, tx = t % prm.tiles_x;
  const int p = s * prm.R + find_band(prm.band_edges, prm.R, ty);
  int2 range = prm.tile_bins[(size_t)p * T + t];
  range.x = __builtin_amdgcn_readfirstlane(range.x);
  range.y = __builtin_amdgcn_readfirstlane(range.y);
  if (range.y <= range.x) return;

  const int px = tx * K::kTile + (lane & 15);
  const int py0 = ty * K::kTile + (lane >> 4) * 4;
  const float pxf = (float)px + 0.5f;
  const float bgr = prm.background[0], bgg = prm.background[1], bgb = prm.background[2];

  // per pixel: Tk = transmittance behind the current Gaussian, Dv = (colour accumulated from behind,
  // dotted with v_out) - va, where va = T_final * (v_alpha_out - bg . v_out).  Only the DOT of the
  // behind-colour with v_out is ever needed, so one float replaces the three colour channels.
  float Tk[4], Dv[4], vr[4], vg[4], vb[4], pyf[4];
  int fin[4];
  int my_end = range.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = py0 + k;
    pyf[k] = (float)y + 0.5f;
    Tk[k] = 1.f; Dv[k] = 0.f; fin[k] = range.x; vr[k] = vg[k] = vb[k] = 0.f;
    if (px < prm.W && y < prm.H) {
      size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
      const float Tfin = out_T[pix];
      fin[k] = final_idx[pix];
      vr[k] = v_img[pix * 3 + 0]; vg[k] = v_img[pix * 3 + 1]; vb[k] = v_img[pix * 3 + 2];
      const float va_out = v_alpha ? v_alpha[pix] : 0.f;
      const float va = Tfin * (va_out - (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]));
      Tk[k] = Tfin;
      Dv[k] = -va;
      if (STATE) {
        Tk[k] = bwd_T[pix];
        Dv[k] = bwd_B[pix] - va;
      }
    }
    my_end = max(my_end, fin[k]);
  }
  const int wave_end = __builtin_amdgcn_readfirstlane(wave_max_i(my_end));
  const int* __restrict__ vals = prm.sorted_vals;
  const float kL2E = -1.4426950408889634f;
  const int row = lane;                        // row-sum role: lanes 0..35
  const int row_g = row / 9, row_c = row - row_g * 9;

  for (int batch_end = wave_end; batch_end > range.x; batch_end -= 64) {
    const int idx = batch_end - 1 - lane;
    const bool valid = idx >= range.x;
    const int eid = valid ? vals[idx] : 0;
    const int gid = (valid && prm.gi_of_e) ? prm.gi_of_e[eid] : eid;
    const Rec9 rec = load_rec(prm.records, gid, valid);
    const float sx = rec.cx * (0.5f * kL2E), sy = rec.cy * kL2E, sz = rec.cz * (0.5f * kL2E);
#pragma unroll
    for (int c = 0; c < 9; ++c) tot[lane * 9 + c] = 0.f;
    const int n = min(64, batch_end - range.x);
    unsigned filled = 0;                          // which slots of the current group hold data
    int g = 0, gbase = 0;                         // slot inside the group, batch position of its first Gaussian
    for (int j = 0; j < n; ++j, ++g) {
      const int idx_j = batch_end - 1 - j;
      const float gx = readlane_f(rec.x, j), gy = readlane_f(rec.y, j);
      const float qx = readlane_f(sx, j), qy = readlane_f(sy, j), qz = readlane_f(sz, j);
      const float op = readlane_f(rec.op, j);
      const float dx = gx - pxf;
      const float hx = qx * dx * dx;             // exponent terms, pre-scaled by -log2(e)
      const float bx = qy * dx;
      float vis[4], ov[4];
      bool hit[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = gy - pyf[k];
        const float s2 = hx + dy * (bx + qz * dy);
        vis[k] = __builtin_amdgcn_exp2f(s2);
        ov[k] = op * vis[k];
        hit[k] = (idx_j < fin[k]) && (s2 <= 0.f) && (fminf(K::kAlphaMax, ov[k]) >= K::kAlphaMin);
      }
      if (__ballot(hit[0] || hit[1] || hit[2] || hit[3]) != 0ull) {
        const float cx = readlane_f(rec.cx, j), cy = readlane_f(rec.cy, j), cz = readlane_f(rec.cz, j);
        const float cr = readlane_f(rec.r, j), cg = readlane_f(rec.g, j), cb = readlane_f(rec.b, j);
        const float hdx2 = 0.5f * dx * dx;
        const float cxdx = cx * dx, cydx = cy * dx;
        float p_x = 0.f, p_y = 0.f, p_cx = 0.f, p_cy = 0.f, p_cz = 0.f, p_op = 0.f, p_r = 0.f, p_g = 0.f, p_b = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (hit[k]) {
            const float dy = gy - pyf[k];
            const float alpha = fminf(K::kAlphaMax, ov[k]);
            const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
            Tk[k] *= ra;                       // transmittance in front of this Gaussian
            const float fac = alpha * Tk[k];
            p_r += fac * vr[k]; p_g += fac * vg[k]; p_b += fac * vb[k];
            const float cv = cr * vr[k] + cg * vg[k] + cb * vb[k];
            const float v_al = Tk[k] * cv - ra * Dv[k];
            Dv[k] += fac * cv;
            const bool free_ = ov[k] <= K::kAlphaMax;     // d min(0.999, o*vis) = 0 when clamped
            const float v_sigma = free_ ? -ov[k] * v_al : 0.f;
            p_op += free_ ? vis[k] * v_al : 0.f;
            const float vsdy = v_sigma * dy;
            p_cx += v_sigma * hdx2;
            p_cy += vsdy * dx;
            p_cz += vsdy * (0.5f * dy);
            p_x += v_sigma * (cxdx + cy * dy);
            p_y += v_sigma * (cydx + cz * dy);
          }
        }
        filled |= 1u << g;
        float* r0 = red + g * (9 * kRedStride) + lane;
        r0[0 * kRedStride] = p_x;  r0[1 * kRedStride] = p_y;  r0[2 * kRedStride] = p_cx;
        r0[3 * kRedStride] = p_cy; r0[4 * kRedStride] = p_cz; r0[5 * kRedStride] = p_op;
        r0[6 * kRedStride] = p_r;  r0[7 * kRedStride] = p_g;  r0[8 * kRedStride] = p_b;
      }
      if (g == kRedG - 1 || j == n - 1) {
        if (filled) {
          __builtin_amdgcn_wave_barrier();
          if (row < kRedG * 9 && ((filled >> row_g) & 1u)) {
            const float4* rp = reinterpret_cast<const float4*>(red + row * kRedStride);
            float4 a0 = rp[0], a1 = rp[1], a2 = rp[2], a3 = rp[3];
#pragma unroll
            for (int q = 4; q < 16; q += 4) {
              float4 b0 = rp[q], b1 = rp[q + 1], b2 = rp[q + 2], b3 = rp[q + 3];
              a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
              a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
              a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
              a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
            }
            const float sum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) +
                              ((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w));
            const int jj = gbase + row_g;                   // batch position of this row's Gaussian
            tot[jj * 9 + row_c] = sum;
          }
          __builtin_amdgcn_wave_barrier();
          filled = 0;
        }
        g = -1;
        gbase = j + 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      float a[9];
      bool nz = false;
#pragma unroll
      for (int c = 0; c < 9; ++c) { a[c] = tot[lane * 9 + c]; nz |= a[c] != 0.f; }
      if (OUT == 1) {
        if (nz) {
          float4* dst = reinterpret_cast<float4*>(tuples + (size_t)eid * kRecFloats);
          dst[0] = make_float4(a[0], a[1], a[2], a[3]);
          dst[1] = make_float4(a[4], a[5], a[6], a[7]);
          dst[2] = make_float4(a[8], 0.f, 0.f, 0.f);
          flags[eid] = 1;
        }
      } else {
        float* dst = v_records + (size_t)gid * kRecFloats;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          if (OUT == 2) {                  // timing experiment only (wrong gradients): plain stores
            if (a[c] != 0.f) dst[c] = a[c];
          } else if (a[c] != 0.f) {
            atomic_add_f32(dst + c, a[c]);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (STATE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = py0 + k;
      if (px < prm.W && y < prm.H) {
        size_t pix = ((size_t)s * prm.H + y) * prm.W + px;
        const float Tfin = out_T[pix];
        const float va_out = v_alpha ? v_alpha[pix] : 0.f;
        const float va = Tfin * (va_out - (bgr * vr[k] + bgg * vg[k] + bgb * vb[k]));
        bwd_T[pix] = Tk[k];
        bwd_B[pix] = Dv[k] + va;       // behind-colour . v_out
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Segmented sum of the gradient tuples of one depth slice.  The tuples of slice Gaussian j occupy
// [cum[j], cum[j]+counts[j]) (emission order); flags mark the entries the backward actually touched.
// A wave owns 64 Gaussians: short segments are summed by their own lane, long ones (near Gaussians
// cover hundreds of tiles) by the whole wave with one DPP reduction per component.  Every Gaussian
// belongs to exactly one slice, so the result is a plain store into v_records — no atomics anywhere.
// ---------------------------------------------------------------------------
constexpr unsigned kReduceSolo = 16;

__global__ __launch_bounds__(256) void reduce_tuples_kernel(int n_slice, const unsigned* __restrict__ slice_gi,
                                                            const unsigned* __restrict__ counts,
                                                            const unsigned* __restrict__ cum,
                                                            const float* __restrict__ tuples,
                                                            const unsigned char* __restrict__ flags,
                                                            float* __restrict__ v_records,
                                                            unsigned char* __restrict__ touched) {
  const int lane = lane_id();
  const int j = blockIdx.x * 256 + threadIdx.x;
  unsigned cnt = 0, e0 = 0, gi = 0;
  if (j < n_slice) { cnt = counts[j]; e0 = cum[j]; gi = slice_gi[j]; }
  float acc[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) acc[c] = 0.f;
  bool any = false;
  if (cnt && cnt <= kReduceSolo) {
    for (unsigned i = 0; i < cnt; ++i) {
      if (flags[e0 + i]) {
        const float4* t = reinterpret_cast<const float4*>(tuples + (size_t)(e0 + i) * kRecFloats);
        float4 a = t[0], b = t[1], c = t[2];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w; acc[8] += c.x;
        any = true;
      }
    }
  }
  unsigned long long big = __ballot(cnt > kReduceSolo);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const unsigned c_n = (unsigned)readlane_i((int)cnt, src), c_e = (unsigned)readlane_i((int)e0, src);
    float part[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) part[c] = 0.f;
    bool hit = false;
    for (unsigned i = lane; i < c_n; i += 64) {
      if (flags[c_e + i]) {
        const float4* t = reinterpret_cast<const float4*>(tuples + (size_t)(c_e + i) * kRecFloats);
        float4 a = t[0], b = t[1], c = t[2];
        part[0] += a.x; part[1] += a.y; part[2] += a.z; part[3] += a.w;
        part[4] += b.x; part[5] += b.y; part[6] += b.z; part[7] += b.w; part[8] += c.x;
        hit = true;
      }
    }
    if (__ballot(hit) != 0ull) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const float tsum = wave_sum_uniform(part[c]);
        if (lane == src) acc[c] = tsum;
      }
      if (lane == src) any = true;
    }
  }
  if (any) {
    float4* dst = reinterpret_cast<float4*>(v_records + (size_t)gi * kRecFloats);
    dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    dst[2] = make_float4(acc[8], 0.f, 0.f, 0.f);
    if (touched) touched[gi] = 1;
  }
}

// wave-per-Gaussian form for slices of few, large Gaussians (the nearest slice: ~50k Gaussians owning
// ~400 tiles each): the thread-per-Gaussian form would run 200 blocks with 64-deep serial ballot loops.
__global__ __launch_bounds__(256) void reduce_tuples_wave_kernel(int n_slice, const unsigned* __restrict__ slice_gi,
                                                                 const unsigned* __restrict__ counts,
                                                                 const unsigned* __restrict__ cum,
                                                                 const float* __restrict__ tuples,
                                                                 const unsigned char* __restrict__ flags,
                                                                 float* __restrict__ v_records,
                                                                 unsigned char* __restrict__ touched) {
  const int lane = lane_id();
  const int j = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (j >= n_slice) return;
  const unsigned c_n = counts[j], c_e = cum[j];
  if (c_n == 0) return;
  float part[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) part[c] = 0.f;
  bool hit = false;
  for (unsigned i = lane; i < c_n; i += 64) {
    if (flags[c_e + i]) {
      const float4* t = reinterpret_cast<const float4*>(tuples + (size_t)(c_e + i) * kRecFloats);
      float4 a = t[0], b = t[1], c = t[2];
      part[0] += a.x; part[1] += a.y; part[2] += a.z; part[3] += a.w;
      part[4] += b.x; part[5] += b.y; part[6] += b.z; part[7] += b.w; part[8] += c.x;
      hit = true;
    }
  }
  if (__ballot(hit) == 0ull) return;
  float tot[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) tot[c] = wave_sum_uniform(part[c]);
  if (lane == 0) {
    const unsigned gi = slice_gi[j];
    float4* dst = reinterpret_cast<float4*>(v_records + (size_t)gi * kRecFloats);
    dst[0] = make_float4(tot[0], tot[1], tot[2], tot[3]);
    dst[1] = make_float4(tot[4], tot[5], tot[6], tot[7]);
    dst[2] = make_float4(tot[8], 0.f, 0.f, 0.f);
    if (touched) touched[gi] = 1;
  }
}

// ---------------------------------------------------------------------------
// sub-frame averaging in linearised colour:
//   out = ( mean_k max(C_k, m)^gamma )^(1/gamma),  m = min_rgb_level/255
// ---------------------------------------------------------------------------
// x^y for x > 0 as exp2(y*log2(x)) on the hardware transcendentals (HIP's __powf expands to the
// full-precision ocml pow, ~100 instructions)
__device__ __forceinline__ float fast_pow(float x, float y) {
  return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}

__device__ __forceinline__ float combine_lin(float c, float gamma, float m) {
  if (m > 0.f) c = fmaxf(c, m);
  if (gamma != 1.f) c = fast_pow(fmaxf(c, 1e-12f), gamma);
  return c;
}

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

__device__ __forceinline__ float combine_grad(float c, float g, float gamma, float m) {
  if (m > 0.f && c < m) return 0.f;
  if (gamma != 1.f) {
    if (c < 1e-12f) return 0.f;
    g *= gamma * fast_pow(c, gamma - 1.f);
  }
  return g;
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
    float4 d = make_float4(invS, invS, invS, invS);
    if (gamma != 1.f) {
      d.x *= fast_pow(fmaxf(o.x, 1e-12f), 1.f - gamma) / gamma; d.y *= fast_pow(fmaxf(o.y, 1e-12f), 1.f - gamma) / gamma;
      d.z *= fast_pow(fmaxf(o.z, 1e-12f), 1.f - gamma) / gamma; d.w *= fast_pow(fmaxf(o.w, 1e-12f), 1.f - gamma) / gamma;
    }
    d.x *= vo.x; d.y *= vo.y; d.z *= vo.z; d.w *= vo.w;
    for (int k = 0; k < S; ++k) {
      float4 c = *reinterpret_cast<const float4*>(samples + (size_t)k * n + i);
      float4 g = make_float4(combine_grad(c.x, d.x, gamma, m), combine_grad(c.y, d.y, gamma, m),
                             combine_grad(c.z, d.z, gamma, m), combine_grad(c.w, d.w, gamma, m));
      *reinterpret_cast<float4*>(v_samples + (size_t)k * n + i) = g;
    }
  } else {
    for (size_t j = i; j < n; ++j) {
      float dm = invS * v_out[j];
      if (gamma != 1.f) dm *= fast_pow(fmaxf(out[j], 1e-12f), 1.f - gamma) / gamma;
      for (int k = 0; k < S; ++k) v_samples[(size_t)k * n + j] = combine_grad(samples[(size_t)k * n + j], dm, gamma, m);
    }
  }
}

}  // namespace gs

using namespace gs;

// C ABI -----------------------------------------------------------------------
// Replaces the device side of gsplat.rasterize_gaussians' forward
// (_C.rasterize_forward in the absent fork; SURVEY.md §8 a7, boundary §8b).
GS_EXPORT int gs_rasterize_fwd(const float* records, const int* sorted_vals, const int* tile_bins,
                               const int* band_edges, const float* background, int S, int R, int H, int W,
                               float* out_img, float* out_T, int* final_idx, int variant, void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  RasterParams prm;
  prm.records = records; prm.sorted_vals = sorted_vals; prm.gi_of_e = nullptr;
  prm.tile_bins = reinterpret_cast<const int2*>(tile_bins);
  prm.band_edges = band_edges; prm.background = background;
  prm.S = S; prm.R = R; prm.H = H; prm.W = W;
  prm.tiles_x = (W + K::kTile - 1) / K::kTile; prm.tiles_y = (H + K::kTile - 1) / K::kTile;
  unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
  if (variant == 1)
    hipLaunchKernelGGL(raster_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, out_img, out_T,
                       final_idx, blocks);
  else
    hipLaunchKernelGGL(raster_fwd_kernel_v2, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, out_img, out_T,
                       final_idx, blocks);
  return gs_launch_status();
}

// Replaces _C.rasterize_backward.  v_records must be zeroed by the caller;
// gradients are accumulated with fp32 atomics.
GS_EXPORT int gs_rasterize_bwd(const float* records, const int* sorted_vals, const int* tile_bins,
                               const int* band_edges, const float* background, int S, int R, int H, int W,
                               const float* out_T, const int* final_idx, const float* v_img, const float* v_alpha,
                               float* v_records, void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  RasterParams prm;
  prm.records = records; prm.sorted_vals = sorted_vals; prm.gi_of_e = nullptr;
  prm.tile_bins = reinterpret_cast<const int2*>(tile_bins);
  prm.band_edges = band_edges; prm.background = background;
  prm.S = S; prm.R = R; prm.H = H; prm.W = W;
  prm.tiles_x = (W + K::kTile - 1) / K::kTile; prm.tiles_y = (H + K::kTile - 1) / K::kTile;
  unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
  hipLaunchKernelGGL(raster_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, out_T, final_idx,
                     v_img, v_alpha, v_records, blocks, (float*)nullptr, (float*)nullptr);
  return gs_launch_status();
}

// ---- depth-sliced variants (see binning.hip "depth-sliced binning") ---------------------------------
// One forward launch per slice, slices front to back.  out_img / out_T / live_T carry the per-pixel
// state between launches; tile_done [S*R*T] (zeroed by the caller before the first slice) flags tiles
// whose pixels have all stopped.  first/last mark the first and the final slice (first && last ==
// the unsliced pass).  final_idx is per slice (the backward needs one per slice).
GS_EXPORT int gs_rasterize_fwd_slice(const float* records, const int* sorted_vals, const int* tile_bins,
                                     const int* band_edges, const float* background, int S, int R, int H, int W,
                                     float* out_img, float* out_T, float* live_T, int* final_idx,
                                     unsigned char* tile_done, int first, int last, const int* gi_of_e, int variant,
                                     void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  RasterParams prm;
  prm.records = records; prm.sorted_vals = sorted_vals; prm.gi_of_e = nullptr;
  prm.tile_bins = reinterpret_cast<const int2*>(tile_bins);
  prm.band_edges = band_edges; prm.background = background;
  prm.S = S; prm.R = R; prm.H = H; prm.W = W;
  prm.tiles_x = (W + K::kTile - 1) / K::kTile; prm.tiles_y = (H + K::kTile - 1) / K::kTile;
  prm.gi_of_e = gi_of_e;
  SliceState st; st.tile_done = tile_done; st.live_T = live_T; st.first = first; st.last = last;
  unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
  if (variant == 1)
    hipLaunchKernelGGL(raster_fwd_slice_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, st,
                       out_img, out_T, final_idx, blocks);
  else
    hipLaunchKernelGGL(raster_fwd_slice_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, st,
                       out_img, out_T, final_idx, blocks);
  return gs_launch_status();
}

// One backward launch per slice, slices back to front.  bwd_T (initialised by the caller to out_T) and
// bwd_B [S,H,W,3] (initialised to 0) carry the reverse-traversal state between launches.
GS_EXPORT int gs_rasterize_bwd_slice(const float* records, const int* sorted_vals, const int* tile_bins,
                                     const int* band_edges, const float* background, int S, int R, int H, int W,
                                     const float* out_T, const int* final_idx, const float* v_img,
                                     const float* v_alpha, float* bwd_T, float* bwd_B, float* v_records,
                                     const int* gi_of_e, float* tuples, unsigned char* flags, int variant,
                                     void* stream) {
  if (S <= 0 || R <= 0 || H <= 0 || W <= 0) return GS_ERR_INVALID;
  RasterParams prm;
  prm.records = records; prm.sorted_vals = sorted_vals; prm.gi_of_e = nullptr;
  prm.tile_bins = reinterpret_cast<const int2*>(tile_bins);
  prm.band_edges = band_edges; prm.background = background;
  prm.S = S; prm.R = R; prm.H = H; prm.W = W;
  prm.tiles_x = (W + K::kTile - 1) / K::kTile; prm.tiles_y = (H + K::kTile - 1) / K::kTile;
  unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
  prm.gi_of_e = gi_of_e;
  hipStream_t st = (hipStream_t)stream;
  if (variant == 1) {          // DPP reference kernel (atomics); needs plain Gaussian ids in the list
    if (gi_of_e) return GS_ERR_INVALID;
    hipLaunchKernelGGL(raster_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, prm, out_T, final_idx, v_img, v_alpha,
                       v_records, blocks, bwd_T, bwd_B);
  } else if (variant == 2) {   // ablation: no atomics (timing experiments only)
    hipLaunchKernelGGL((raster_bwd_kernel_v2<true, 2>), dim3(blocks), dim3(256), 0, st, prm, out_T, final_idx, v_img,
                       v_alpha, v_records, blocks, bwd_T, bwd_B, tuples, flags);
  } else if (tuples && flags && gi_of_e) {
    hipLaunchKernelGGL((raster_bwd_kernel_v2<true, 1>), dim3(blocks), dim3(256), 0, st, prm, out_T, final_idx, v_img,
                       v_alpha, v_records, blocks, bwd_T, bwd_B, tuples, flags);
  } else {
    hipLaunchKernelGGL((raster_bwd_kernel_v2<true, 0>), dim3(blocks), dim3(256), 0, st, prm, out_T, final_idx, v_img,
                       v_alpha, v_records, blocks, bwd_T, bwd_B, tuples, flags);
  }
  return gs_launch_status();
}

// Sum each slice Gaussian's gradient tuples (written by gs_rasterize_bwd_slice with tuples != NULL) into
// v_records[slice_gi[j]] (plain stores; Gaussians without a touched entry are left as they are).
GS_EXPORT int gs_reduce_grad_tuples(int n_slice, const unsigned* slice_gi, const unsigned* counts,
                                    const unsigned* cum_excl, const float* tuples, const unsigned char* flags,
                                    float* v_records, unsigned char* touched, long long n_isect, void* stream) {
  if (n_slice <= 0) return GS_ERR_INVALID;
  if (n_isect > 32ll * n_slice)    // few large Gaussians: one wave each
    hipLaunchKernelGGL(reduce_tuples_wave_kernel, dim3((n_slice + 3) / 4), dim3(256), 0, (hipStream_t)stream, n_slice,
                       slice_gi, counts, cum_excl, tuples, flags, v_records, touched);
  else
    hipLaunchKernelGGL(reduce_tuples_kernel, dim3((n_slice + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_slice,
                       slice_gi, counts, cum_excl, tuples, flags, v_records, touched);
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
