// raster_common.h — types shared by the forward (raster.hip) and backward (raster_bwd.hip) compositing
// kernels.
#pragma once
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
  // backward only: fold the gamma-space sub-frame average's backward into the pixel prologue.  With
  // cmb_scale != null the kernel's v_img argument holds the SAMPLE IMAGES [S,H,W,3] and every pixel derives its
  // own d loss / d sample from cmb_scale [H,W,3] = d loss / d (mean of the linearised samples) / S
  // (gs_combine_bwd_scale: the part of the chain rule that is the same for all S samples).
  const float* cmb_scale;
  float cmb_gamma, cmb_min;
  // backward only: d min(0.999, o*vis) / d(o*vis) is 0 where the clamp is active (the true derivative, default:
  // K::kAlphaMax); upstream gsplat 0.1.11 lets the gradient through (deviation 1): FLT_MAX
  float alpha_grad_max;
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

__device__ __forceinline__ float combine_grad(float c, float g, float gamma, float m) {
  if (m > 0.f && c < m) return 0.f;
  if (gamma != 1.f) {
    if (c < 1e-12f) return 0.f;
    g *= gamma * fast_pow(c, gamma - 1.f);
  }
  return g;
}

// the sample-independent factor of the average's backward, given the averaged value o and its gradient vo:
// (1/S) * d out / d mean * vo, with d out / d mean = (1/gamma) mean^(1/gamma - 1) = out^(1-gamma) / gamma
__device__ __forceinline__ float combine_scale(float o, float vo, float invS, float gamma) {
  float d = invS;
  if (gamma != 1.f) d *= fast_pow(fmaxf(o, 1e-12f), 1.f - gamma) / gamma;
  return d * vo;
}

// fills the launch parameters shared by every compositing entry point
static inline RasterParams make_raster_params(const float* records, const int* sorted_vals, const int* tile_bins,
                                              const int* band_edges, const float* background, int S, int R, int H,
                                              int W) {
  RasterParams prm;
  prm.records = records; prm.sorted_vals = sorted_vals; prm.gi_of_e = nullptr;
  prm.tile_bins = reinterpret_cast<const int2*>(tile_bins);
  prm.band_edges = band_edges; prm.background = background;
  prm.S = S; prm.R = R; prm.H = H; prm.W = W;
  prm.tiles_x = (W + K::kTile - 1) / K::kTile; prm.tiles_y = (H + K::kTile - 1) / K::kTile;
  prm.cmb_scale = nullptr; prm.cmb_gamma = 1.f; prm.cmb_min = 0.f;
  prm.alpha_grad_max = K::kAlphaMax;
  return prm;
}

}  // namespace gs
