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
  return prm;
}

}  // namespace gs
