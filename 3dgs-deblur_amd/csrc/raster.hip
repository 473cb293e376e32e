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
                                                            const float* __restrict__ v_out, float* __restrict__ scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = combine_scale(out[i], v_out[i], 1.f / (float)S, gamma);
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
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, R, H, W);
  // one pass over the complete tile lists = the sliced kernel with first == last (no persistent state)
  SliceState st; st.tile_done = nullptr; st.live_T = nullptr; st.first = 1; st.last = 1;
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
  RasterParams prm = make_raster_params(records, sorted_vals, tile_bins, band_edges, background, S, R, H, W);
  unsigned work = (unsigned)(S * prm.tiles_x * prm.tiles_y);
  unsigned blocks = (work + 3) / 4;
  prm.gi_of_e = gi_of_e;
  SliceState st; st.tile_done = tile_done; st.live_T = live_T; st.first = first; st.last = last;
  if (variant == 1)
    hipLaunchKernelGGL(raster_fwd_slice_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, st,
                       out_img, out_T, final_idx, blocks);
  else
    hipLaunchKernelGGL(raster_fwd_slice_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, prm, st,
                       out_img, out_T, final_idx, blocks);
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
  unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(combine_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (size_t)n, gamma, out,
                     v_out, scale);
  return gs_launch_status();
}
