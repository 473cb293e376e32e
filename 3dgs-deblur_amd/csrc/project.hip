// project.hip — per-Gaussian kernels: SE(3) sub-pose interpolation, EWA
// projection (single pose, gsplat-compatible arrays) and the fused multi-sub-pose
// projection + SH colour + antialiased opacity that writes rasterizer records.
//
// Compiled with -ffp-contract=off: radii / tile bounds / depth key bits must be
// bit-identical to oracle/gs_oracle.py::project_gaussians in float32.
//
// Restates (absent fork sources): gsplat project_gaussians
// forward/backward kernels, compute_sh_forward/backward (SURVEY.md §2.3, §8 a1,
// a3, a9; App. A) and the model-level sub-pose loop (§8 a2, a10; north_star).
#include "gs_common.h"
#include "../../include/gsdeblur.h"     // gs_project_inputs (the prototypes are checked against these definitions)

namespace gs {

// ---------------------------------------------------------------------------
// SE(3) screw interpolation of sub-pose viewmats
// ---------------------------------------------------------------------------
__global__ void subpose_fwd_kernel(int P, const float* __restrict__ V0, const float* __restrict__ lin,
                                   const float* __restrict__ ang, const float* __restrict__ times,
                                   float* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float v0[12], l[3], a[3], o[12];
  for (int j = 0; j < 12; ++j) v0[j] = V0[j];
  for (int j = 0; j < 3; ++j) { l[j] = lin[j]; a[j] = ang[j]; }
  subpose_viewmat<float>(v0, l, a, times[p], o);
  for (int j = 0; j < 12; ++j) out[16 * p + j] = o[j];
  out[16 * p + 12] = 0.f; out[16 * p + 13] = 0.f; out[16 * p + 14] = 0.f; out[16 * p + 15] = 1.f;
}

// one work item per (sub-pose, input tangent): item (p, t) seeds input t of the 18 (12 viewmat, 3 lin, 3 ang)
// with a unit dual part and pushes it through the closed form — a Dual<1> chain is short enough to stay in
// registers, where the former one-thread-per-sub-pose Dual<18> version spilled (41 us for 5 sub-poses).
// ONE block (round 6): the P contributions to every tangent are parked in LDS and added up in sub-pose order by one
// thread per tangent — the camera-level gradients are what the pose / velocity optimizers consume
// (/root/reference/train.py:40,66), and fp32 atomics across sub-poses made them differ from run to run.
__global__ __launch_bounds__(256) void subpose_bwd_kernel(int P, const float* __restrict__ V0, const float* __restrict__ lin,
                                   const float* __restrict__ ang, const float* __restrict__ times,
                                   const float* __restrict__ v_out, float* __restrict__ v_V0,
                                   float* __restrict__ v_lin, float* __restrict__ v_ang) {
  typedef Dual<1> D;
  extern __shared__ float sp_part[];          // [P][18]
  for (int gid = threadIdx.x; gid < P * 18; gid += blockDim.x) {
    const int p = gid / 18, t = gid - p * 18;
    D dV[12], dl[3], da[3], o[12];
    for (int j = 0; j < 12; ++j) { dV[j] = D(V0[j]); dV[j].d[0] = (j == t) ? 1.f : 0.f; }
    for (int j = 0; j < 3; ++j) {
      dl[j] = D(lin[j]); dl[j].d[0] = (12 + j == t) ? 1.f : 0.f;
      da[j] = D(ang[j]); da[j].d[0] = (15 + j == t) ? 1.f : 0.f;
    }
    subpose_viewmat<D>(dV, dl, da, D(times[p]), o);
    float acc = 0.f;
    for (int j = 0; j < 12; ++j) acc += v_out[16 * p + j] * o[j].d[0];
    sp_part[gid] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    const int t = threadIdx.x;
    float sum = 0.f;
    for (int p = 0; p < P; ++p) sum += sp_part[p * 18 + t];
    float* dst = t < 12 ? v_V0 + t : (t < 15 ? v_lin + (t - 12) : v_ang + (t - 15));
    *dst += sum;                               // (accumulating contract: the caller zeroes; single writer)
  }
}

// ---------------------------------------------------------------------------
// gsplat-compatible single-pose projection (separate output arrays)
// ---------------------------------------------------------------------------
struct Intrin { float fx, fy, cx, cy; int W, H, tiles_x, tiles_y; float clip; };

__global__ __launch_bounds__(256) void project_fwd_kernel(int N, const float* __restrict__ means,
    const float* __restrict__ scales, float glob, const float* __restrict__ quats, const float* __restrict__ V,
    Intrin in, float* __restrict__ xys, float* __restrict__ depths, int* __restrict__ radii,
    float* __restrict__ conics, float* __restrict__ comp, int* __restrict__ ntiles, float* __restrict__ cov3d,
    int* __restrict__ tbounds) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float Vm[12];
  for (int j = 0; j < 12; ++j) Vm[j] = V[j];
  float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
  float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
  float q[4] = {quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]};
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(s, glob, R, M, c3);
  Proj o; ProjCtx k;
  bool ok = project_one(m, c3, Vm, in.fx, in.fy, in.cx, in.cy, in.W, in.H, in.tiles_x, in.tiles_y, in.clip, o, k);
  for (int j = 0; j < 6; ++j) cov3d[6 * i + j] = c3[j];
  depths[i] = o.depth;
  radii[i] = ok ? o.radius : 0;
  ntiles[i] = ok ? o.ntiles : 0;
  xys[2 * i] = ok ? o.x : 0.f; xys[2 * i + 1] = ok ? o.y : 0.f;
  conics[3 * i] = ok ? o.conic_x : 0.f; conics[3 * i + 1] = ok ? o.conic_y : 0.f; conics[3 * i + 2] = ok ? o.conic_z : 0.f;
  comp[i] = ok ? o.comp : 0.f;
  if (tbounds) {
    tbounds[4 * i] = o.tmin_x; tbounds[4 * i + 1] = o.tmin_y; tbounds[4 * i + 2] = o.tmax_x; tbounds[4 * i + 3] = o.tmax_y;
  }
}

// Camera-level gradients (view matrix, twist), deterministic since round 6: a block adds the 12 components of its
// threads up in a fixed tree (DPP wave sums, then the four waves in order) into a block-private LDS accumulator
// (`acc`, one writer per component; a block that runs the body several times adds its rounds in order), writes the
// accumulator to ITS row of a caller-owned scratch array when it is done, and pose_reduce_kernel adds the rows up in
// block order.  (Rounds 1-5: fp32 atomics onto v_V — the same frame gave gradients that differed in the last bits from
// run to run, and the test bar of exactly these tensors had been widened 3x to absorb it.)
__device__ __forceinline__ void reduce_vV(const float vV[12], float* __restrict__ acc /*LDS [12]*/, float* lds /*[4*12]*/) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    float t = wave_sum_to_lane63(vV[j]);
    if (lane == 63) lds[wave * 12 + j] = t;
  }
  __syncthreads();
  if (threadIdx.x < 12) acc[threadIdx.x] += (lds[threadIdx.x] + lds[12 + threadIdx.x]) + (lds[24 + threadIdx.x] + lds[36 + threadIdx.x]);
  __syncthreads();
}

// partial [nblocks][slots][12] (every row written by its block) -> dst(slot)[0..11] += sum over the blocks, in order:
// thread t adds rows t, t + 256, ... ; the 256 partial sums then go through a fixed tree.  One block per slot.
struct PoseDst { float* slot0; float* slot1; int stride; };   // slot s -> (s == 1 && slot1) ? slot1 : slot0 + s * stride
__global__ __launch_bounds__(256) void pose_reduce_kernel(const float* __restrict__ partial, int nblocks, int slots,
                                                          PoseDst dst) {
  __shared__ double red[256][13];              // (double: the rows are fp32 block sums; adding thousands of them costs nothing)
  const int sl = blockIdx.x;
  float* out = (sl == 1 && dst.slot1) ? dst.slot1 : dst.slot0 + (size_t)sl * dst.stride;
  if (!out) return;
  double a[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) a[j] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const float* row = partial + ((size_t)b * slots + sl) * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j) a[j] += (double)row[j];
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) red[threadIdx.x][j] = a[j];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
#pragma unroll
      for (int j = 0; j < 12; ++j) red[threadIdx.x][j] += red[threadIdx.x + w][j];
    __syncthreads();
  }
  if (threadIdx.x < 12) out[threadIdx.x] += (float)red[0][threadIdx.x];
}

__global__ __launch_bounds__(256) void project_bwd_kernel(int N, const float* __restrict__ means,
    const float* __restrict__ scales, float glob, const float* __restrict__ quats, const float* __restrict__ V,
    Intrin in, const float* __restrict__ v_xys, const float* __restrict__ v_depths,
    const float* __restrict__ v_conics, const float* __restrict__ v_comp, float* __restrict__ v_means,
    float* __restrict__ v_scales, float* __restrict__ v_quats, float* __restrict__ v_V /*scratch row array or null*/,
    int flags) {
  __shared__ float lds[48];
  __shared__ float acc[12];
  if (threadIdx.x < 12) acc[threadIdx.x] = 0.f;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float vV[12];
  for (int j = 0; j < 12; ++j) vV[j] = 0.f;
  if (i < N) {
    float Vm[12];
    for (int j = 0; j < 12; ++j) Vm[j] = V[j];
    float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    float q[4] = {quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]};
    float R[9], qn[4], inv, M[9], c3[6];
    quat_to_rotmat(q, R, qn, &inv);
    scale_rot_to_cov3d(s, glob, R, M, c3);
    Proj o; ProjCtx k;
    bool ok = project_one(m, c3, Vm, in.fx, in.fy, in.cx, in.cy, in.W, in.H, in.tiles_x, in.tiles_y, in.clip, o, k);
    float vm[3] = {0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      float vxy[2] = {v_xys[2 * i], v_xys[2 * i + 1]};
      float vc[3] = {v_conics[3 * i], v_conics[3 * i + 1], v_conics[3 * i + 2]};
      float vc3[6];
      project_one_bwd(m, c3, Vm, in.fx, in.fy, k, o.comp, vxy, v_depths ? v_depths[i] : 0.f, vc,
                      v_comp ? v_comp[i] : 0.f, vm, vc3, vV, nullptr, (flags & 1) != 0);
      cov3d_bwd(s, glob, q, vc3, vs, vq, (flags & 2) != 0);
    }
    for (int j = 0; j < 3; ++j) { v_means[3 * i + j] = vm[j]; v_scales[3 * i + j] = vs[j]; }
    for (int j = 0; j < 4; ++j) v_quats[4 * i + j] = vq[j];
  }
  if (v_V) {
    __syncthreads();
    reduce_vV(vV, acc, lds);
    if (threadIdx.x < 12) v_V[(size_t)blockIdx.x * 12 + threadIdx.x] = acc[threadIdx.x];
  }
}

// ---------------------------------------------------------------------------
// spherical harmonics (gsplat.spherical_harmonics compat)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sh_fwd_kernel(int N, int K_stride, int deg, const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs, float* __restrict__ colors) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
  float inv = 1.0f / sqrtf(x * x + y * y + z * z);
  float B[25];
  sh_basis(deg, x * inv, y * inv, z * inv, B);
  int nb = (deg + 1) * (deg + 1);
  float r = 0.f, g = 0.f, b = 0.f;
  const float* c = coeffs + (size_t)i * K_stride * 3;
#pragma unroll
  for (int k = 0; k < 25; ++k)
    if (k < nb) { r += B[k] * c[3 * k]; g += B[k] * c[3 * k + 1]; b += B[k] * c[3 * k + 2]; }
  colors[3 * i] = r; colors[3 * i + 1] = g; colors[3 * i + 2] = b;
}

__global__ __launch_bounds__(256) void sh_bwd_kernel(int N, int K_stride, int deg, const float* __restrict__ dirs,
                                                     const float* __restrict__ v_colors, float* __restrict__ v_coeffs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
  float inv = 1.0f / sqrtf(x * x + y * y + z * z);
  float B[25];
  sh_basis(deg, x * inv, y * inv, z * inv, B);
  int nb = (deg + 1) * (deg + 1);
  float r = v_colors[3 * i], g = v_colors[3 * i + 1], b = v_colors[3 * i + 2];
  float* c = v_coeffs + (size_t)i * K_stride * 3;
#pragma unroll
  for (int k = 0; k < 25; ++k) {
    if (k < K_stride) {
      float bk = k < nb ? B[k] : 0.f;
      c[3 * k] = bk * r; c[3 * k + 1] = bk * g; c[3 * k + 2] = bk * b;
    }
  }
  for (int k = 25; k < K_stride; ++k) { c[3 * k] = 0.f; c[3 * k + 1] = 0.f; c[3 * k + 2] = 0.f; }
}

// ---------------------------------------------------------------------------
// pack gsplat-style arrays into rasterizer records (compat rasterize path) —
// recomputes the tile bounds from (xy, radius) like upstream's get_tile_bbox.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void tile_bbox(float x, float y, int radius, int tiles_x, int tiles_y, int& x0, int& y0,
                                          int& x1, int& y1) {
  const float inv_tile = 1.0f / (float)K::kTile;
  float radf = (float)radius;
  float tcx = x * inv_tile, tcy = y * inv_tile, tr = radf * inv_tile;
  x0 = (int)(tcx - tr); x1 = (int)(tcx + tr + 1.0f);
  y0 = (int)(tcy - tr); y1 = (int)(tcy + tr + 1.0f);
  x0 = x0 < 0 ? 0 : (x0 > tiles_x ? tiles_x : x0);
  x1 = x1 < 0 ? 0 : (x1 > tiles_x ? tiles_x : x1);
  y0 = y0 < 0 ? 0 : (y0 > tiles_y ? tiles_y : y0);
  y1 = y1 < 0 ? 0 : (y1 > tiles_y ? tiles_y : y1);
}

__global__ __launch_bounds__(256) void pack_records_kernel(int N, const float* __restrict__ xys,
    const float* __restrict__ depths, const int* __restrict__ radii, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacity, int tiles_x, int tiles_y,
    float* __restrict__ records, unsigned* __restrict__ depth_keys, int* __restrict__ ntiles) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int rad = radii[i];
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  float x = xys[2 * i], y = xys[2 * i + 1];
  if (rad > 0) tile_bbox(x, y, rad, tiles_x, tiles_y, x0, y0, x1, y1);
  int area = (x1 - x0) * (y1 - y0);
  bool ok = rad > 0 && area > 0;
  float4* r = reinterpret_cast<float4*>(records + (size_t)i * kRecFloats);
  float d = depths[i];
  if (ok) {
    r[0] = make_float4(x, y, conics[3 * i], conics[3 * i + 1]);
    r[1] = make_float4(conics[3 * i + 2], opacity[i], colors[3 * i], colors[3 * i + 1]);
    r[2] = make_float4(colors[3 * i + 2], d, __int_as_float(x0 | (y0 << 16)), __int_as_float(x1 | (y1 << 16)));
    float aux[4];
    rec_aux(opacity[i], conics[3 * i], conics[3 * i + 2], aux);
    r[3] = make_float4(aux[0], aux[1], aux[2], aux[3]);
  } else {
    r[0] = make_float4(0.f, 0.f, 0.f, 0.f); r[1] = r[0]; r[2] = r[0]; r[3] = r[0];
  }
  depth_keys[i] = ok ? (unsigned)__float_as_int(d) : 0xFFFFFFFFu;
  ntiles[i] = ok ? area : 0;
}

// unpack record gradients into gsplat-style gradient arrays
__global__ __launch_bounds__(256) void unpack_grads_kernel(int N, const float* __restrict__ v_records,
    float* __restrict__ v_xys, float* __restrict__ v_conics, float* __restrict__ v_colors,
    float* __restrict__ v_opacity) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float4* r = reinterpret_cast<const float4*>(v_records + (size_t)i * kGradFloats);
  float4 a = r[0], b = r[1], c = r[2];
  v_xys[2 * i] = a.x; v_xys[2 * i + 1] = a.y;
  v_conics[3 * i] = a.z; v_conics[3 * i + 1] = a.w; v_conics[3 * i + 2] = b.x;
  v_opacity[i] = b.y;
  v_colors[3 * i] = b.z; v_colors[3 * i + 1] = b.w; v_colors[3 * i + 2] = c.x;
}

// ---------------------------------------------------------------------------
// fused multi-sub-pose projection: one thread per Gaussian reads its parameters
// ONCE (means/scales/quats/opacity/SH = 59 floats at degree 3) and projects it
// under all P sub-pose viewmats, evaluating SH colour per sub-pose view direction
// and writing one 48-byte rasterizer record per (sub-pose, Gaussian).
// ---------------------------------------------------------------------------
struct FusedParams {
  int N, P;
  const float* means; const float* scales; const float* quats; const float* opacities;
  const float* sh;       // [N, K_stride, 3]
  const float* viewmats; // [P,16]
  float glob;
  int K_stride, deg, antialiased;
  int defer_color;   // 1: leave rgb = 0, gs_slice_colors fills it for the Gaussians a depth slice actually emits
  int skip_culled;   // 1: write no record for a culled (Gaussian, sub-pose) pair (its row stays uninitialised)
  // 1 (defer_color bit 4, round 5): write NO record at all — only depth keys, tile counts and radii.  The frame path then
  // projects the records of the Gaussians a depth slice actually holds (gs_slice_project_records, same arithmetic):
  // the benchmark scene writes 4 M records (256 MB of the kernel's 314 MB) and its one slice reads 55 k of them.
  int no_records;
  int records_only;  // 1 (round 6, gs_project_records): write the records and NOTHING else (keys, counts and radii exist already)
  // > 1: sub-pose p = s * rs_bands + r only ever composites the tile rows of rolling-shutter band r
  // ([r * tiles_y / R, (r + 1) * tiles_y / R), the formula of ops._band_edges): a pair whose tile box misses its band's
  // rows is culled HERE (culled depth key, no tile count, no record) instead of being keyed, depth-sorted, scanned and
  // planned for nothing, and a box that straddles the band is clipped to it.  radii keeps the un-banded definition.
  int rs_bands;
  Intrin in;
  // pixel-velocity model (SURVEY App. A "Paper's blur/RS model"): ONE projection under viewmats[0] (mid-exposure),
  // sub-pose p re-centres the splat at xy + times[p] * pixel_velocity; twist = {lin[3], ang[3]} (device)
  int pixvel;
  const float* twist;
  const float* times;
  int flags;         // GS_FLAG_* : upstream-compatible gradient conventions (backward only)
  // exact per-row rolling shutter of the pixel-velocity model (rs_half = readout time / 2; 0: off): sub-pose p is the
  // blur sample at times[p], every pixel ROW y then sees the splat at xy + (times[p] + tau(y)) * pv with
  // tau(y) = ((y + 0.5) / H - 0.5) * T_ro — the compositors of raster_rs.hip add the row term, the projection only
  // widens the tile box by the sweep and hands out pv [N,2]; the backward takes d loss / d pv from v_records[9..10]
  float rs_half;
  float* pix_vel_out;
  // round 4: the caller's RAW parameters (splatfacto stores log-scales, opacity logits and the SH coefficients as
  // features_dc [N,3] + features_rest [N,K-1,3]) are taken as they are — no exp / sigmoid / cat launches in front of
  // the projection, no backward launches of theirs behind it:
  //   act bit 0: `scales` holds log-scales (scale = exp), bit 1: `opacities` holds logits (opacity = sigmoid);
  //   sh_rest != null: `sh` is features_dc [N,3] and sh_rest features_rest [N,K_stride-1,3].
  // The backward returns the gradients of what was handed in (d/d log-scale = d/d scale * scale, d/d logit =
  // d/d opacity * opacity (1 - opacity); v_sh / v_sh_rest split the same way).
  int act;
  const float* sh_rest;
};
constexpr int GS_ACT_LOG_SCALES = 1, GS_ACT_OPACITY_LOGITS = 2;

__device__ __forceinline__ void load_scales(const FusedParams& fp, int i, float s[3]) {
  s[0] = fp.scales[3 * i]; s[1] = fp.scales[3 * i + 1]; s[2] = fp.scales[3 * i + 2];
  if (fp.act & GS_ACT_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
}
__device__ __forceinline__ float load_opacity(const FusedParams& fp, int i) {
  const float o = fp.opacities[i];
  return (fp.act & GS_ACT_OPACITY_LOGITS) ? 1.0f / (1.0f + expf(-o)) : o;
}
// coefficient row of Gaussian i as two runs: c0 = basis 0 (3 floats), c1 = bases 1.. ((K_stride-1)*3 floats)
__device__ __forceinline__ void sh_rows(const float* __restrict__ sh, const float* __restrict__ sh_rest, int K_stride,
                                        size_t i, const float*& c0, const float*& c1) {
  if (sh_rest) { c0 = sh + i * 3; c1 = sh_rest + i * (size_t)(K_stride - 1) * 3; }
  else { c0 = sh + i * (size_t)K_stride * 3; c1 = c0 + 3; }
}

constexpr int GS_FLAG_UPSTREAM_FOV_CLAMP_GRAD = 1;   // back-propagate through the fov clamp as if inactive
constexpr int GS_FLAG_RAW_QUAT_GRAD = 2;             // no projection of the quaternion gradient through q/|q|
constexpr int GS_FLAG_NO_NEEDLE_HP = 8;              // skip the double-precision covariance chain of needle Gaussians
constexpr int GS_FLAG_RS_PIXVEL_GRAD = 16;           // pixel-velocity model, exact rolling shutter: v_records[9..10] = d loss / d pv
constexpr float kNeedleRatio = 8.0f;                 // largest / smallest scale above which the chain runs in double

// DEFER (== fp.defer_color, as a template parameter): the SH coefficients are neither loaded nor held — 48 VGPRs of
// zeros otherwise, a third of the kernel's 153 and the difference between 3 and 5 waves per SIMD.
template <int MAXB, bool DEFER>
__global__ __launch_bounds__(256) void project_fused_fwd_kernel(FusedParams fp, float* __restrict__ records,
    unsigned* __restrict__ depth_keys, int* __restrict__ ntiles, int* __restrict__ radii) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= fp.N) return;
  float m[3] = {fp.means[3 * i], fp.means[3 * i + 1], fp.means[3 * i + 2]};
  float s[3];
  load_scales(fp, i, s);
  float q[4] = {fp.quats[4 * i], fp.quats[4 * i + 1], fp.quats[4 * i + 2], fp.quats[4 * i + 3]};
  float opac = load_opacity(fp, i);
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(s, fp.glob, R, M, c3);
  const int nb = (fp.deg + 1) * (fp.deg + 1);
  float coef[DEFER ? 1 : MAXB * 3];
  if (!DEFER) {
    const float *c0, *c1;
    sh_rows(fp.sh, fp.sh_rest, fp.K_stride, (size_t)i, c0, c1);
#pragma unroll
    for (int k = 0; k < MAXB * 3; ++k) coef[DEFER ? 0 : k] = (k < nb * 3) ? (k < 3 ? c0[k] : c1[k - 3]) : 0.f;
  }
  // pixel-velocity model: geometry of the mid-exposure pose once, then one re-centred record per sub-pose
  Proj o0; ProjCtx k0;
  float pv[2] = {0.f, 0.f};
  if (fp.pixvel) {
    float Vm0[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) Vm0[j] = fp.viewmats[j];
    project_one(m, c3, Vm0, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x, fp.in.tiles_y,
                fp.in.clip, o0, k0);
    // A centre outside the projection's fov guard band (|x/z| or |y/z| beyond 1.3 tan(fov/2) at the mid-exposure pose)
    // moves with the Jacobian of the band edge (gs_math.h pixel_velocity): a sub-pose whose re-centred box misses the
    // image drops it below, a large near splat that still covers the image stays.
    if (k0.geom_ok) {
      const float lin[3] = {fp.twist[0], fp.twist[1], fp.twist[2]}, ang[3] = {fp.twist[3], fp.twist[4], fp.twist[5]};
      pixel_velocity(k0.pc, k0.clamp_x ? k0.tx : k0.pc[0], k0.clamp_y ? k0.ty : k0.pc[1], k0.rz, fp.in.fx, fp.in.fy,
                     lin, ang, pv);
    }
    if (fp.pix_vel_out) { fp.pix_vel_out[2 * i] = pv[0]; fp.pix_vel_out[2 * i + 1] = pv[1]; }
  }
  for (int p = 0; p < fp.P; ++p) {
    const float* V = fp.viewmats + (fp.pixvel ? 0 : 16 * p);
    float Vm[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) Vm[j] = V[j];
    Proj o; ProjCtx k;
    bool ok;
    if (fp.pixvel) {
      o = o0; k = k0;
      ok = k0.geom_ok != 0;
      if (ok) {
        const float tau = fp.times[p];
        o.x = o0.x + tau * pv[0];
        o.y = o0.y + tau * pv[1];
        o.radius = (int)k0.radf;
        if (fp.rs_half != 0.f)
          ok = tile_bounds_swept(o.x - fp.rs_half * pv[0], o.y - fp.rs_half * pv[1], o.x + fp.rs_half * pv[0],
                                 o.y + fp.rs_half * pv[1], k0.radf, fp.in.tiles_x, fp.in.tiles_y, o);
        else
          ok = tile_bounds(o.x, o.y, k0.radf, fp.in.tiles_x, fp.in.tiles_y, o);
      }
    } else {
      ok = project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x,
                       fp.in.tiles_y, fp.in.clip, o, k);
    }
    size_t idx = (size_t)p * fp.N + i;
    float4* r = reinterpret_cast<float4*>(records + idx * kRecFloats);
    const int radius_out = ok ? o.radius : 0;
    if (ok && fp.rs_bands > 1) {
      const int rb = p % fp.rs_bands;
      const int by0 = (rb * fp.in.tiles_y) / fp.rs_bands, by1 = ((rb + 1) * fp.in.tiles_y) / fp.rs_bands;
      o.tmin_y = max(o.tmin_y, by0);
      o.tmax_y = min(o.tmax_y, by1);
      ok = o.tmax_y > o.tmin_y;
      o.ntiles = ok ? (o.tmax_x - o.tmin_x) * (o.tmax_y - o.tmin_y) : 0;
    }
    if (fp.no_records) {
      // (light mode: the record is projected again, by gs_slice_project_records, if a depth slice ever holds this pair)
    } else if (ok) {
      // camera centre = -R^T t ; view direction = mean - centre (no gradient, like splatfacto's detach)
      float cxw = -(Vm[0] * Vm[3] + Vm[4] * Vm[7] + Vm[8] * Vm[11]);
      float cyw = -(Vm[1] * Vm[3] + Vm[5] * Vm[7] + Vm[9] * Vm[11]);
      float czw = -(Vm[2] * Vm[3] + Vm[6] * Vm[7] + Vm[10] * Vm[11]);
      float dx = m[0] - cxw, dy = m[1] - cyw, dz = m[2] - czw;
      float dinv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      float cr = 0.f, cg = 0.f, cb = 0.f;
      if (!DEFER) {
        float B[MAXB];
        sh_basis(fp.deg, dx * dinv, dy * dinv, dz * dinv, B);
        cr = 0.5f; cg = 0.5f; cb = 0.5f;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
          if (b < nb) {
            cr += B[b] * coef[DEFER ? 0 : 3 * b]; cg += B[b] * coef[DEFER ? 0 : 3 * b + 1];
            cb += B[b] * coef[DEFER ? 0 : 3 * b + 2];
          }
        }
        cr = fmaxf(cr, 0.f); cg = fmaxf(cg, 0.f); cb = fmaxf(cb, 0.f);
      }
      float op = fp.antialiased ? opac * o.comp : opac;
      float aux[4];
      rec_aux(op, o.conic_x, o.conic_z, aux);
      // (all four 16-byte stores of the 64-byte record: leaving the last one out — or non-temporal stores — made this
      //  kernel SLOWER, 0.107 -> 0.137 / 0.283 ms on the headline; visit r4_v4)
      r[0] = make_float4(o.x, o.y, o.conic_x, o.conic_y);
      if (fp.records_only) {
        // the rows of an earlier depth slice exist already and carry their deferred COLOUR (gs_slice_colors): everything
        // but the three colour floats is written (same values), those are left as they are
        float* rf = reinterpret_cast<float*>(r);
        rf[4] = o.conic_z; rf[5] = op; rf[9] = o.depth;
        rf[10] = __int_as_float(o.tmin_x | (o.tmin_y << 16)); rf[11] = __int_as_float(o.tmax_x | (o.tmax_y << 16));
      } else {
        r[1] = make_float4(o.conic_z, op, cr, cg);
        r[2] = make_float4(cb, o.depth, __int_as_float(o.tmin_x | (o.tmin_y << 16)),
                           __int_as_float(o.tmax_x | (o.tmax_y << 16)));
      }
      r[3] = make_float4(aux[0], aux[1], aux[2], aux[3]);
    } else if (!fp.skip_culled) {
      // (a fifth of the benchmark scene's pairs, nine in ten of a band-aware rolling-shutter frame's: 64 bytes each that
      //  nothing reads once the depth pre-sort drops culled Gaussians — the caller says so with defer_color bit 1)
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      r[0] = z; r[1] = z; r[2] = z; r[3] = z;
    }
    if (fp.records_only) continue;
    depth_keys[idx] = ok ? (unsigned)__float_as_int(o.depth) : 0xFFFFFFFFu;
    ntiles[idx] = ok ? o.ntiles : 0;
    if (radii) radii[idx] = radius_out;
  }
}

// Body shared by the two launch shapes below: Gaussian `i` of this thread (`live` false = idle lane that only
// takes part in the block reductions).  MUST be called block-uniformly (it contains barriers).
// gradient outputs of the fused projection backward (v_sh_rest non-null: v_sh is [N,3], v_sh_rest [N,K_stride-1,3])
struct FusedOut {
  float* v_means; float* v_scales; float* v_quats; float* v_opac; float* v_sh; float* v_sh_rest;
  float* v_viewmats;      // [P,16] accumulated, may be null
  float* v_xy_sum;        // [N,2] or null
  float* v_twist;         // [12] accumulated, pixel-velocity model, may be null
  float* pose_partial;    // scratch [blocks][slots][12] of the ordered pose-gradient reduction (with v_viewmats / v_twist)
};

template <int MAXB>
__device__ __forceinline__ void fused_bwd_body(const FusedParams& fp, const float* __restrict__ records,
    const float* __restrict__ v_records, const FusedOut& out, const unsigned char* __restrict__ touched,
    int i, bool live, float* lds, float* __restrict__ blk_acc /*LDS [slots][12], see reduce_vV*/) {
  float* __restrict__ v_means = out.v_means; float* __restrict__ v_scales = out.v_scales;
  float* __restrict__ v_quats = out.v_quats; float* __restrict__ v_opac = out.v_opac;
  float* __restrict__ v_sh = out.v_sh; float* __restrict__ v_sh_rest = out.v_sh_rest;
  float* __restrict__ v_viewmats = out.v_viewmats; float* __restrict__ v_xy_sum = out.v_xy_sum;
  float* __restrict__ v_twist = out.v_twist;
  const int ii = live ? i : 0;
  float m[3] = {0.f, 0.f, 1.f}, s[3] = {1.f, 1.f, 1.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, opac = 0.f;
  if (live) {
    m[0] = fp.means[3 * ii]; m[1] = fp.means[3 * ii + 1]; m[2] = fp.means[3 * ii + 2];
    load_scales(fp, ii, s);
    q[0] = fp.quats[4 * ii]; q[1] = fp.quats[4 * ii + 1]; q[2] = fp.quats[4 * ii + 2]; q[3] = fp.quats[4 * ii + 3];
    opac = load_opacity(fp, ii);
  }
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(s, fp.glob, R, M, c3);
  const int nb = (fp.deg + 1) * (fp.deg + 1);
  float vcoef[MAXB * 3];
#pragma unroll
  for (int k = 0; k < MAXB * 3; ++k) vcoef[k] = 0.f;
  float vm[3] = {0.f, 0.f, 0.f}, vc3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vop = 0.f, vxs = 0.f, vys = 0.f;
  const bool up_clamp = (fp.flags & GS_FLAG_UPSTREAM_FOV_CLAMP_GRAD) != 0;
  if (fp.pixvel) {
    // pixel-velocity model: the P records of a Gaussian are ONE projection re-centred at xy + tau_p * pv, so their
    // gradients are summed first (d xy = sum_p, d pv = sum_p tau_p * d xy_p) and pushed through one projection
    // backward plus the pixel-velocity VJP (-> camera-space mean, twist)
    float Vm[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) Vm[j] = fp.viewmats[j];
    float vV[12], vtw[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { vV[j] = 0.f; vtw[j] = 0.f; }
    bool mine = false;
    if (live)
      for (int p = 0; p < fp.P; ++p) mine |= (!touched || touched[(size_t)p * fp.N + ii]);
    if (mine) {
      Proj o; ProjCtx k;
      project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x, fp.in.tiles_y,
                  fp.in.clip, o, k);
      if (k.geom_ok) {
        float vxy[2] = {0.f, 0.f}, vpv[2] = {0.f, 0.f}, vcon[3] = {0.f, 0.f, 0.f}, v_comp = 0.f;
        float vr = 0.f, vg = 0.f, vb = 0.f;
        for (int p = 0; p < fp.P; ++p) {
          if (touched && !touched[(size_t)p * fp.N + ii]) continue;
          const float tau = fp.times[p];
          size_t idx = (size_t)p * fp.N + ii;
          const float4* g4 = reinterpret_cast<const float4*>(v_records + idx * kGradFloats);
          float4 ga = g4[0], gb = g4[1], gc = g4[2];
          const float4* r4 = reinterpret_cast<const float4*>(records + idx * kRecFloats);
          float4 ra = r4[0], rb = r4[1], rc = r4[2];
          // a sub-pose in which the re-centred splat covers no tile has an all-zero record and takes no gradient
          if (ra.z == 0.f && ra.w == 0.f && rb.x == 0.f) continue;
          vr += rb.z > 0.f ? gb.z : 0.f; vg += rb.w > 0.f ? gb.w : 0.f; vb += rc.x > 0.f ? gc.x : 0.f;
          if (fp.antialiased) { vop += gb.y * o.comp; v_comp += gb.y * opac; } else { vop += gb.y; }
          vxy[0] += ga.x; vxy[1] += ga.y;
          vpv[0] += tau * ga.x; vpv[1] += tau * ga.y;
          if (fp.flags & GS_FLAG_RS_PIXVEL_GRAD) { vpv[0] += gc.y; vpv[1] += gc.z; }   // the compositor's row-time term
          vcon[0] += ga.z; vcon[1] += ga.w; vcon[2] += gb.x;
        }
        vxs = vxy[0]; vys = vxy[1];
        if (vr != 0.f || vg != 0.f || vb != 0.f) {
          float cxw = -(Vm[0] * Vm[3] + Vm[4] * Vm[7] + Vm[8] * Vm[11]);
          float cyw = -(Vm[1] * Vm[3] + Vm[5] * Vm[7] + Vm[9] * Vm[11]);
          float czw = -(Vm[2] * Vm[3] + Vm[6] * Vm[7] + Vm[10] * Vm[11]);
          float dx = m[0] - cxw, dy = m[1] - cyw, dz = m[2] - czw;
          float dinv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
          float B[MAXB];
          sh_basis(fp.deg, dx * dinv, dy * dinv, dz * dinv, B);
#pragma unroll
          for (int b = 0; b < MAXB; ++b) {
            if (b < nb) { vcoef[3 * b] += B[b] * vr; vcoef[3 * b + 1] += B[b] * vg; vcoef[3 * b + 2] += B[b] * vb; }
          }
        }
        const float lin[3] = {fp.twist[0], fp.twist[1], fp.twist[2]}, ang[3] = {fp.twist[3], fp.twist[4], fp.twist[5]};
        float vpc[3], vlin[3], vang[3];
        pixel_velocity_bwd(k.pc, k.clamp_x ? k.tx : k.pc[0], k.clamp_y ? k.ty : k.pc[1], k.clamp_x, k.clamp_y, up_clamp,
                           k.rz, fp.in.fx, fp.in.fy, lin, ang, vpv, vpc, vlin, vang);
        vtw[0] = vlin[0]; vtw[1] = vlin[1]; vtw[2] = vlin[2]; vtw[3] = vang[0]; vtw[4] = vang[1]; vtw[5] = vang[2];
        project_one_bwd(m, c3, Vm, fp.in.fx, fp.in.fy, k, o.comp, vxy, 0.f, vcon, v_comp, vm, vc3, vV, vpc, up_clamp);
      }
    }
    if (__syncthreads_or(mine)) {
      if (v_viewmats) reduce_vV(vV, blk_acc, lds);
      if (v_twist) reduce_vV(vtw, blk_acc + 12, lds);  // 6 components, padded to the reducer's 12
    }
  } else
  for (int p = 0; p < fp.P; ++p) {
    const float* V = fp.viewmats + 16 * p;
    float Vm[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) Vm[j] = V[j];
    float vV[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) vV[j] = 0.f;
    // early termination leaves most Gaussians without any gradient: their records are neither read nor
    // re-projected (the compositor's tuple reduce marks the ones it touched)
    const bool mine = live && (!touched || touched[(size_t)p * fp.N + ii]);
    if (mine) {
      Proj o; ProjCtx k;
      bool ok = project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x,
                            fp.in.tiles_y, fp.in.clip, o, k);
      if (ok) {
        size_t idx = (size_t)p * fp.N + ii;
        const float4* g4 = reinterpret_cast<const float4*>(v_records + idx * kGradFloats);
        float4 ga = g4[0], gb = g4[1], gc = g4[2];
        const float4* r4 = reinterpret_cast<const float4*>(records + idx * kRecFloats);
        float4 rb = r4[1], rc = r4[2];
        // colour: rgb = max(SH + 0.5, 0)
        float vr = rb.z > 0.f ? gb.z : 0.f, vg = rb.w > 0.f ? gb.w : 0.f, vb = rc.x > 0.f ? gc.x : 0.f;
        if (vr != 0.f || vg != 0.f || vb != 0.f) {
          float cxw = -(Vm[0] * Vm[3] + Vm[4] * Vm[7] + Vm[8] * Vm[11]);
          float cyw = -(Vm[1] * Vm[3] + Vm[5] * Vm[7] + Vm[9] * Vm[11]);
          float czw = -(Vm[2] * Vm[3] + Vm[6] * Vm[7] + Vm[10] * Vm[11]);
          float dx = m[0] - cxw, dy = m[1] - cyw, dz = m[2] - czw;
          float dinv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
          float B[MAXB];
          sh_basis(fp.deg, dx * dinv, dy * dinv, dz * dinv, B);
#pragma unroll
          for (int b = 0; b < MAXB; ++b) {
            if (b < nb) { vcoef[3 * b] += B[b] * vr; vcoef[3 * b + 1] += B[b] * vg; vcoef[3 * b + 2] += B[b] * vb; }
          }
        }
        float v_comp = 0.f;
        if (fp.antialiased) { vop += gb.y * o.comp; v_comp = gb.y * opac; } else { vop += gb.y; }
        float vxy[2] = {ga.x, ga.y};
        vxs += ga.x; vys += ga.y;
        float vcon[3] = {ga.z, ga.w, gb.x};
        float vm1[3], vc31[6];
        project_one_bwd(m, c3, Vm, fp.in.fx, fp.in.fy, k, o.comp, vxy, 0.f, vcon, v_comp, vm1, vc31, vV, nullptr,
                        up_clamp);
        for (int j = 0; j < 3; ++j) vm[j] += vm1[j];
        for (int j = 0; j < 6; ++j) vc3[j] += vc31[j];
      }
    }
    // block-uniform skip: most blocks hold nothing the compositor touched in this sub-pose
    if (v_viewmats && __syncthreads_or(mine)) reduce_vV(vV, blk_acc + 12 * p, lds);
  }
  if (!live) return;
  float vs[3], vq[4];
  cov3d_bwd(s, fp.glob, q, vc3, vs, vq, (fp.flags & GS_FLAG_RAW_QUAT_GRAD) != 0);
  if (fp.act & GS_ACT_LOG_SCALES) { vs[0] *= s[0]; vs[1] *= s[1]; vs[2] *= s[2]; }
  if (fp.act & GS_ACT_OPACITY_LOGITS) vop *= opac * (1.0f - opac);
  for (int j = 0; j < 3; ++j) { v_means[3 * i + j] = vm[j]; v_scales[3 * i + j] = vs[j]; }
  for (int j = 0; j < 4; ++j) v_quats[4 * i + j] = vq[j];
  v_opac[i] = vop;
  if (v_xy_sum) { v_xy_sum[2 * i] = vxs; v_xy_sum[2 * i + 1] = vys; }
  // (v_sh_rest travels in v_sh's slot of FusedOut; the rows split like the inputs)
  float *c0, *c1;
  if (v_sh_rest) { c0 = v_sh + (size_t)i * 3; c1 = v_sh_rest + (size_t)i * (fp.K_stride - 1) * 3; }
  else { c0 = v_sh + (size_t)i * fp.K_stride * 3; c1 = c0 + 3; }
  const int kn = fp.K_stride * 3;
#pragma unroll
  for (int k = 0; k < MAXB * 3; ++k)   // static indices only: a runtime index would push vcoef to scratch
    if (k < kn) { if (k < 3) c0[k] = vcoef[k]; else c1[k - 3] = vcoef[k]; }
  for (int k = MAXB * 3; k < kn; ++k) c1[k - 3] = 0.f;
}

// ---------------------------------------------------------------------------
// Needle fix-up (round 3): for every Gaussian whose scales differ by more than
// `ratio_limit` and that received a gradient, v_means / v_scales / v_quats are recomputed with the covariance chain
// of gs_math.h (project_ctx_t / project_one_bwd_t / cov3d_bwd_t) in DOUBLE, from the fp32 parameters, and overwrite
// what the fp32 kernel above wrote for that row.  The inputs (the compositor's v_xy / v_conic / v_opacity sums in
// v_records) are the same; only the ill-conditioned part of the chain changes precision.
// ---------------------------------------------------------------------------
// work item of the SE(3) model: ONE (needle, sub-pose) pair -> its contribution to v_mean[3] and v_cov3d[6], in double
__device__ __forceinline__ void needle_item_se3(const FusedParams& fp, const float* __restrict__ records,
                                                const float* __restrict__ v_records,
                                                const unsigned char* __restrict__ touched, int i, int p, double out[9]) {
#pragma unroll
  for (int j = 0; j < 9; ++j) out[j] = 0.0;
  if (touched && !touched[(size_t)p * fp.N + i]) return;
  float s[3];
  load_scales(fp, i, s);
  const float m[3] = {fp.means[3 * i], fp.means[3 * i + 1], fp.means[3 * i + 2]};
  const float q[4] = {fp.quats[4 * i], fp.quats[4 * i + 1], fp.quats[4 * i + 2], fp.quats[4 * i + 3]};
  // fp32 covariance for the culling decisions (exactly what the forward took), double covariance for the chain
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(s, fp.glob, R, M, c3);
  const float* V = fp.viewmats + 16 * p;
  float Vm[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) Vm[j] = V[j];
  Proj o; ProjCtx k;
  if (!project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x, fp.in.tiles_y,
                   fp.in.clip, o, k))
    return;
  double Rd[9], qnd[4], invd, Md[9], c3d[6];
  quat_to_rotmat_t<double>(q, Rd, qnd, &invd);
  scale_rot_to_cov3d_t<double>(s, fp.glob, Rd, Md, c3d);
  ProjCtxT<double> kd;
  project_ctx_t<double>(m, c3d, Vm, fp.in.fx, fp.in.fy, fp.in.W, fp.in.H, kd);
  kd.clamp_x = k.clamp_x; kd.clamp_y = k.clamp_y;
  const double comp = ::sqrt(fmax(0.0, kd.det0 / kd.det));
  const size_t idx = (size_t)p * fp.N + i;
  const float4* g4 = reinterpret_cast<const float4*>(v_records + idx * kGradFloats);
  const float4 ga = g4[0], gb = g4[1];
  const double vxy[2] = {ga.x, ga.y}, vcon[3] = {ga.z, ga.w, gb.x};
  const double v_comp = fp.antialiased ? (double)gb.y * (double)load_opacity(fp, i) : 0.0;
  double vV[12];
  project_one_bwd_t<double>(m, c3d, Vm, fp.in.fx, fp.in.fy, kd, comp, vxy, 0.0, vcon, v_comp, out, out + 3, vV, nullptr,
                            (fp.flags & GS_FLAG_UPSTREAM_FOV_CLAMP_GRAD) != 0);
}

// work item of the pixel-velocity model: ONE needle (its P records are one projection: their gradients are summed first)
__device__ __forceinline__ void needle_item_pixvel(const FusedParams& fp, const float* __restrict__ records,
                                                   const float* __restrict__ v_records,
                                                   const unsigned char* __restrict__ touched, int i, double out[9]) {
#pragma unroll
  for (int j = 0; j < 9; ++j) out[j] = 0.0;
  float s[3];
  load_scales(fp, i, s);
  const float m[3] = {fp.means[3 * i], fp.means[3 * i + 1], fp.means[3 * i + 2]};
  const float q[4] = {fp.quats[4 * i], fp.quats[4 * i + 1], fp.quats[4 * i + 2], fp.quats[4 * i + 3]};
  const float opac = load_opacity(fp, i);
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(s, fp.glob, R, M, c3);
  float Vm[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) Vm[j] = fp.viewmats[j];
  Proj o; ProjCtx k;
  project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x, fp.in.tiles_y,
              fp.in.clip, o, k);
  if (!k.geom_ok) return;
  double Rd[9], qnd[4], invd, Md[9], c3d[6];
  quat_to_rotmat_t<double>(q, Rd, qnd, &invd);
  scale_rot_to_cov3d_t<double>(s, fp.glob, Rd, Md, c3d);
  ProjCtxT<double> kd;
  project_ctx_t<double>(m, c3d, Vm, fp.in.fx, fp.in.fy, fp.in.W, fp.in.H, kd);
  kd.clamp_x = k.clamp_x; kd.clamp_y = k.clamp_y;
  const double comp = ::sqrt(fmax(0.0, kd.det0 / kd.det));
  double vxy[2] = {0., 0.}, vcon[3] = {0., 0., 0.}, v_comp = 0.;
  float vpv[2] = {0.f, 0.f};
  for (int p = 0; p < fp.P; ++p) {
    if (touched && !touched[(size_t)p * fp.N + i]) continue;
    const size_t idx = (size_t)p * fp.N + i;
    const float4* g4 = reinterpret_cast<const float4*>(v_records + idx * kGradFloats);
    const float4 ga = g4[0], gb = g4[1], gc = g4[2];
    const float4* r4 = reinterpret_cast<const float4*>(records + idx * kRecFloats);
    const float4 ra = r4[0], rb = r4[1];
    if (ra.z == 0.f && ra.w == 0.f && rb.x == 0.f) continue;
    if (fp.antialiased) v_comp += (double)gb.y * (double)opac;
    vxy[0] += ga.x; vxy[1] += ga.y;
    vpv[0] += fp.times[p] * ga.x; vpv[1] += fp.times[p] * ga.y;
    if (fp.flags & GS_FLAG_RS_PIXVEL_GRAD) { vpv[0] += gc.y; vpv[1] += gc.z; }
    vcon[0] += ga.z; vcon[1] += ga.w; vcon[2] += gb.x;
  }
  const float lin[3] = {fp.twist[0], fp.twist[1], fp.twist[2]}, ang[3] = {fp.twist[3], fp.twist[4], fp.twist[5]};
  float vpc[3], vlin[3], vang[3];
  pixel_velocity_bwd(k.pc, k.clamp_x ? k.tx : k.pc[0], k.clamp_y ? k.ty : k.pc[1], k.clamp_x, k.clamp_y,
                     (fp.flags & GS_FLAG_UPSTREAM_FOV_CLAMP_GRAD) != 0, k.rz, fp.in.fx, fp.in.fy, lin, ang, vpv, vpc,
                     vlin, vang);
  const double vpcd[3] = {vpc[0], vpc[1], vpc[2]};
  double vV[12];
  project_one_bwd_t<double>(m, c3d, Vm, fp.in.fx, fp.in.fy, kd, comp, vxy, 0.0, vcon, v_comp, out, out + 3, vV, vpcd,
                            (fp.flags & GS_FLAG_UPSTREAM_FOV_CLAMP_GRAD) != 0);
}

// A block owns kNeedleChunk consecutive Gaussians.  Phase 1: every thread looks at eight of them (their touched flags
// come as one 8-byte word per sub-pose: a frame in which half a percent of the Gaussians carry a gradient costs a few
// microseconds) and appends the touched ones whose scale ratio exceeds the limit to a list in LDS.  Phase 2: the list
// is worked off in rounds of 256 (needle, sub-pose) ITEMS — one double-precision chain per thread, so a needle's
// latency is one sub-pose's, not P of them in a row — whose partial sums a needle's owner thread adds up in sub-pose
// order (deterministic) before it runs the covariance -> scale / quaternion step and writes the three rows.
constexpr int kNeedleChunk = 2048;

__global__ __launch_bounds__(256) void project_needle_hp_kernel(FusedParams fp, const float* __restrict__ records,
    const float* __restrict__ v_records, float* __restrict__ v_means, float* __restrict__ v_scales,
    float* __restrict__ v_quats, const unsigned char* __restrict__ touched, float ratio_limit) {
  __shared__ int list[kNeedleChunk];
  __shared__ int n_list;
  __shared__ double part[256][9];
  if (threadIdx.x == 0) n_list = 0;
  __syncthreads();
  const int i0 = blockIdx.x * kNeedleChunk + (int)threadIdx.x * 8;
  if (i0 < fp.N) {
    unsigned long long any = ~0ull;
    if (touched) {
      any = 0ull;
      const bool word = (fp.N % 8) == 0 && i0 + 8 <= fp.N;
      for (int p = 0; p < fp.P; ++p) {
        const unsigned char* t = touched + (size_t)p * fp.N + i0;
        if (word) any |= *reinterpret_cast<const unsigned long long*>(t);
        else for (int j = 0; j < 8 && i0 + j < fp.N; ++j) any |= (unsigned long long)t[j] << (8 * j);
      }
    }
    if (any != 0ull) {
      for (int j = 0; j < 8 && i0 + j < fp.N; ++j) {
        if (((any >> (8 * j)) & 0xFFull) == 0ull) continue;
        const int i = i0 + j;
        // (log-scales: the ratio test in the log domain needs no exp)
        const float s0 = fp.scales[3 * i], s1 = fp.scales[3 * i + 1], s2 = fp.scales[3 * i + 2];
        const float smax = fmaxf(s0, fmaxf(s1, s2)), smin = fminf(s0, fminf(s1, s2));
        const bool needle = (fp.act & GS_ACT_LOG_SCALES) ? (smax - smin > __logf(ratio_limit)) : (smax > ratio_limit * smin);
        if (needle) list[atomicAdd(&n_list, 1)] = i;
      }
    }
  }
  __syncthreads();
  const int n = n_list;
  if (n == 0) return;
  const int items_per = fp.pixvel ? 1 : fp.P;                  // <= 256 (kMaxSubposes)
  const int per_round = max(1, 256 / items_per);               // needles per round
  for (int base = 0; base < n; base += per_round) {
    const int cnt = min(per_round, n - base);
    const int a = (int)threadIdx.x / items_per, p = (int)threadIdx.x - a * items_per;
    if (a < cnt) {
      double o9[9];
      if (fp.pixvel) needle_item_pixvel(fp, records, v_records, touched, list[base + a], o9);
      else needle_item_se3(fp, records, v_records, touched, list[base + a], p, o9);
#pragma unroll
      for (int j = 0; j < 9; ++j) part[threadIdx.x][j] = o9[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
      const int i = list[base + threadIdx.x];
      double acc[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
      for (int pp = 0; pp < items_per; ++pp)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += part[threadIdx.x * items_per + pp][j];
      float s[3];
  load_scales(fp, i, s);
      const float q[4] = {fp.quats[4 * i], fp.quats[4 * i + 1], fp.quats[4 * i + 2], fp.quats[4 * i + 3]};
      double vs[3], vq[4];
      cov3d_bwd_t<double>(s, fp.glob, q, acc + 3, vs, vq, (fp.flags & GS_FLAG_RAW_QUAT_GRAD) != 0);
      if (fp.act & GS_ACT_LOG_SCALES) { vs[0] *= (double)s[0]; vs[1] *= (double)s[1]; vs[2] *= (double)s[2]; }
      for (int j = 0; j < 3; ++j) { v_means[3 * i + j] = (float)acc[j]; v_scales[3 * i + j] = (float)vs[j]; }
      for (int j = 0; j < 4; ++j) v_quats[4 * i + j] = (float)vq[j];
    }
    __syncthreads();
  }
}

// dense launch: one thread per Gaussian (no touched flags: every Gaussian gets its gradient written)
template <int MAXB>
__global__ __launch_bounds__(256) void project_fused_bwd_kernel(FusedParams fp, const float* __restrict__ records,
    const float* __restrict__ v_records, FusedOut out) {
  __shared__ float lds[48];
  extern __shared__ float blk_acc[];             // [slots][12]: this block's pose-gradient sums (reduce_vV)
  const int slots = out.pose_partial ? (fp.pixvel ? 2 : fp.P) : 0;
  for (int k = threadIdx.x; k < slots * 12; k += 256) blk_acc[k] = 0.f;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  fused_bwd_body<MAXB>(fp, records, v_records, out, nullptr, i, i < fp.N, lds, blk_acc);
  __syncthreads();
  for (int k = threadIdx.x; k < slots * 12; k += 256) out.pose_partial[(size_t)blockIdx.x * slots * 12 + k] = blk_acc[k];
}

// sparse launch (touched flags): under early termination ~1 % of the Gaussians carry a gradient and they are
// scattered, so with one thread per Gaussian nearly every wave ran the whole body for one or two lanes.  A block owns
// kFusedChunk consecutive Gaussians, compacts the ids of the touched ones into LDS (ballot + prefix, deterministic
// order) and runs the body on dense rounds of 256.
// ZERO_FILL (round 4): the block first zero-fills ITS rows of the five (six) gradient arrays and of v_xy_sum with
// coalesced 16-byte stores — the caller hands over uninitialised buffers and launches no fills of its own (one 236 MB
// fill + one per small output in round 3); without it the caller pre-zeroes every output.
constexpr int kFusedChunk = 2048;

__device__ __forceinline__ void zero_rows(float* __restrict__ base, size_t first, size_t count) {
  // floats [first, first + count) of `base`: leading / trailing floats singly, the 16-byte-aligned middle as float4
  if (!base || count == 0) return;
  float* p = base + first;
  const size_t mis = ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) ? ((16u - (reinterpret_cast<uintptr_t>(p) & 15u)) >> 2) : 0;
  const size_t head = mis < count ? mis : count;
  for (size_t k = threadIdx.x; k < head; k += blockDim.x) p[k] = 0.f;
  const size_t n4 = (count - head) >> 2;
  float4* p4 = reinterpret_cast<float4*>(p + head);
  for (size_t k = threadIdx.x; k < n4; k += blockDim.x) p4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t k = head + 4 * n4 + threadIdx.x; k < count; k += blockDim.x) p[k] = 0.f;
}

// (256, 2): two waves per SIMD = at most 256 VGPRs.  The body sits right at that cliff (256 with the round-3 pixel-
// velocity VJP, 258 with round 4's — one wave per SIMD, and the zero fill of the 236 MB of gradient outputs, which is
// most of this kernel's time on a sparse frame, dropped from 3.3 to 2.3 TB/s: 72 -> 104 us, found by an on-GPU bisect)
template <int MAXB, bool ZERO_FILL>
__global__ __launch_bounds__(256, 2) void project_fused_bwd_sparse_kernel(FusedParams fp,
    const float* __restrict__ records, const float* __restrict__ v_records, FusedOut out,
    const unsigned char* __restrict__ touched /* [P*N] */) {
  __shared__ float lds[48];
  __shared__ int list[kFusedChunk];
  __shared__ int wave_cnt[4];
  extern __shared__ float blk_acc[];             // [slots][12]: this block's pose-gradient sums (reduce_vV)
  const int slots = out.pose_partial ? (fp.pixvel ? 2 : fp.P) : 0;
  for (int k = threadIdx.x; k < slots * 12; k += 256) blk_acc[k] = 0.f;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int base = blockIdx.x * kFusedChunk;
  if (ZERO_FILL) {
    const size_t g0 = (size_t)base, cnt = (size_t)min(kFusedChunk, fp.N - base);
    zero_rows(out.v_means, 3 * g0, 3 * cnt);
    zero_rows(out.v_scales, 3 * g0, 3 * cnt);
    zero_rows(out.v_quats, 4 * g0, 4 * cnt);
    zero_rows(out.v_opac, g0, cnt);
    if (out.v_sh_rest) {
      zero_rows(out.v_sh, 3 * g0, 3 * cnt);
      zero_rows(out.v_sh_rest, (size_t)(fp.K_stride - 1) * 3 * g0, (size_t)(fp.K_stride - 1) * 3 * cnt);
    } else {
      zero_rows(out.v_sh, (size_t)fp.K_stride * 3 * g0, (size_t)fp.K_stride * 3 * cnt);
    }
    zero_rows(out.v_xy_sum, 2 * g0, 2 * cnt);
    // the rows of this block's touched Gaussians are written again below, by other threads of the block
    __threadfence_block();
  }
  int n_list = 0;
  for (int r = 0; r < kFusedChunk / 256; ++r) {
    const int g = base + r * 256 + (int)threadIdx.x;
    bool any = false;
    if (g < fp.N)
      for (int p = 0; p < fp.P; ++p) any |= touched[(size_t)p * fp.N + g] != 0;
    const unsigned long long bal = __ballot(any);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = n_list, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = wave_cnt[w]; if (w < wave) off += c; total += c; }
    if (any) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = g;
    n_list += total;
    __syncthreads();
  }
  for (int k0 = 0; k0 < n_list; k0 += 256) {
    const int k = k0 + (int)threadIdx.x;
    const bool live = k < n_list;
    fused_bwd_body<MAXB>(fp, records, v_records, out, touched, live ? list[k] : 0, live, lds, blk_acc);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < slots * 12; k += 256) out.pose_partial[(size_t)blockIdx.x * slots * 12 + k] = blk_acc[k];
}

// Lazy records (round 5): the records of the n_slice (sub-pose, Gaussian) pairs a depth slice holds — pair j is depth
// rank slice_rank(sd, j) of sorted_gi, the walk gs_slice_counts_exact makes right afterwards — projected by the SAME
// expressions as project_fused_fwd_kernel<., DEFER = true> (quat_to_rotmat, scale_rot_to_cov3d, project_one, the band
// clip, rec_aux), so a record written here is bit-identical to the one the eager kernel would have written.  Pairs with
// a rank are visible by construction (their depth key was not the culled marker); should one fail here all the same, its
// record is zero-filled (the count kernels then see an empty box).  Parameter reads are random gathers (44 bytes per
// pair out of three arrays): fine for the tens of thousands of pairs of an early-terminating frame's slice, not for
// millions — the caller only goes lazy for scenes whose frames stop within their first slice (ops.FrameHints).
__global__ __launch_bounds__(256) void slice_records_kernel(int n_slice, SliceDesc sd, const unsigned* __restrict__ sorted_gi,
                                                            FusedParams fp, float* __restrict__ records) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_slice) return;
  const unsigned gi = sorted_gi[slice_rank(sd, j)];
  const int p = (int)(gi / (unsigned)fp.N), i = (int)(gi - (unsigned)p * (unsigned)fp.N);
  float m[3] = {fp.means[3 * i], fp.means[3 * i + 1], fp.means[3 * i + 2]};
  float sc[3];
  load_scales(fp, i, sc);
  float q[4] = {fp.quats[4 * i], fp.quats[4 * i + 1], fp.quats[4 * i + 2], fp.quats[4 * i + 3]};
  const float opac = load_opacity(fp, i);
  float R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat(q, R, qn, &inv);
  scale_rot_to_cov3d(sc, fp.glob, R, M, c3);
  const float* V = fp.viewmats + 16 * p;
  float Vm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) Vm[k] = V[k];
  Proj o; ProjCtx k;
  bool ok = project_one(m, c3, Vm, fp.in.fx, fp.in.fy, fp.in.cx, fp.in.cy, fp.in.W, fp.in.H, fp.in.tiles_x,
                        fp.in.tiles_y, fp.in.clip, o, k);
  if (ok && fp.rs_bands > 1) {
    const int rb = p % fp.rs_bands;
    const int by0 = (rb * fp.in.tiles_y) / fp.rs_bands, by1 = ((rb + 1) * fp.in.tiles_y) / fp.rs_bands;
    o.tmin_y = max(o.tmin_y, by0);
    o.tmax_y = min(o.tmax_y, by1);
    ok = o.tmax_y > o.tmin_y;
  }
  float4* r = reinterpret_cast<float4*>(records + (size_t)gi * kRecFloats);
  if (ok) {
    const float op = fp.antialiased ? opac * o.comp : opac;
    float aux[4];
    rec_aux(op, o.conic_x, o.conic_z, aux);
    r[0] = make_float4(o.x, o.y, o.conic_x, o.conic_y);
    r[1] = make_float4(o.conic_z, op, 0.f, 0.f);
    r[2] = make_float4(0.f, o.depth, __int_as_float(o.tmin_x | (o.tmin_y << 16)),
                       __int_as_float(o.tmax_x | (o.tmax_y << 16)));
    r[3] = make_float4(aux[0], aux[1], aux[2], aux[3]);
  } else {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r[0] = z; r[1] = z; r[2] = z; r[3] = z;
  }
}

// Deferred SH colour: with early termination only a few percent of the Gaussians are ever composited, so
// the fused projection can skip the 192-byte SH read and the per-sub-pose evaluation for everyone and this
// kernel colours just the Gaussians a depth slice emits (counts[j] > 0).  Same arithmetic as the fused
// kernel (basis, +0.5, clamp >= 0).
// One thread per slice Gaussian does the arithmetic (same order, term for term, as the fused kernel: the colours are
// bit-identical), but the coefficient rows reach it through wave-private LDS: a lane that fetches its own row touches
// 192 contiguous bytes at a random place, so each of its 48 loads costs the texture path 64 different cache lines
// (this kernel was 0.56 ms per frame on the fitted-model-like scene, 1.2 TB/s).  Here 16 lanes fetch one row together
// (12 bytes each, four rows per load instruction), park it in LDS with an odd row stride, and every lane then reads its
// own row conflict-free.
template <int MAXB> constexpr int slice_colors_waves() { return MAXB <= 16 ? 4 : 2; }     // 64 KB of static LDS at most

template <int MAXB>
__global__ __launch_bounds__(256) void slice_colors_kernel(int n_slice, const unsigned* __restrict__ slice_gi,
                                                           const unsigned* __restrict__ counts, int N,
                                                           const float* __restrict__ means,
                                                           const float* __restrict__ sh,
                                                           const float* __restrict__ sh_rest, int K_stride, int deg,
                                                           const float* __restrict__ viewmats,
                                                           float* __restrict__ records) {
  constexpr int kRow = MAXB * 3 + 1;                       // odd stride (49 / 76 -> 77): lane l reads bank (l*kRow + k) % 32
  constexpr int kRowPad = (kRow & 1) ? kRow : kRow + 1;
  __shared__ float s_coef[slice_colors_waves<MAXB>()][64 * kRowPad];
  const int lane = threadIdx.x & 63;
  float* rows = s_coef[threadIdx.x >> 6];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < n_slice && counts[j] != 0;
  const unsigned gi = active ? slice_gi[j] : 0u;
  const unsigned p = gi / (unsigned)N, g = gi - p * (unsigned)N;
  const int nb = (deg + 1) * (deg + 1);
  // cooperative fetch: sub-lane q of a 16-lane group takes basis q (and q + 16 for degree 4) of the group's row
  const int grp = lane >> 4, q = lane & 15;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = it * 4 + grp;                            // row (= lane of the wave that owns the Gaussian)
    const unsigned g_r = (unsigned)__shfl((int)g, r);
    const int act_r = __shfl((int)active, r);
    if (act_r) {
      const float *r0, *r1;
      sh_rows(sh, sh_rest, K_stride, (size_t)g_r, r0, r1);
      for (int b = q; b < nb; b += 16) {
        const float* c = b == 0 ? r0 : r1 + 3 * (b - 1);
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        float* d = rows + r * kRowPad + 3 * b;
        d[0] = c0; d[1] = c1; d[2] = c2;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (!active) return;
  const float* V = viewmats + 16 * p;
  const float m0 = means[3 * g], m1 = means[3 * g + 1], m2 = means[3 * g + 2];
  float cxw = -(V[0] * V[3] + V[4] * V[7] + V[8] * V[11]);
  float cyw = -(V[1] * V[3] + V[5] * V[7] + V[9] * V[11]);
  float czw = -(V[2] * V[3] + V[6] * V[7] + V[10] * V[11]);
  float dx = m0 - cxw, dy = m1 - cyw, dz = m2 - czw;
  float dinv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  float B[MAXB];
  sh_basis(deg, dx * dinv, dy * dinv, dz * dinv, B);
  const float* coef = rows + lane * kRowPad;
  float cr = 0.5f, cg = 0.5f, cb = 0.5f;
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    if (b < nb) { cr += B[b] * coef[3 * b]; cg += B[b] * coef[3 * b + 1]; cb += B[b] * coef[3 * b + 2]; }
  }
  float* r = records + (size_t)gi * kRecFloats;
  r[6] = fmaxf(cr, 0.f); r[7] = fmaxf(cg, 0.f); r[8] = fmaxf(cb, 0.f);
}

}  // namespace gs

using namespace gs;

static inline Intrin make_intrin(float fx, float fy, float cx, float cy, int H, int W, float clip) {
  Intrin in;
  in.fx = fx; in.fy = fy; in.cx = cx; in.cy = cy; in.W = W; in.H = H;
  in.tiles_x = (W + K::kTile - 1) / K::kTile; in.tiles_y = (H + K::kTile - 1) / K::kTile; in.clip = clip;
  return in;
}

// C ABI -------------------------------------------------------------------------
GS_EXPORT int gs_subpose_viewmats_fwd(int P, const float* viewmat, const float* lin_vel, const float* ang_vel,
                                      const float* times, float* out_viewmats, void* stream) {
  if (P <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(subpose_fwd_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, viewmat, lin_vel,
                     ang_vel, times, out_viewmats);
  return gs_launch_status();
}

// v_viewmat[16], v_lin[3], v_ang[3] are ACCUMULATED into (caller zeroes).
GS_EXPORT int gs_subpose_viewmats_bwd(int P, const float* viewmat, const float* lin_vel, const float* ang_vel,
                                      const float* times, const float* v_out, float* v_viewmat, float* v_lin,
                                      float* v_ang, void* stream) {
  if (P <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(subpose_bwd_kernel, dim3(1), dim3(256), (size_t)P * 18 * sizeof(float), (hipStream_t)stream, P, viewmat, lin_vel,
                     ang_vel, times, v_out, v_viewmat, v_lin, v_ang);
  return gs_launch_status();
}

// gsplat.project_gaussians forward (device side).  tile_bounds [N,4] may be null.
GS_EXPORT int gs_project_fwd(int N, const float* means, const float* scales, float glob_scale, const float* quats,
                             const float* viewmat, float fx, float fy, float cx, float cy, int H, int W, float clip,
                             float* xys, float* depths, int* radii, float* conics, float* comp, int* num_tiles_hit,
                             float* cov3d, int* tile_bounds, void* stream) {
  if (N <= 0) return GS_ERR_INVALID;
  Intrin in = make_intrin(fx, fy, cx, cy, H, W, clip);
  hipLaunchKernelGGL(project_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, means, scales,
                     glob_scale, quats, viewmat, in, xys, depths, radii, conics, comp, num_tiles_hit, cov3d,
                     tile_bounds);
  return gs_launch_status();
}

// gsplat.project_gaussians backward.  v_viewmat[16] (rows 0..2 used) is accumulated into; may be null.
GS_EXPORT int gs_project_bwd(int N, const float* means, const float* scales, float glob_scale, const float* quats,
                             const float* viewmat, float fx, float fy, float cx, float cy, int H, int W, float clip,
                             const float* v_xys, const float* v_depths, const float* v_conics, const float* v_comp,
                             float* v_means, float* v_scales, float* v_quats, float* v_viewmat, int grad_flags,
                             void* pose_scratch, long long pose_scratch_bytes_, void* stream) {
  if (N <= 0) return GS_ERR_INVALID;
  Intrin in = make_intrin(fx, fy, cx, cy, H, W, clip);
  const int blocks = (N + 255) / 256;
  // v_viewmat [16] is accumulated into (caller zeroes; nullable): the blocks' sums go through one scratch row each and
  // are added up in block order (gs_project_pose_scratch_bytes(N, 1, 0))
  if (v_viewmat && (!pose_scratch || pose_scratch_bytes_ < (long long)blocks * 12 * (long long)sizeof(float)))
    return GS_ERR_WORKSPACE;
  float* rows = v_viewmat ? reinterpret_cast<float*>(pose_scratch) : nullptr;
  hipLaunchKernelGGL(project_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, means, scales,
                     glob_scale, quats, viewmat, in, v_xys, v_depths, v_conics, v_comp, v_means, v_scales, v_quats,
                     rows, grad_flags);
  if (v_viewmat)
    hipLaunchKernelGGL(pose_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, blocks, 1,
                       PoseDst{v_viewmat, nullptr, 0});
  return gs_launch_status();
}

GS_EXPORT int gs_sh_fwd(int N, int K_stride, int degrees_to_use, const float* dirs, const float* coeffs,
                        float* colors, void* stream) {
  if (N <= 0 || degrees_to_use < 0 || degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K_stride)
    return GS_ERR_INVALID;
  hipLaunchKernelGGL(sh_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, K_stride,
                     degrees_to_use, dirs, coeffs, colors);
  return gs_launch_status();
}

GS_EXPORT int gs_sh_bwd(int N, int K_stride, int degrees_to_use, const float* dirs, const float* v_colors,
                        float* v_coeffs, void* stream) {
  if (N <= 0 || degrees_to_use < 0 || degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K_stride)
    return GS_ERR_INVALID;
  hipLaunchKernelGGL(sh_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, K_stride,
                     degrees_to_use, dirs, v_colors, v_coeffs);
  return gs_launch_status();
}

GS_EXPORT int gs_pack_records(int N, const float* xys, const float* depths, const int* radii, const float* conics,
                              const float* colors, const float* opacity, int H, int W, float* records,
                              unsigned* depth_keys, int* ntiles, void* stream) {
  if (N <= 0) return GS_ERR_INVALID;
  int tiles_x = (W + K::kTile - 1) / K::kTile, tiles_y = (H + K::kTile - 1) / K::kTile;
  hipLaunchKernelGGL(pack_records_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, xys, depths,
                     radii, conics, colors, opacity, tiles_x, tiles_y, records, depth_keys, ntiles);
  return gs_launch_status();
}

GS_EXPORT int gs_unpack_record_grads(int N, const float* v_records, float* v_xys, float* v_conics, float* v_colors,
                                     float* v_opacity, void* stream) {
  if (N <= 0) return GS_ERR_INVALID;
  hipLaunchKernelGGL(unpack_grads_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, v_records,
                     v_xys, v_conics, v_colors, v_opacity);
  return gs_launch_status();
}

static inline FusedParams make_fused(int N, int P, const float* means, const float* scales, float glob,
                                     const float* quats, const float* opac, const float* sh, int K_stride, int deg,
                                     const float* viewmats, float fx, float fy, float cx, float cy, int H, int W,
                                     float clip, int antialiased, int defer_color = 0) {
  FusedParams fp;
  fp.N = N; fp.P = P; fp.means = means; fp.scales = scales; fp.quats = quats; fp.opacities = opac; fp.sh = sh;
  fp.viewmats = viewmats; fp.glob = glob; fp.K_stride = K_stride; fp.deg = deg; fp.antialiased = antialiased;
  fp.defer_color = defer_color & 1;
  fp.skip_culled = (defer_color >> 1) & 1;
  fp.no_records = (defer_color >> 4) & 1;
  fp.records_only = 0;
  fp.rs_bands = (defer_color & 4) ? ((defer_color >> 8) & 0xFFFF) : 0;
  fp.in = make_intrin(fx, fy, cx, cy, H, W, clip);
  fp.pixvel = 0; fp.twist = nullptr; fp.times = nullptr; fp.flags = 0; fp.rs_half = 0.f; fp.pix_vel_out = nullptr;
  fp.act = 0; fp.sh_rest = nullptr;
  return fp;
}

// Fused projection of N Gaussians under P sub-pose viewmats (+SH colour, antialiased opacity).
//   records    [P*N*12] f32, depth_keys [P*N] u32, num_tiles_hit [P*N] i32, radii [P*N] i32 (nullable)
GS_EXPORT int gs_project_fused_fwd(int N, int P, const float* means, const float* scales, float glob_scale,
                                   const float* quats, const float* opacities, const float* sh, int K_stride,
                                   int sh_degree, const float* viewmats, float fx, float fy, float cx, float cy,
                                   int H, int W, float clip, int antialiased, int defer_color, float* records,
                                   unsigned* depth_keys, int* num_tiles_hit, int* radii, const float* sh_rest,
                                   int param_flags, void* stream) {
  if (N <= 0 || P <= 0 || sh_degree < 0 || sh_degree > 4 || (sh_degree + 1) * (sh_degree + 1) > K_stride)
    return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, means, scales, glob_scale, quats, opacities, sh, K_stride, sh_degree, viewmats,
                              fx, fy, cx, cy, H, W, clip, antialiased, defer_color);
  fp.act = param_flags; fp.sh_rest = sh_rest;
  dim3 grid((N + 255) / 256), block(256);
  if (fp.defer_color)       // the SH degree plays no part: one instantiation
    hipLaunchKernelGGL((project_fused_fwd_kernel<16, true>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  else if (sh_degree <= 3)
    hipLaunchKernelGGL((project_fused_fwd_kernel<16, false>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  else
    hipLaunchKernelGGL((project_fused_fwd_kernel<25, false>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  return gs_launch_status();
}

// Records of the pairs of one depth slice, for a frame projected with defer_color bit 4 (no records): see
// slice_records_kernel.  slice_begin / slice_prefix / sorted_gi as gs_slice_counts_exact takes them; `in` holds the
// arguments the frame's gs_project_fused_fwd call was made with (colour deferred: rgb = 0, gs_slice_colors follows).
GS_EXPORT int gs_slice_project_records(int n_slice, int P, int N, const int* slice_begin, const int* slice_prefix,
                                       const unsigned* sorted_gi, const gs_project_inputs* in, int H, int W,
                                       float* records, void* stream) {
  SliceDesc sd;
  if (n_slice <= 0 || N <= 0 || !in || !sorted_gi || !records || !make_slice_desc(P, slice_begin, slice_prefix, sd))
    return GS_ERR_INVALID;
  if (!in->means || !in->scales || !in->quats || !in->opacities || !in->viewmats) return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, in->means, in->scales, in->glob_scale, in->quats, in->opacities, nullptr, 1, 0,
                              in->viewmats, in->fx, in->fy, in->cx, in->cy, H, W, in->clip_thresh, in->antialiased,
                              in->defer_color | 1);
  fp.act = in->param_flags;
  hipLaunchKernelGGL(slice_records_kernel, dim3((n_slice + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_slice, sd,
                     sorted_gi, fp, records);
  return gs_launch_status();
}

// The records of ALL pairs of a frame projected with defer_color bit 4, in one coalesced pass of the eager kernel (which
// writes nothing but records here: the frame's keys, counts and radii exist and the keys may be consumed already).  A
// lazy frame that goes beyond its first depth slice calls this once instead of projecting slice after slice by gathers
// (round 6: a view that looks past the scene's edge walks every slice of its plan, and 4 M gathered pairs cost 0.6 ms
// where this pass costs 0.08).  Rows bit-identical to gs_project_fused_fwd's and to gs_slice_project_records'.
GS_EXPORT int gs_project_records(int P, int N, const gs_project_inputs* in, int H, int W, float* records, void* stream) {
  if (P <= 0 || N <= 0 || !in || !records) return GS_ERR_INVALID;
  if (!in->means || !in->scales || !in->quats || !in->opacities || !in->viewmats) return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, in->means, in->scales, in->glob_scale, in->quats, in->opacities, nullptr, 1, 0,
                              in->viewmats, in->fx, in->fy, in->cx, in->cy, H, W, in->clip_thresh, in->antialiased,
                              (in->defer_color | 1 | 2) & ~16);
  fp.act = in->param_flags;
  fp.records_only = 1;
  hipLaunchKernelGGL((project_fused_fwd_kernel<16, true>), dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, fp,
                     records, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr);
  return gs_launch_status();
}

// Colours (rgb = max(SH + 0.5, 0)) of the slice Gaussians with counts[j] > 0, written into their records;
// pairs with gs_project_fused_fwd(defer_color = 1).
GS_EXPORT int gs_slice_colors(int n_slice, const unsigned* slice_gi, const unsigned* counts, int N,
                              const float* means, const float* sh, const float* sh_rest, int K_stride, int sh_degree,
                              const float* viewmats, float* records, void* stream) {
  if (n_slice <= 0 || N <= 0 || sh_degree < 0 || sh_degree > 4 || (sh_degree + 1) * (sh_degree + 1) > K_stride)
    return GS_ERR_INVALID;
  if (sh_degree <= 3) {
    const int th = 64 * slice_colors_waves<16>();
    hipLaunchKernelGGL(slice_colors_kernel<16>, dim3((n_slice + th - 1) / th), dim3(th), 0, (hipStream_t)stream,
                       n_slice, slice_gi, counts, N, means, sh, sh_rest, K_stride, sh_degree, viewmats, records);
  } else {
    const int th = 64 * slice_colors_waves<25>();
    hipLaunchKernelGGL(slice_colors_kernel<25>, dim3((n_slice + th - 1) / th), dim3(th), 0, (hipStream_t)stream,
                       n_slice, slice_gi, counts, N, means, sh, sh_rest, K_stride, sh_degree, viewmats, records);
  }
  return gs_launch_status();
}

constexpr int GS_FLAG_ZERO_FILL = 32;     // sparse backward: zero-fill the dense gradient outputs in the kernel itself

// bytes of scratch the ordered pose-gradient reduction of a projection backward takes (reduce_vV / pose_reduce_kernel):
// one row of 12 floats per block and slot.  with_touched: the sparse form (a block per kFusedChunk Gaussians) runs.
static long long pose_scratch_bytes(int N, int P, bool pixvel, bool with_touched) {
  const long long blocks = with_touched ? (N + kFusedChunk - 1) / kFusedChunk : (N + 255) / 256;
  return blocks * (long long)(pixvel ? 2 : P) * 12 * (long long)sizeof(float);
}

static int launch_fused_bwd(const FusedParams& fp, int sh_degree, const float* records, const float* v_records,
                            FusedOut out, const unsigned char* touched, void* scratch, long long scratch_bytes,
                            hipStream_t st) {
  dim3 block(256);
  const int N = fp.N;
  const bool pose = out.v_viewmats || out.v_twist;
  const int slots = pose ? (fp.pixvel ? 2 : fp.P) : 0;
  const size_t lds = (size_t)slots * 12 * sizeof(float);
  if (pose) {
    if (!scratch || scratch_bytes < pose_scratch_bytes(N, fp.P, fp.pixvel != 0, touched != nullptr)) return GS_ERR_WORKSPACE;
    out.pose_partial = reinterpret_cast<float*>(scratch);
  }
  dim3 grid(touched ? (N + kFusedChunk - 1) / kFusedChunk : (N + 255) / 256);
  if (touched) {
    const bool zf = (fp.flags & GS_FLAG_ZERO_FILL) != 0;
    if (sh_degree <= 3 && zf)
      hipLaunchKernelGGL((project_fused_bwd_sparse_kernel<16, true>), grid, block, lds, st, fp, records, v_records, out, touched);
    else if (sh_degree <= 3)
      hipLaunchKernelGGL((project_fused_bwd_sparse_kernel<16, false>), grid, block, lds, st, fp, records, v_records, out, touched);
    else if (zf)
      hipLaunchKernelGGL((project_fused_bwd_sparse_kernel<25, true>), grid, block, lds, st, fp, records, v_records, out, touched);
    else
      hipLaunchKernelGGL((project_fused_bwd_sparse_kernel<25, false>), grid, block, lds, st, fp, records, v_records, out, touched);
  } else {
    if (sh_degree <= 3)
      hipLaunchKernelGGL(project_fused_bwd_kernel<16>, grid, block, lds, st, fp, records, v_records, out);
    else
      hipLaunchKernelGGL(project_fused_bwd_kernel<25>, grid, block, lds, st, fp, records, v_records, out);
  }
  if (pose) {
    // the blocks' rows, added up in block order: v_viewmats[p] (SE(3): one slot per sub-pose, 16 floats apart), or
    // v_viewmat and v_twist (pixel-velocity model: slots 0 and 1)
    PoseDst dst{out.v_viewmats, fp.pixvel ? out.v_twist : nullptr, fp.pixvel ? 0 : 16};
    hipLaunchKernelGGL(pose_reduce_kernel, dim3(slots), dim3(256), 0, st, out.pose_partial, (int)grid.x, slots, dst);
  }
  if (!(fp.flags & GS_FLAG_NO_NEEDLE_HP))
    // needles (scale ratio above kNeedleRatio) get their means / scales / quaternion gradients again, in double
    hipLaunchKernelGGL(project_needle_hp_kernel, dim3((N + kNeedleChunk - 1) / kNeedleChunk), dim3(256), 0, st, fp, records,
                       v_records, out.v_means, out.v_scales, out.v_quats, touched, kNeedleRatio);
  return gs_launch_status();
}

// Scratch of the deterministic pose-gradient reduction (gs_project_bwd: P = 1, with_touched = 0; gs_project_fused_bwd /
// gs_project_pixvel_bwd: with_touched = whether touched flags are passed).  Needed only when a view-matrix / twist
// gradient is asked for.
GS_EXPORT long long gs_project_pose_scratch_bytes(int N, int P, int with_touched) {
  if (N <= 0 || P <= 0) return 0;
  return pose_scratch_bytes(N, std::max(P, 2), false, with_touched != 0) + 256;
}

// Backward of the fused projection.  v_viewmats [P,16] is accumulated into (caller zeroes; nullable; with it,
// pose_scratch of gs_project_pose_scratch_bytes(N, P, touched != NULL) bytes: the sum over the Gaussians is ordered).
// grad_flags: 1 = back-propagate through the fov clamp as upstream gsplat 0.1.11 does (as if inactive), 2 = return
// the quaternion gradient without the projection through q/|q| (deviations 2 and 3).
GS_EXPORT int gs_project_fused_bwd(int N, int P, const float* means, const float* scales, float glob_scale,
                                   const float* quats, const float* opacities, const float* sh, int K_stride,
                                   int sh_degree, const float* viewmats, float fx, float fy, float cx, float cy,
                                   int H, int W, float clip, int antialiased, const float* records,
                                   const float* v_records, float* v_means, float* v_scales, float* v_quats,
                                   float* v_opacities, float* v_sh, float* v_viewmats,
                                   const unsigned char* touched, float* v_xy_sum, int grad_flags, const float* sh_rest,
                                   int param_flags, float* v_sh_rest, void* pose_scratch, long long pose_scratch_bytes_,
                                   void* stream) {
  if (N <= 0 || P <= 0 || sh_degree < 0 || sh_degree > 4 || (sh_degree + 1) * (sh_degree + 1) > K_stride ||
      (sh_rest != nullptr) != (v_sh_rest != nullptr) || ((grad_flags & GS_FLAG_ZERO_FILL) && !touched))
    return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, means, scales, glob_scale, quats, opacities, sh, K_stride, sh_degree, viewmats,
                              fx, fy, cx, cy, H, W, clip, antialiased);
  fp.flags = grad_flags; fp.act = param_flags; fp.sh_rest = sh_rest;
  const FusedOut out = {v_means, v_scales, v_quats, v_opacities, v_sh, v_sh_rest, v_viewmats, v_xy_sum, nullptr, nullptr};
  return launch_fused_bwd(fp, sh_degree, records, v_records, out, touched, pose_scratch, pose_scratch_bytes_,
                          (hipStream_t)stream);
}

// ---- pixel-velocity model (the paper's first-order blur / rolling-shutter model; SURVEY App. A, C1;
// /root/reference/README.md:200 "Fixed a bug in pixel velocity formulas") ---------------------------------------------
// ONE projection under `viewmat` (mid-exposure pose); sub-pose p's record is the same splat re-centred at
// xy + times[p] * pv, pv = J(-(ang x pc + lin)); conic, opacity, colour and the depth key are shared by all P.
GS_EXPORT int gs_project_pixvel_fwd(int N, int P, const float* means, const float* scales, float glob_scale,
                                    const float* quats, const float* opacities, const float* sh, int K_stride,
                                    int sh_degree, const float* viewmat, const float* twist, const float* times,
                                    float fx, float fy, float cx, float cy, int H, int W, float clip, int antialiased,
                                    int defer_color, float* records, unsigned* depth_keys, int* num_tiles_hit,
                                    int* radii, float rolling_shutter_time, float* pix_vel, const float* sh_rest,
                                    int param_flags, void* stream) {
  if (N <= 0 || P <= 0 || sh_degree < 0 || sh_degree > 4 || (sh_degree + 1) * (sh_degree + 1) > K_stride || !twist ||
      !times)
    return GS_ERR_INVALID;
  if (rolling_shutter_time != 0.f && !pix_vel) return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, means, scales, glob_scale, quats, opacities, sh, K_stride, sh_degree, viewmat,
                              fx, fy, cx, cy, H, W, clip, antialiased, defer_color);
  fp.pixvel = 1; fp.twist = twist; fp.times = times;
  fp.rs_half = 0.5f * rolling_shutter_time; fp.pix_vel_out = pix_vel;
  fp.act = param_flags; fp.sh_rest = sh_rest;
  dim3 grid((N + 255) / 256), block(256);
  if (fp.defer_color)       // the SH degree plays no part: one instantiation
    hipLaunchKernelGGL((project_fused_fwd_kernel<16, true>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  else if (sh_degree <= 3)
    hipLaunchKernelGGL((project_fused_fwd_kernel<16, false>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  else
    hipLaunchKernelGGL((project_fused_fwd_kernel<25, false>), grid, block, 0, (hipStream_t)stream, fp, records,
                       depth_keys, num_tiles_hit, radii);
  return gs_launch_status();
}

// v_viewmat [16] and v_twist [12: lin 3, ang 3, 6 unused] are accumulated into (caller zeroes; nullable; with either,
// pose_scratch as in gs_project_fused_bwd)
GS_EXPORT int gs_project_pixvel_bwd(int N, int P, const float* means, const float* scales, float glob_scale,
                                    const float* quats, const float* opacities, const float* sh, int K_stride,
                                    int sh_degree, const float* viewmat, const float* twist, const float* times,
                                    float fx, float fy, float cx, float cy, int H, int W, float clip, int antialiased,
                                    const float* records, const float* v_records, float* v_means, float* v_scales,
                                    float* v_quats, float* v_opacities, float* v_sh, float* v_viewmat, float* v_twist,
                                    const unsigned char* touched, float* v_xy_sum, int grad_flags, const float* sh_rest,
                                    int param_flags, float* v_sh_rest, void* pose_scratch, long long pose_scratch_bytes_,
                                    void* stream) {
  if (N <= 0 || P <= 0 || sh_degree < 0 || sh_degree > 4 || (sh_degree + 1) * (sh_degree + 1) > K_stride || !twist ||
      !times || (sh_rest != nullptr) != (v_sh_rest != nullptr) || ((grad_flags & GS_FLAG_ZERO_FILL) && !touched))
    return GS_ERR_INVALID;
  FusedParams fp = make_fused(N, P, means, scales, glob_scale, quats, opacities, sh, K_stride, sh_degree, viewmat,
                              fx, fy, cx, cy, H, W, clip, antialiased);
  fp.pixvel = 1; fp.twist = twist; fp.times = times; fp.flags = grad_flags;
  fp.act = param_flags; fp.sh_rest = sh_rest;
  const FusedOut out = {v_means, v_scales, v_quats, v_opacities, v_sh, v_sh_rest, v_viewmat, v_xy_sum, v_twist, nullptr};
  return launch_fused_bwd(fp, sh_degree, records, v_records, out, touched, pose_scratch, pose_scratch_bytes_,
                          (hipStream_t)stream);
}
