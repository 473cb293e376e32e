// gs_math.h — per-Gaussian math of the hot path, shared by every HIP kernel.
//
// Everything here is `GS_HD` (host+device) plain C++ so that the very same
// code can be (a) inlined into the gfx950 kernels and (b) compiled with g++
// into tests/host_math (TEST INFRASTRUCTURE ONLY) and checked against the
// float64 oracle on a box without a GPU.  Nothing in the product path calls
// the host build.
//
// What it restates (the reference's fork sources are NOT in /root/reference —
// see SURVEY.md §0; the spec is SURVEY.md App. A, recollected from upstream
// gsplat 0.1.11, base commit named at /root/reference/README.md:199):
//   project_gaussians fwd/bwd  (SURVEY §8 a1,a3)
//   spherical_harmonics fwd/bwd
//   SE(3) screw interpolation of sub-poses (north_star)
// All recollected constants live in `gs::K` below so they can be corrected in
// one place.
//
// Floating-point contract: the projection path is compiled with
// -ffp-contract=off and written with an explicit left-to-right association so
// the integer outputs (radii, tile bounds, depth key bits) are bit-identical
// to the oracle's float32 restatement (oracle/gs_oracle.py::project_f32).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD inline
#endif

namespace gs {

// ---- recollected constants ------------------------------
struct K {
  static constexpr float kFovLimit   = 1.3f;    // clamp of x/z, y/z: 1.3 * tan(fov/2)
  static constexpr float kDilation   = 0.3f;    // EWA low-pass added to cov2d diagonal
  static constexpr float kMinEigDisc = 0.1f;    // max(0.1, mid^2 - det)
  static constexpr float kRadiusSig  = 3.0f;    // radius = ceil(3 sqrt(lambda_max))
  static constexpr float kAlphaMax   = 0.999f;  // alpha = min(0.999, o * exp(-sigma))
  static constexpr float kAlphaMin   = 1.0f / 255.0f;  // skip below
  static constexpr float kTMin       = 1e-4f;   // stop when T*(1-alpha) <= 1e-4
  static constexpr int   kTile       = 16;      // block_width
};

// Per-Gaussian projected record consumed by the rasterizer: 16 floats = 64 B (one cache line, four 16-byte
// loads per gather; the compositors fetch dwords 0..8(9) and 12..15 through the scalar cache).
//   [0] x  [1] y  [2] conic.x  [3] conic.y | [4] conic.z [5] opacity [6] r [7] g |
//   [8] b  [9] depth [10] tile box min (x | y<<16) [11] tile box max | [12] nmid [13] kmul [14] qx [15] qz
// [12..15] are the compositors' per-entry constants (round 4), functions of [2],[4],[5] alone (rec_aux below):
// with s2 = -log2(e) * sigma (sigma = the conic form of the pixel offset) the blend needs
//   alpha = op * 2^s2   and the test   sigma >= 0  and  alpha >= 1/255   <=>   -log2(255 op) <= s2 <= 0 .
// The interval is made symmetric: nmid = log2(255 op) / 2, u = s2 + nmid, and the test is ONE compare |u| <= nmid
// (absolute value is a free source modifier), alpha = kmul * 2^u with kmul = op * 2^-nmid = sqrt(op / 255).
// An opacity below 1/255 gives nmid < 0: never valid.  qx, qz = conic.x, conic.z pre-scaled by -log2(e) / 2.
constexpr int kRecFloats = 16;
// gradient records / gradient tuples keep 12 floats (x y conic3 opacity rgb | two pixel-velocity slots | pad)
constexpr int kGradFloats = 12;
constexpr int kRecNmid = 12, kRecKmul = 13, kRecQx = 14, kRecQz = 15;
constexpr float kNegLog2e = -1.4426950408889634f;

struct Proj {
  float x, y, depth;
  float conic_x, conic_y, conic_z;
  float comp;
  int   radius;
  int   tmin_x, tmin_y, tmax_x, tmax_y;
  int   ntiles;
  float cov3d[6];
};

// the compositors' per-entry constants of a record, see the layout above
GS_HD void rec_aux(float op, float conic_x, float conic_z, float out[4]) {
  float nmid = -1.0f, kmul = 0.0f;
  if (op >= K::kAlphaMin) {              // (NaN and op < 1/255: never blended)
#if defined(__HIP_DEVICE_COMPILE__)
    // hardware transcendentals (1 ulp): the compositor only needs nmid and kmul to be consistent with each other
    // (alpha = kmul * 2^(s2 + nmid) = op * 2^s2), and 1e-7 relative on the position of the alpha = 1/255 edge
    nmid = 0.5f * __builtin_amdgcn_logf(255.0f * op);
    kmul = op * __builtin_amdgcn_exp2f(-nmid);
#else
    nmid = 0.5f * log2f(255.0f * op);
    kmul = op * exp2f(-nmid);
#endif
  }
  out[0] = nmid; out[1] = kmul; out[2] = conic_x * (0.5f * kNegLog2e); out[3] = conic_z * (0.5f * kNegLog2e);
}

// normalised quaternion (w,x,y,z) -> rotation matrix, row-major R[9]
GS_HD void quat_to_rotmat(const float q[4], float R[9], float qn[4], float* inv_norm) {
  float n2 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
  float inv = 1.0f / sqrtf(n2);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; *inv_norm = inv;
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// cov3d (upper triangle, 6) = (R S)(R S)^T ; also returns M = R S (row-major 9)
GS_HD void scale_rot_to_cov3d(const float s[3], float glob, const float R[9], float M[9], float c[6]) {
  float s0 = glob * s[0], s1 = glob * s[1], s2 = glob * s[2];
  M[0] = R[0] * s0; M[1] = R[1] * s1; M[2] = R[2] * s2;
  M[3] = R[3] * s0; M[4] = R[4] * s1; M[5] = R[5] * s2;
  M[6] = R[6] * s0; M[7] = R[7] * s1; M[8] = R[8] * s2;
  c[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
  c[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
  c[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
  c[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
  c[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
  c[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
}

// Intermediates of one projection that the backward pass re-derives.
struct ProjCtx {
  float pc[3];      // camera-space mean
  float rz;         // 1/z
  float tx, ty;     // fov-clamped x,y
  int   clamp_x, clamp_y;  // -1/0/+1 which side the clamp hit
  float J00, J02, J11, J12;
  float T[6];       // J*W, 2x3
  float a, b, c;    // cov2d after dilation
  float a0, c0;     // before dilation
  float det, det0;
  float radf;       // 3-sigma radius in pixels (integer-valued)
  int   geom_ok;    // 1 once centre / conic / radius are valid (a false return then only means "covers no tile")
};

// Tile bounding box of the 3-sigma circle (upstream rule): fills o.tmin/tmax/ntiles; false (and radius = 0, ntiles = 0)
// when it covers no tile.  Separate from project_one because the pixel-velocity model re-centres the SAME splat
// per sub-pose (xy + tau * pixel velocity) and only this part changes.
GS_HD bool tile_bounds(float x, float y, float radf, int tiles_x, int tiles_y, Proj& o) {
  const float inv_tile = 1.0f / (float)K::kTile;
  float tcx = x * inv_tile, tcy = y * inv_tile, tr = radf * inv_tile;
  int x0 = (int)(tcx - tr), x1 = (int)(tcx + tr + 1.0f);
  int y0 = (int)(tcy - tr), y1 = (int)(tcy + tr + 1.0f);
  x0 = x0 < 0 ? 0 : (x0 > tiles_x ? tiles_x : x0);
  x1 = x1 < 0 ? 0 : (x1 > tiles_x ? tiles_x : x1);
  y0 = y0 < 0 ? 0 : (y0 > tiles_y ? tiles_y : y0);
  y1 = y1 < 0 ? 0 : (y1 > tiles_y ? tiles_y : y1);
  int area = (x1 - x0) * (y1 - y0);
  if (area <= 0) { o.radius = 0; o.ntiles = 0; o.tmin_x = o.tmin_y = o.tmax_x = o.tmax_y = 0; return false; }
  o.ntiles = area;
  o.tmin_x = x0; o.tmin_y = y0; o.tmax_x = x1; o.tmax_y = y1;
  return true;
}

// tile_bounds for a splat whose centre sweeps the segment (xa, ya) - (xb, yb) while the rolling shutter reads the frame
// out (exact per-row model, see project.hip / raster_rs.hip): the box of the 3-sigma circle dragged along the segment
GS_HD bool tile_bounds_swept(float xa, float ya, float xb, float yb, float radf, int tiles_x, int tiles_y, Proj& o) {
  const float inv_tile = 1.0f / (float)K::kTile;
  const float xlo = fminf(xa, xb) * inv_tile, xhi = fmaxf(xa, xb) * inv_tile;
  const float ylo = fminf(ya, yb) * inv_tile, yhi = fmaxf(ya, yb) * inv_tile, tr = radf * inv_tile;
  int x0 = (int)(xlo - tr), x1 = (int)(xhi + tr + 1.0f);
  int y0 = (int)(ylo - tr), y1 = (int)(yhi + tr + 1.0f);
  x0 = x0 < 0 ? 0 : (x0 > tiles_x ? tiles_x : x0);
  x1 = x1 < 0 ? 0 : (x1 > tiles_x ? tiles_x : x1);
  y0 = y0 < 0 ? 0 : (y0 > tiles_y ? tiles_y : y0);
  y1 = y1 < 0 ? 0 : (y1 > tiles_y ? tiles_y : y1);
  int area = (x1 - x0) * (y1 - y0);
  if (area <= 0) { o.radius = 0; o.ntiles = 0; o.tmin_x = o.tmin_y = o.tmax_x = o.tmax_y = 0; return false; }
  o.ntiles = area;
  o.tmin_x = x0; o.tmin_y = y0; o.tmax_x = x1; o.tmax_y = y1;
  return true;
}

// Project one Gaussian with covariance c3 (6) under viewmat V (row-major 4x4,
// rows 0..2 used).  Returns false when culled (near plane / singular / no tile).
GS_HD bool project_one(const float mean[3], const float c3[6], const float* V,
                       float fx, float fy, float cx, float cy, int img_w, int img_h,
                       int tiles_x, int tiles_y, float clip, Proj& o, ProjCtx& k) {
  o.radius = 0; o.ntiles = 0; o.tmin_x = o.tmin_y = o.tmax_x = o.tmax_y = 0;
  k.geom_ok = 0; k.radf = 0.f;
  float px = ((V[0] * mean[0] + V[1] * mean[1]) + V[2] * mean[2]) + V[3];
  float py = ((V[4] * mean[0] + V[5] * mean[1]) + V[6] * mean[2]) + V[7];
  float pz = ((V[8] * mean[0] + V[9] * mean[1]) + V[10] * mean[2]) + V[11];
  k.pc[0] = px; k.pc[1] = py; k.pc[2] = pz;
  o.depth = pz;
  if (!(pz > clip)) return false;
  float rz = 1.0f / pz;
  k.rz = rz;
  float lim_x = K::kFovLimit * (0.5f * (float)img_w / fx);
  float lim_y = K::kFovLimit * (0.5f * (float)img_h / fy);
  float xz = px * rz, yz = py * rz;
  k.clamp_x = xz > lim_x ? 1 : (xz < -lim_x ? -1 : 0);
  k.clamp_y = yz > lim_y ? 1 : (yz < -lim_y ? -1 : 0);
  float tx = pz * fminf(lim_x, fmaxf(-lim_x, xz));
  float ty = pz * fminf(lim_y, fmaxf(-lim_y, yz));
  k.tx = tx; k.ty = ty;
  float rz2 = rz * rz;
  float J00 = fx * rz, J02 = -(fx * tx) * rz2;
  float J11 = fy * rz, J12 = -(fy * ty) * rz2;
  k.J00 = J00; k.J02 = J02; k.J11 = J11; k.J12 = J12;
  // T = J * W   (W = V[0:3,0:3])
  float* T = k.T;
  T[0] = J00 * V[0] + J02 * V[8];
  T[1] = J00 * V[1] + J02 * V[9];
  T[2] = J00 * V[2] + J02 * V[10];
  T[3] = J11 * V[4] + J12 * V[8];
  T[4] = J11 * V[5] + J12 * V[9];
  T[5] = J11 * V[6] + J12 * V[10];
  // U = T * Sigma (2x3)
  float U0 = (T[0] * c3[0] + T[1] * c3[1]) + T[2] * c3[2];
  float U1 = (T[0] * c3[1] + T[1] * c3[3]) + T[2] * c3[4];
  float U2 = (T[0] * c3[2] + T[1] * c3[4]) + T[2] * c3[5];
  float U3 = (T[3] * c3[0] + T[4] * c3[1]) + T[5] * c3[2];
  float U4 = (T[3] * c3[1] + T[4] * c3[3]) + T[5] * c3[4];
  float U5 = (T[3] * c3[2] + T[4] * c3[4]) + T[5] * c3[5];
  float a0 = (U0 * T[0] + U1 * T[1]) + U2 * T[2];
  float b  = (U0 * T[3] + U1 * T[4]) + U2 * T[5];
  float c0 = (U3 * T[3] + U4 * T[4]) + U5 * T[5];
  float det0 = a0 * c0 - b * b;
  float a = a0 + K::kDilation, c = c0 + K::kDilation;
  float det = a * c - b * b;
  k.a = a; k.b = b; k.c = c; k.a0 = a0; k.c0 = c0; k.det = det; k.det0 = det0;
  if (det == 0.0f) return false;
  o.comp = sqrtf(fmaxf(0.0f, det0 / det));
  float inv_det = 1.0f / det;
  o.conic_x = c * inv_det; o.conic_y = -b * inv_det; o.conic_z = a * inv_det;
  float mid = 0.5f * (a + c);
  float lam = mid + sqrtf(fmaxf(K::kMinEigDisc, mid * mid - det));
  float radf = ceilf(K::kRadiusSig * sqrtf(lam));
  int radius = (int)radf;
  o.x = (fx * px) * rz + cx;
  o.y = (fy * py) * rz + cy;
  o.radius = radius;
  k.radf = radf; k.geom_ok = 1;
  return tile_bounds(o.x, o.y, radf, tiles_x, tiles_y, o);
}

// ---- projection backward -----------------------------------------------------
// Gradients flowing in: v_xy[2], v_depth, v_conic[3], v_comp.
// Out (accumulated with =, caller sums): v_mean[3], v_cov3d[6] (for upper-triangle
// parametrisation: off-diagonals carry the sum of both symmetric entries),
// v_V[12] (rows 0..2 of the viewmat).
// v_pc_extra (nullable): an additional gradient on the camera-space mean (the pixel-velocity model's).
// upstream_clamp_grad: gsplat 0.1.11 back-propagates through the fov clamp of x/z, y/z as if it were inactive: a
// straight-through rule on tx, ty (v_px += v_tx whether or not the clamp is active).  Round 5 also
// built and measured the other reading — the VJP of the UNCLAMPED EWA projection (J rebuilt from px, py) — and dropped
// it: no better evidence, and the end-to-end pose recovery it costs is worse; false = the true derivative.
GS_HD void project_one_bwd(const float mean[3], const float c3[6], const float* V,
                           float fx, float fy, const ProjCtx& k, float comp,
                           const float v_xy[2], float v_depth, const float v_conic[3], float v_comp,
                           float v_mean[3], float v_c3[6], float v_V[12],
                           const float* v_pc_extra = nullptr, bool upstream_clamp_grad = false) {
  const float a = k.a, b = k.b, c = k.c, det = k.det;
  const float inv_det = 1.0f / det, inv_det2 = inv_det * inv_det;
  // conic -> (a,b,c)
  float v0 = v_conic[0], v1 = v_conic[1], v2 = v_conic[2];
  float v_a = (-c * c * v0 + b * c * v1 - b * b * v2) * inv_det2;
  float v_c = (-b * b * v0 + a * b * v1 - a * a * v2) * inv_det2;
  float v_b = (2.f * b * c * v0 - (a * c + b * b) * v1 + 2.f * a * b * v2) * inv_det2;
  // compensation = sqrt(max(0, det0/det))
  if (comp > 0.0f && v_comp != 0.0f) {
    float v_r = v_comp * 0.5f / comp;
    float det0 = k.det0;
    v_a += v_r * (k.c0 * inv_det - det0 * c * inv_det2);
    v_c += v_r * (k.a0 * inv_det - det0 * a * inv_det2);
    v_b += v_r * (-2.f * b * inv_det + det0 * 2.f * b * inv_det2);
  }
  // cov2d = T Sigma T^T ; Gc = [[v_a, v_b/2],[v_b/2, v_c]]
  const float* T = k.T;
  float g00 = v_a, g01 = 0.5f * v_b, g11 = v_c;
  // v_Sigma = T^T Gc T  (3x3 symmetric) -> upper-triangle parametrisation
  float GT0 = g00 * T[0] + g01 * T[3], GT1 = g00 * T[1] + g01 * T[4], GT2 = g00 * T[2] + g01 * T[5];
  float GT3 = g01 * T[0] + g11 * T[3], GT4 = g01 * T[1] + g11 * T[4], GT5 = g01 * T[2] + g11 * T[5];
  float vS00 = T[0] * GT0 + T[3] * GT3;
  float vS01 = T[0] * GT1 + T[3] * GT4;
  float vS02 = T[0] * GT2 + T[3] * GT5;
  float vS11 = T[1] * GT1 + T[4] * GT4;
  float vS12 = T[1] * GT2 + T[4] * GT5;
  float vS22 = T[2] * GT2 + T[5] * GT5;
  v_c3[0] = vS00; v_c3[1] = 2.f * vS01; v_c3[2] = 2.f * vS02;
  v_c3[3] = vS11; v_c3[4] = 2.f * vS12; v_c3[5] = vS22;
  // v_T = 2 Gc T Sigma   (2x3)
  float S00 = c3[0], S01 = c3[1], S02 = c3[2], S11 = c3[3], S12 = c3[4], S22 = c3[5];
  float vT0 = 2.f * (GT0 * S00 + GT1 * S01 + GT2 * S02);
  float vT1 = 2.f * (GT0 * S01 + GT1 * S11 + GT2 * S12);
  float vT2 = 2.f * (GT0 * S02 + GT1 * S12 + GT2 * S22);
  float vT3 = 2.f * (GT3 * S00 + GT4 * S01 + GT5 * S02);
  float vT4 = 2.f * (GT3 * S01 + GT4 * S11 + GT5 * S12);
  float vT5 = 2.f * (GT3 * S02 + GT4 * S12 + GT5 * S22);
  // T = J W : v_J = v_T W^T ; v_W = J^T v_T
  float vJ00 = vT0 * V[0] + vT1 * V[1] + vT2 * V[2];
  float vJ02 = vT0 * V[8] + vT1 * V[9] + vT2 * V[10];
  float vJ11 = vT3 * V[4] + vT4 * V[5] + vT5 * V[6];
  float vJ12 = vT3 * V[8] + vT4 * V[9] + vT5 * V[10];
  float vW[9];
  vW[0] = k.J00 * vT0; vW[1] = k.J00 * vT1; vW[2] = k.J00 * vT2;
  vW[3] = k.J11 * vT3; vW[4] = k.J11 * vT4; vW[5] = k.J11 * vT5;
  vW[6] = k.J02 * vT0 + k.J12 * vT3; vW[7] = k.J02 * vT1 + k.J12 * vT4; vW[8] = k.J02 * vT2 + k.J12 * vT5;
  // J(pc)
  const float px = k.pc[0], py = k.pc[1], pz = k.pc[2], rz = k.rz;
  const float rz2 = rz * rz, rz3 = rz2 * rz;
  float v_tx = -fx * rz2 * vJ02;
  float v_ty = -fy * rz2 * vJ12;
  float v_pz = -fx * rz2 * vJ00 - fy * rz2 * vJ11 + 2.f * fx * k.tx * rz3 * vJ02 + 2.f * fy * k.ty * rz3 * vJ12;
  float v_px = 0.f, v_py = 0.f;
  if (k.clamp_x == 0 || upstream_clamp_grad) v_px += v_tx; else v_pz += v_tx * (k.tx * rz);  // tx = (+-lim) * z
  if (k.clamp_y == 0 || upstream_clamp_grad) v_py += v_ty; else v_pz += v_ty * (k.ty * rz);
  // pixel centre + depth
  v_px += fx * rz * v_xy[0];
  v_py += fy * rz * v_xy[1];
  v_pz += -(fx * px * rz2 * v_xy[0] + fy * py * rz2 * v_xy[1]) + v_depth;
  if (v_pc_extra) { v_px += v_pc_extra[0]; v_py += v_pc_extra[1]; v_pz += v_pc_extra[2]; }
  // pc = W mean + t
  v_mean[0] = V[0] * v_px + V[4] * v_py + V[8] * v_pz;
  v_mean[1] = V[1] * v_px + V[5] * v_py + V[9] * v_pz;
  v_mean[2] = V[2] * v_px + V[6] * v_py + V[10] * v_pz;
  v_V[0] = vW[0] + v_px * mean[0]; v_V[1] = vW[1] + v_px * mean[1]; v_V[2]  = vW[2] + v_px * mean[2]; v_V[3]  = v_px;
  v_V[4] = vW[3] + v_py * mean[0]; v_V[5] = vW[4] + v_py * mean[1]; v_V[6]  = vW[5] + v_py * mean[2]; v_V[7]  = v_py;
  v_V[8] = vW[6] + v_pz * mean[0]; v_V[9] = vW[7] + v_pz * mean[1]; v_V[10] = vW[8] + v_pz * mean[2]; v_V[11] = v_pz;
}

// ---- pixel velocity (the paper's first-order blur / rolling-shutter model) ------------------
// A static point seen from a camera moving with body twist (lin, ang) (camera frame) moves in camera space with
// velocity u = -(ang x pc + lin); its pixel moves with  pv = J u,  J = d(pixel)/d(pc) of the pinhole projection,
// evaluated where the covariance projection evaluates its own Jacobian: at (jx, jy, z) = the camera-space centre with
// x/z and y/z clamped to the fov guard band (project_one's tx, ty; inside the band jx = pc[0], jy = pc[1] exactly).
// x/z is unbounded at grazing angles: with the unclamped J a large near splat whose centre lies beside the image would
// be dragged across it at thousands of pixels per frame; round 3 culled such Gaussians altogether (and a floor whose
// centre is out of band vanished) — now they move with the velocity of the band edge, bounded by 1/z > 1/clip.
// The splat rendered at time tau sits at xy + tau * pv; covariance, opacity, colour and depth order are those of the
// mid-exposure pose.
GS_HD void pixel_velocity(const float pc[3], float jx, float jy, float rz, float fx, float fy, const float lin[3],
                          const float ang[3], float pv[2]) {
  const float ux = -(ang[1] * pc[2] - ang[2] * pc[1]) - lin[0];
  const float uy = -(ang[2] * pc[0] - ang[0] * pc[2]) - lin[1];
  const float uz = -(ang[0] * pc[1] - ang[1] * pc[0]) - lin[2];
  const float rz2 = rz * rz;
  pv[0] = (fx * rz) * ux - ((fx * jx) * rz2) * uz;
  pv[1] = (fy * rz) * uy - ((fy * jy) * rz2) * uz;
}

// VJP of pixel_velocity: v_pv[2] -> v_pc[3] (=), v_lin[3] (=), v_ang[3] (=).  clamp_x / clamp_y: which side of the guard
// band jx / jy sit on (0: inside, jx = pc[0]); a clamped jx = +-lim * z passes its gradient to z (or, with
// upstream_clamp_grad, to x as if the clamp were inactive — the rule of project_one_bwd).
GS_HD void pixel_velocity_bwd(const float pc[3], float jx, float jy, int clamp_x, int clamp_y, bool upstream_clamp_grad,
                              float rz, float fx, float fy, const float lin[3], const float ang[3],
                              const float v_pv[2], float v_pc[3], float v_lin[3], float v_ang[3]) {
  const float ux = -(ang[1] * pc[2] - ang[2] * pc[1]) - lin[0];
  const float uy = -(ang[2] * pc[0] - ang[0] * pc[2]) - lin[1];
  const float uz = -(ang[0] * pc[1] - ang[1] * pc[0]) - lin[2];
  const float rz2 = rz * rz, rz3 = rz2 * rz;
  const float gx = v_pv[0], gy = v_pv[1];
  const float vu[3] = {fx * rz * gx, fy * rz * gy, -(fx * jx * rz2 * gx + fy * jy * rz2 * gy)};
  // through J(jx, jy, z)
  const float v_jx = -(fx * rz2 * uz) * gx, v_jy = -(fy * rz2 * uz) * gy;
  v_pc[0] = 0.f; v_pc[1] = 0.f;
  v_pc[2] = gx * (-fx * rz2 * ux + 2.f * fx * jx * rz3 * uz) + gy * (-fy * rz2 * uy + 2.f * fy * jy * rz3 * uz);
  if (clamp_x == 0 || upstream_clamp_grad) v_pc[0] += v_jx; else v_pc[2] += v_jx * (jx * rz);    // jx = (+-lim) * z
  if (clamp_y == 0 || upstream_clamp_grad) v_pc[1] += v_jy; else v_pc[2] += v_jy * (jy * rz);
  // through u = -ang x pc - lin :  v_pc += ang x v_u ;  v_ang = v_u x pc ;  v_lin = -v_u
  v_pc[0] += ang[1] * vu[2] - ang[2] * vu[1];
  v_pc[1] += ang[2] * vu[0] - ang[0] * vu[2];
  v_pc[2] += ang[0] * vu[1] - ang[1] * vu[0];
  v_ang[0] = vu[1] * pc[2] - vu[2] * pc[1];
  v_ang[1] = vu[2] * pc[0] - vu[0] * pc[2];
  v_ang[2] = vu[0] * pc[1] - vu[1] * pc[0];
  v_lin[0] = -vu[0]; v_lin[1] = -vu[1]; v_lin[2] = -vu[2];
}

// cov3d (upper-triangle grads as produced above) -> scale, quat grads.
// raw_quat_grad: gsplat 0.1.11 returns the gradient w.r.t. the (assumed unit) quaternion without the projection
// through q/|q| (DESIGN.md §1 deviation 3); false = gradient of the normalising kernel.
GS_HD void cov3d_bwd(const float s[3], float glob, const float q[4], const float v_c3[6],
                     float v_s[3], float v_q[4], bool raw_quat_grad = false) {
  float R[9], qn[4], inv;
  quat_to_rotmat(q, R, qn, &inv);
  float M[9], c3[6];
  scale_rot_to_cov3d(s, glob, R, M, c3);
  // symmetric v_Sigma from upper-triangle parametrisation
  float S00 = v_c3[0], S01 = 0.5f * v_c3[1], S02 = 0.5f * v_c3[2];
  float S11 = v_c3[3], S12 = 0.5f * v_c3[4], S22 = v_c3[5];
  // v_M = 2 v_Sigma M
  float vM[9];
  vM[0] = 2.f * (S00 * M[0] + S01 * M[3] + S02 * M[6]);
  vM[1] = 2.f * (S00 * M[1] + S01 * M[4] + S02 * M[7]);
  vM[2] = 2.f * (S00 * M[2] + S01 * M[5] + S02 * M[8]);
  vM[3] = 2.f * (S01 * M[0] + S11 * M[3] + S12 * M[6]);
  vM[4] = 2.f * (S01 * M[1] + S11 * M[4] + S12 * M[7]);
  vM[5] = 2.f * (S01 * M[2] + S11 * M[5] + S12 * M[8]);
  vM[6] = 2.f * (S02 * M[0] + S12 * M[3] + S22 * M[6]);
  vM[7] = 2.f * (S02 * M[1] + S12 * M[4] + S22 * M[7]);
  vM[8] = 2.f * (S02 * M[2] + S12 * M[5] + S22 * M[8]);
  v_s[0] = glob * (vM[0] * R[0] + vM[3] * R[3] + vM[6] * R[6]);
  v_s[1] = glob * (vM[1] * R[1] + vM[4] * R[4] + vM[7] * R[7]);
  v_s[2] = glob * (vM[2] * R[2] + vM[5] * R[5] + vM[8] * R[8]);
  float s0 = glob * s[0], s1 = glob * s[1], s2 = glob * s[2];
  float vR[9] = {vM[0] * s0, vM[1] * s1, vM[2] * s2, vM[3] * s0, vM[4] * s1, vM[5] * s2,
                 vM[6] * s0, vM[7] * s1, vM[8] * s2};
  float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
  float g[4];
  g[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  g[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
  g[2] = 2.f * (x * (vR[3] + vR[1]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
  g[3] = 2.f * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
  if (raw_quat_grad) { v_q[0] = g[0]; v_q[1] = g[1]; v_q[2] = g[2]; v_q[3] = g[3]; return; }
  // through the normalisation q/|q|
  float dotp = g[0] * w + g[1] * x + g[2] * y + g[3] * z;
  v_q[0] = (g[0] - w * dotp) * inv;
  v_q[1] = (g[1] - x * dotp) * inv;
  v_q[2] = (g[2] - y * dotp) * inv;
  v_q[3] = (g[3] - z * dotp) * inv;
}

// ---- the covariance chain in a chosen precision (round 3: needle Gaussians) -----------------------------------------
// The VJP  v_conic -> v_cov2d -> v_cov3d -> (v_scale, v_quat)  cancels catastrophically for a splat whose 2-D
// covariance is ill conditioned (a needle: det = a c - b^2 is 1e-3 .. 1e-5 of a c): in fp32 the gradient along the
// long axis — a small difference of large terms in every basis but the eigenbasis — comes out percent-level wrong
// (found by tests/fuzz_paths.py in round 2, profiles/r02_run28_fuzz.log).  The functions below restate that chain
// (same formulas as project_one / project_one_bwd / cov3d_bwd above) on a scalar type S; the kernels run them in
// double for the Gaussians that need it, starting again from the fp32 PARAMETERS (the fp32 cov2d itself carries a
// relative error of eps * cond, so the chain must not start from it).
template <typename S> struct ProjCtxT {
  S pc[3], rz, tx, ty;
  int clamp_x, clamp_y;
  S J00, J02, J11, J12, T[6], a, b, c, a0, c0, det, det0;
};

template <typename S> GS_HD S gs_sqrt_t(S x);
template <> GS_HD float gs_sqrt_t<float>(float x) { return sqrtf(x); }
template <> GS_HD double gs_sqrt_t<double>(double x) { return ::sqrt(x); }

template <typename S>
GS_HD void quat_to_rotmat_t(const float q[4], S R[9], S qn[4], S* inv_norm) {
  const S q0 = (S)q[0], q1 = (S)q[1], q2 = (S)q[2], q3 = (S)q[3];
  const S inv = S(1) / gs_sqrt_t<S>(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
  const S w = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; *inv_norm = inv;
  R[0] = S(1) - S(2) * (y * y + z * z); R[1] = S(2) * (x * y - w * z);        R[2] = S(2) * (x * z + w * y);
  R[3] = S(2) * (x * y + w * z);        R[4] = S(1) - S(2) * (x * x + z * z); R[5] = S(2) * (y * z - w * x);
  R[6] = S(2) * (x * z - w * y);        R[7] = S(2) * (y * z + w * x);        R[8] = S(1) - S(2) * (x * x + y * y);
}

template <typename S>
GS_HD void scale_rot_to_cov3d_t(const float s[3], float glob, const S R[9], S M[9], S c[6]) {
  const S s0 = (S)glob * (S)s[0], s1 = (S)glob * (S)s[1], s2 = (S)glob * (S)s[2];
  M[0] = R[0] * s0; M[1] = R[1] * s1; M[2] = R[2] * s2;
  M[3] = R[3] * s0; M[4] = R[4] * s1; M[5] = R[5] * s2;
  M[6] = R[6] * s0; M[7] = R[7] * s1; M[8] = R[8] * s2;
  c[0] = (M[0] * M[0] + M[1] * M[1]) + M[2] * M[2];
  c[1] = (M[0] * M[3] + M[1] * M[4]) + M[2] * M[5];
  c[2] = (M[0] * M[6] + M[1] * M[7]) + M[2] * M[8];
  c[3] = (M[3] * M[3] + M[4] * M[4]) + M[5] * M[5];
  c[4] = (M[3] * M[6] + M[4] * M[7]) + M[5] * M[8];
  c[5] = (M[6] * M[6] + M[7] * M[7]) + M[8] * M[8];
}

// geometry of project_one without its culling decisions (the caller took those in fp32): camera-space mean, the
// fov-clamped Jacobian, T = J W, cov2d before / after the dilation, both determinants
template <typename S>
GS_HD void project_ctx_t(const float mean[3], const S c3[6], const float* V, float fx_, float fy_, int img_w, int img_h,
                         ProjCtxT<S>& k) {
  const S fx = (S)fx_, fy = (S)fy_;
  const S m0 = (S)mean[0], m1 = (S)mean[1], m2 = (S)mean[2];
  const S px = (((S)V[0] * m0 + (S)V[1] * m1) + (S)V[2] * m2) + (S)V[3];
  const S py = (((S)V[4] * m0 + (S)V[5] * m1) + (S)V[6] * m2) + (S)V[7];
  const S pz = (((S)V[8] * m0 + (S)V[9] * m1) + (S)V[10] * m2) + (S)V[11];
  k.pc[0] = px; k.pc[1] = py; k.pc[2] = pz;
  const S rz = S(1) / pz;
  k.rz = rz;
  const S lim_x = (S)K::kFovLimit * (S(0.5) * (S)img_w / fx), lim_y = (S)K::kFovLimit * (S(0.5) * (S)img_h / fy);
  const S xz = px * rz, yz = py * rz;
  k.clamp_x = xz > lim_x ? 1 : (xz < -lim_x ? -1 : 0);
  k.clamp_y = yz > lim_y ? 1 : (yz < -lim_y ? -1 : 0);
  const S tx = pz * (xz > lim_x ? lim_x : (xz < -lim_x ? -lim_x : xz));
  const S ty = pz * (yz > lim_y ? lim_y : (yz < -lim_y ? -lim_y : yz));
  k.tx = tx; k.ty = ty;
  const S rz2 = rz * rz;
  k.J00 = fx * rz; k.J02 = -(fx * tx) * rz2; k.J11 = fy * rz; k.J12 = -(fy * ty) * rz2;
  S* T = k.T;
  T[0] = k.J00 * (S)V[0] + k.J02 * (S)V[8];
  T[1] = k.J00 * (S)V[1] + k.J02 * (S)V[9];
  T[2] = k.J00 * (S)V[2] + k.J02 * (S)V[10];
  T[3] = k.J11 * (S)V[4] + k.J12 * (S)V[8];
  T[4] = k.J11 * (S)V[5] + k.J12 * (S)V[9];
  T[5] = k.J11 * (S)V[6] + k.J12 * (S)V[10];
  const S U0 = (T[0] * c3[0] + T[1] * c3[1]) + T[2] * c3[2];
  const S U1 = (T[0] * c3[1] + T[1] * c3[3]) + T[2] * c3[4];
  const S U2 = (T[0] * c3[2] + T[1] * c3[4]) + T[2] * c3[5];
  const S U3 = (T[3] * c3[0] + T[4] * c3[1]) + T[5] * c3[2];
  const S U4 = (T[3] * c3[1] + T[4] * c3[3]) + T[5] * c3[4];
  const S U5 = (T[3] * c3[2] + T[4] * c3[4]) + T[5] * c3[5];
  k.a0 = (U0 * T[0] + U1 * T[1]) + U2 * T[2];
  k.b = (U0 * T[3] + U1 * T[4]) + U2 * T[5];
  k.c0 = (U3 * T[3] + U4 * T[4]) + U5 * T[5];
  k.det0 = k.a0 * k.c0 - k.b * k.b;
  k.a = k.a0 + (S)K::kDilation; k.c = k.c0 + (S)K::kDilation;
  k.det = k.a * k.c - k.b * k.b;
}

// project_one_bwd on scalar type S (v_mean / v_c3 / v_V are ASSIGNED)
template <typename S>
GS_HD void project_one_bwd_t(const float mean[3], const S c3[6], const float* V, float fx_, float fy_,
                             const ProjCtxT<S>& k, S comp, const S v_xy[2], S v_depth, const S v_conic[3], S v_comp,
                             S v_mean[3], S v_c3[6], S v_V[12], const S* v_pc_extra, bool upstream_clamp_grad) {
  const S fx = (S)fx_, fy = (S)fy_;
  const S a = k.a, b = k.b, c = k.c, det = k.det;
  const S inv_det = S(1) / det, inv_det2 = inv_det * inv_det;
  const S v0 = v_conic[0], v1 = v_conic[1], v2 = v_conic[2];
  S v_a = (-c * c * v0 + b * c * v1 - b * b * v2) * inv_det2;
  S v_c = (-b * b * v0 + a * b * v1 - a * a * v2) * inv_det2;
  S v_b = (S(2) * b * c * v0 - (a * c + b * b) * v1 + S(2) * a * b * v2) * inv_det2;
  if (comp > S(0) && v_comp != S(0)) {
    const S v_r = v_comp * S(0.5) / comp;
    const S det0 = k.det0;
    v_a += v_r * (k.c0 * inv_det - det0 * c * inv_det2);
    v_c += v_r * (k.a0 * inv_det - det0 * a * inv_det2);
    v_b += v_r * (-S(2) * b * inv_det + det0 * S(2) * b * inv_det2);
  }
  const S* T = k.T;
  const S g00 = v_a, g01 = S(0.5) * v_b, g11 = v_c;
  const S GT0 = g00 * T[0] + g01 * T[3], GT1 = g00 * T[1] + g01 * T[4], GT2 = g00 * T[2] + g01 * T[5];
  const S GT3 = g01 * T[0] + g11 * T[3], GT4 = g01 * T[1] + g11 * T[4], GT5 = g01 * T[2] + g11 * T[5];
  v_c3[0] = T[0] * GT0 + T[3] * GT3;
  v_c3[1] = S(2) * (T[0] * GT1 + T[3] * GT4);
  v_c3[2] = S(2) * (T[0] * GT2 + T[3] * GT5);
  v_c3[3] = T[1] * GT1 + T[4] * GT4;
  v_c3[4] = S(2) * (T[1] * GT2 + T[4] * GT5);
  v_c3[5] = T[2] * GT2 + T[5] * GT5;
  const S S00 = c3[0], S01 = c3[1], S02 = c3[2], S11 = c3[3], S12 = c3[4], S22 = c3[5];
  const S vT0 = S(2) * (GT0 * S00 + GT1 * S01 + GT2 * S02);
  const S vT1 = S(2) * (GT0 * S01 + GT1 * S11 + GT2 * S12);
  const S vT2 = S(2) * (GT0 * S02 + GT1 * S12 + GT2 * S22);
  const S vT3 = S(2) * (GT3 * S00 + GT4 * S01 + GT5 * S02);
  const S vT4 = S(2) * (GT3 * S01 + GT4 * S11 + GT5 * S12);
  const S vT5 = S(2) * (GT3 * S02 + GT4 * S12 + GT5 * S22);
  const S vJ00 = vT0 * (S)V[0] + vT1 * (S)V[1] + vT2 * (S)V[2];
  const S vJ02 = vT0 * (S)V[8] + vT1 * (S)V[9] + vT2 * (S)V[10];
  const S vJ11 = vT3 * (S)V[4] + vT4 * (S)V[5] + vT5 * (S)V[6];
  const S vJ12 = vT3 * (S)V[8] + vT4 * (S)V[9] + vT5 * (S)V[10];
  S vW[9];
  vW[0] = k.J00 * vT0; vW[1] = k.J00 * vT1; vW[2] = k.J00 * vT2;
  vW[3] = k.J11 * vT3; vW[4] = k.J11 * vT4; vW[5] = k.J11 * vT5;
  vW[6] = k.J02 * vT0 + k.J12 * vT3; vW[7] = k.J02 * vT1 + k.J12 * vT4; vW[8] = k.J02 * vT2 + k.J12 * vT5;
  const S px = k.pc[0], py = k.pc[1], rz = k.rz;
  const S rz2 = rz * rz, rz3 = rz2 * rz;
  const S v_tx = -fx * rz2 * vJ02, v_ty = -fy * rz2 * vJ12;
  S v_pz = -fx * rz2 * vJ00 - fy * rz2 * vJ11 + S(2) * fx * k.tx * rz3 * vJ02 + S(2) * fy * k.ty * rz3 * vJ12;
  S v_px = S(0), v_py = S(0);
  if (k.clamp_x == 0 || upstream_clamp_grad) v_px += v_tx; else v_pz += v_tx * (k.tx * rz);
  if (k.clamp_y == 0 || upstream_clamp_grad) v_py += v_ty; else v_pz += v_ty * (k.ty * rz);
  v_px += fx * rz * v_xy[0];
  v_py += fy * rz * v_xy[1];
  v_pz += -(fx * px * rz2 * v_xy[0] + fy * py * rz2 * v_xy[1]) + v_depth;
  if (v_pc_extra) { v_px += v_pc_extra[0]; v_py += v_pc_extra[1]; v_pz += v_pc_extra[2]; }
  v_mean[0] = (S)V[0] * v_px + (S)V[4] * v_py + (S)V[8] * v_pz;
  v_mean[1] = (S)V[1] * v_px + (S)V[5] * v_py + (S)V[9] * v_pz;
  v_mean[2] = (S)V[2] * v_px + (S)V[6] * v_py + (S)V[10] * v_pz;
  const S m0 = (S)mean[0], m1 = (S)mean[1], m2 = (S)mean[2];
  v_V[0] = vW[0] + v_px * m0; v_V[1] = vW[1] + v_px * m1; v_V[2]  = vW[2] + v_px * m2; v_V[3]  = v_px;
  v_V[4] = vW[3] + v_py * m0; v_V[5] = vW[4] + v_py * m1; v_V[6]  = vW[5] + v_py * m2; v_V[7]  = v_py;
  v_V[8] = vW[6] + v_pz * m0; v_V[9] = vW[7] + v_pz * m1; v_V[10] = vW[8] + v_pz * m2; v_V[11] = v_pz;
}

// cov3d_bwd on scalar type S
template <typename S>
GS_HD void cov3d_bwd_t(const float s[3], float glob_, const float q[4], const S v_c3[6], S v_s[3], S v_q[4],
                       bool raw_quat_grad) {
  S R[9], qn[4], inv, M[9], c3[6];
  quat_to_rotmat_t<S>(q, R, qn, &inv);
  scale_rot_to_cov3d_t<S>(s, glob_, R, M, c3);
  const S glob = (S)glob_;
  const S S00 = v_c3[0], S01 = S(0.5) * v_c3[1], S02 = S(0.5) * v_c3[2];
  const S S11 = v_c3[3], S12 = S(0.5) * v_c3[4], S22 = v_c3[5];
  S vM[9];
  vM[0] = S(2) * (S00 * M[0] + S01 * M[3] + S02 * M[6]);
  vM[1] = S(2) * (S00 * M[1] + S01 * M[4] + S02 * M[7]);
  vM[2] = S(2) * (S00 * M[2] + S01 * M[5] + S02 * M[8]);
  vM[3] = S(2) * (S01 * M[0] + S11 * M[3] + S12 * M[6]);
  vM[4] = S(2) * (S01 * M[1] + S11 * M[4] + S12 * M[7]);
  vM[5] = S(2) * (S01 * M[2] + S11 * M[5] + S12 * M[8]);
  vM[6] = S(2) * (S02 * M[0] + S12 * M[3] + S22 * M[6]);
  vM[7] = S(2) * (S02 * M[1] + S12 * M[4] + S22 * M[7]);
  vM[8] = S(2) * (S02 * M[2] + S12 * M[5] + S22 * M[8]);
  v_s[0] = glob * (vM[0] * R[0] + vM[3] * R[3] + vM[6] * R[6]);
  v_s[1] = glob * (vM[1] * R[1] + vM[4] * R[4] + vM[7] * R[7]);
  v_s[2] = glob * (vM[2] * R[2] + vM[5] * R[5] + vM[8] * R[8]);
  const S s0 = glob * (S)s[0], s1 = glob * (S)s[1], s2 = glob * (S)s[2];
  const S vR[9] = {vM[0] * s0, vM[1] * s1, vM[2] * s2, vM[3] * s0, vM[4] * s1, vM[5] * s2,
                   vM[6] * s0, vM[7] * s1, vM[8] * s2};
  const S w = qn[0], x = qn[1], y = qn[2], z = qn[3];
  S g[4];
  g[0] = S(2) * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  g[1] = S(2) * (-S(2) * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
  g[2] = S(2) * (x * (vR[3] + vR[1]) - S(2) * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
  g[3] = S(2) * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - S(2) * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
  if (raw_quat_grad) { v_q[0] = g[0]; v_q[1] = g[1]; v_q[2] = g[2]; v_q[3] = g[3]; return; }
  const S dotp = g[0] * w + g[1] * x + g[2] * y + g[3] * z;
  v_q[0] = (g[0] - w * dotp) * inv;
  v_q[1] = (g[1] - x * dotp) * inv;
  v_q[2] = (g[2] - y * dotp) * inv;
  v_q[3] = (g[3] - z * dotp) * inv;
}

// ---- spherical harmonics (real, degree <= 4) ---------------------------------
// basis values for unit direction (x,y,z); nb = (deg+1)^2.  Sloan-style
// recurrences with the usual 3DGS sign convention.
GS_HD void sh_basis(int deg, float x, float y, float z, float* B) {
  B[0] = 0.2820947917738781f;
  if (deg < 1) return;
  B[1] = -0.48860251190292f * y;
  B[2] = 0.48860251190292f * z;
  B[3] = -0.48860251190292f * x;
  if (deg < 2) return;
  float z2 = z * z;
  float fTmp0B = -1.092548430592079f * z;
  float fC1 = x * x - y * y, fS1 = 2.f * x * y;
  B[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
  B[7] = fTmp0B * x;
  B[5] = fTmp0B * y;
  B[8] = 0.5462742152960395f * fC1;
  B[4] = 0.5462742152960395f * fS1;
  if (deg < 3) return;
  float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
  float fTmp1B = 1.445305721320277f * z;
  float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
  B[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
  B[13] = fTmp0C * x;
  B[11] = fTmp0C * y;
  B[14] = fTmp1B * fC1;
  B[10] = fTmp1B * fS1;
  B[15] = -0.5900435899266435f * fC2;
  B[9]  = -0.5900435899266435f * fS2;
  if (deg < 4) return;
  float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
  float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
  float fTmp2B = -1.770130769779931f * z;
  float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
  B[20] = 1.984313483298443f * z * B[12] - 1.006230589874905f * B[6];
  B[21] = fTmp0D * x;
  B[19] = fTmp0D * y;
  B[22] = fTmp1C * fC1;
  B[18] = fTmp1C * fS1;
  B[23] = fTmp2B * fC2;
  B[17] = fTmp2B * fS2;
  B[24] = 0.6258357354491763f * fC3;
  B[16] = 0.6258357354491763f * fS3;
}

// ---- SE(3) screw interpolation ------------------------------------------------
// Camera-to-world at time t under constant body twist xi = (v, w), both in the
// (OpenCV) camera frame:  C(t) = C0 * Exp(t * xi)  =>  viewmat(t) = Exp(-t*xi) * viewmat0.
// Velocity frame convention: /root/reference/process_synthetic_inputs.py:157-165,
// /root/reference/render_video.py:100-115 (R_w2c @ velocity_w).
// Templated so that the backward kernel can push dual numbers through it.
template <typename S>
GS_HD void se3_exp(const S v[3], const S w[3], S E[12]) {
  // E = [R | V v] with R = I + A K + B K^2, V = I + B K + C K^2, K = [w]x
  S th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  S A, B, C;
  if (th2 < S(0.25f)) {
    // Series in theta^2 up to theta^10 (theta < 0.5 rad per sub-pose: every realistic exposure; truncation < 2e-12).
    // Round 6: the closed forms were taken from theta = 1e-4 on, and in fp32 (1 - cos theta) / theta^2 and
    // (1 - sin theta / theta) / theta^2 lose their digits to cancellation exactly in the range a blurred frame lives
    // in (theta = 7e-3: 3e-3 and 8e-3 relative).  The images hardly see it (B and C multiply second-order terms);
    // the VELOCITY gradients do — they are differences between sub-poses, sum_p t_p g_p with symmetric t_p, so the
    // first-order parts cancel and what is left carried an error of 9x the test bar (found by the camera-sequence test
    // of tests/test_gpu_round6.py; these are the gradients the velocity optimizer consumes, /root/reference/train.py:66).
    A = S(1.f) + th2 * (S(-1.f / 6.f) + th2 * (S(1.f / 120.f) + th2 * (S(-1.f / 5040.f) + th2 * (S(1.f / 362880.f) + th2 * S(-1.f / 39916800.f)))));
    B = S(0.5f) + th2 * (S(-1.f / 24.f) + th2 * (S(1.f / 720.f) + th2 * (S(-1.f / 40320.f) + th2 * (S(1.f / 3628800.f) + th2 * S(-1.f / 479001600.f)))));
    C = S(1.f / 6.f) + th2 * (S(-1.f / 120.f) + th2 * (S(1.f / 5040.f) + th2 * (S(-1.f / 362880.f) + th2 * (S(1.f / 39916800.f) + th2 * S(-1.f / 6227020800.f)))));
  } else {
    S th = sqrt(th2);
    S sh = sin(th * S(0.5f));
    A = sin(th) / th;
    B = S(2.f) * sh * sh / th2;          // (1 - cos theta) without the cancellation
    C = (S(1.f) - A) / th2;              // theta >= 0.5: theta - sin theta keeps six digits
  }
  S K[9] = {S(0.f), -w[2], w[1], w[2], S(0.f), -w[0], -w[1], w[0], S(0.f)};
  S K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      K2[i * 3 + j] = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
  for (int i = 0; i < 3; ++i) {
    S t = S(0.f);
    for (int j = 0; j < 3; ++j) {
      S I = S(i == j ? 1.f : 0.f);
      E[i * 4 + j] = I + A * K[i * 3 + j] + B * K2[i * 3 + j];
      S Vij = I + B * K[i * 3 + j] + C * K2[i * 3 + j];
      t = t + Vij * v[j];
    }
    E[i * 4 + 3] = t;
  }
}

// viewmat(t) rows 0..2 = Exp(-t xi) * viewmat0
template <typename S>
GS_HD void subpose_viewmat(const S V0[12], const S lin[3], const S ang[3], S t, S out[12]) {
  S v[3] = {-t * lin[0], -t * lin[1], -t * lin[2]};
  S w[3] = {-t * ang[0], -t * ang[1], -t * ang[2]};
  S E[12];
  se3_exp<S>(v, w, E);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      S acc = E[i * 4 + 0] * V0[0 * 4 + j] + E[i * 4 + 1] * V0[1 * 4 + j] + E[i * 4 + 2] * V0[2 * 4 + j];
      if (j == 3) acc = acc + E[i * 4 + 3];
      out[i * 4 + j] = acc;
    }
  }
}

// minimal forward-mode dual number with N tangents (used only on P<=~100 sub-poses)
template <int N>
struct Dual {
  float v;
  float d[N];
  GS_HD Dual() : v(0.f) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
  GS_HD explicit Dual(float x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
};
template <int N> GS_HD Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> GS_HD Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> GS_HD Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> GS_HD Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> GS_HD Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; float ib = 1.f / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> GS_HD bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> GS_HD Dual<N> sqrt(const Dual<N>& a) { Dual<N> r; r.v = sqrtf(a.v); float h = 0.5f / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * h; return r; }
template <int N> GS_HD Dual<N> sin(const Dual<N>& a) { Dual<N> r; r.v = sinf(a.v); float c = cosf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> GS_HD Dual<N> cos(const Dual<N>& a) { Dual<N> r; r.v = cosf(a.v); float s = -sinf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }

GS_HD float sqrt(float a) { return sqrtf(a); }
GS_HD float sin(float a) { return sinf(a); }
GS_HD float cos(float a) { return cosf(a); }

}  // namespace gs
