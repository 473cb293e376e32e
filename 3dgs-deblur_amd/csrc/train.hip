// train.hip — the step right after the hot path: splatfacto's image loss
//   L = (1 - lambda) * mean|gt - pred| + lambda * (1 - SSIM(pred, gt))
// with its gradient d L / d pred, and ONE multi-tensor Adam step over the parameter groups.
// Restates (absent fork sources) what nerfstudio 1.1.0's splatfacto `get_loss_dict` computes with
// pytorch_msssim's SSIM (11x11 Gaussian window, sigma 1.5, VALID convolution, C1 = 0.01^2, C2 = 0.03^2, mean over the
// map) and what torch.optim.Adam(eps=1e-15) does per parameter group (/root/reference/train.py:115-122 runs that
// trainer; the scale regularisation flag train.py:120 stays a torch one-liner on [N,3]).
//
// MI355X design.  torch runs this loss as nine depthwise conv2d launches plus their autograd graph, and the optimizer as
// six foreach-Adam passes over 59 floats per Gaussian.  Here:
//   * loss forward AND backward are two tile kernels.  K1: a block stages a (32+10) x (8+10) patch of both images in
//     LDS (interleaved RGB, coalesced rows), runs the separable 11-tap filter over the five moments (x, y, xx, yy, xy),
//     forms the SSIM map value s and its three partial derivatives (d s / d mu_x, d s / d E[xx], d s / d E[xy]) and
//     writes those three maps; K2: the transposed filter of the three maps (the same separable correlation, the
//     window is symmetric) recombined with pred / gt gives d L / d pred, the L1 sign term fused in.  The per-block
//     partial sums of s and |gt - pred| are reduced by one more tiny launch: deterministic, no atomics.
//   * Adam: one launch over the concatenated element space of up to 8 tensors, float4 loads, parameters, gradients and
//     both moments each touched exactly once: 28 bytes per element, the HBM floor of a dense Adam step.
#include "gs_common.h"

namespace gs {

constexpr int kWin = 11;
constexpr int kHalo = kWin - 1;
constexpr int kLT_W = 32, kLT_H = 8;                       // tile of window anchors (K1) / pixels (K2) per block
constexpr int kPatchW = kLT_W + kHalo, kPatchH = kLT_H + kHalo;   // 42 x 18
constexpr int kRowF = kPatchW * 3;                          // 126 floats per staged row (RGB interleaved)
constexpr int kRowPad = 128;
constexpr int kOutF = kLT_W * 3;                            // 96 (column, channel) positions per output row

struct Win { float w[kWin]; };

struct LossDims { int H, W, Hm, Wm; };                      // image and SSIM-map (valid convolution) sizes

// ---------------------------------------------------------------------------------------------------------------
// K1: moments -> SSIM value + partial derivatives at every window anchor
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ssim_stats_kernel(LossDims d, Win win, const float* __restrict__ pred,
                                                         const float* __restrict__ gt, float* __restrict__ map_a,
                                                         float* __restrict__ map_b, float* __restrict__ map_c,
                                                         float* __restrict__ partial_s) {
  __shared__ float sx[kPatchH][kRowPad], sy[kPatchH][kRowPad];
  __shared__ float hp[5][kPatchH][kOutF];
  __shared__ float red[4];
  const int ax0 = blockIdx.x * kLT_W, ay0 = blockIdx.y * kLT_H;       // first anchor of the tile
  // stage the patch: rows ay0 .. ay0+17, columns ax0 .. ax0+41 (clamped reads; out-of-image values are never used
  // by an anchor that is inside the map)
  for (int i = threadIdx.x; i < kPatchH * kRowF; i += 256) {
    const int r = i / kRowF, c = i - r * kRowF;
    const int y = min(ay0 + r, d.H - 1);
    const int xc = min(ax0 * 3 + c, d.W * 3 - 1);
    const size_t o = (size_t)y * d.W * 3 + xc;
    sx[r][c] = pred[o];
    sy[r][c] = gt[o];
  }
  __syncthreads();
  // horizontal pass: (row, column*3+channel) -> five filtered moments
  for (int i = threadIdx.x; i < kPatchH * kOutF; i += 256) {
    const int r = i / kOutF, c = i - r * kOutF;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const float x = sx[r][c + 3 * k], y = sy[r][c + 3 * k], w = win.w[k];
      m0 += w * x; m1 += w * y; m2 += w * (x * x); m3 += w * (y * y); m4 += w * (x * y);
    }
    hp[0][r][c] = m0; hp[1][r][c] = m1; hp[2][r][c] = m2; hp[3][r][c] = m3; hp[4][r][c] = m4;
  }
  __syncthreads();
  // vertical pass + SSIM algebra
  float s_sum = 0.f;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  for (int i = threadIdx.x; i < kLT_H * kOutF; i += 256) {
    const int r = i / kOutF, c = i - r * kOutF;
    const int ay = ay0 + r, ax = ax0 + c / 3;
    if (ay >= d.Hm || ax >= d.Wm) continue;
    float mx = 0.f, my = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const float w = win.w[k];
      mx += w * hp[0][r + k][c]; my += w * hp[1][r + k][c]; exx += w * hp[2][r + k][c];
      eyy += w * hp[3][r + k][c]; exy += w * hp[4][r + k][c];
    }
    const float sxx = exx - mx * mx, syy = eyy - my * my, sxy = exy - mx * my;
    const float A1 = 2.f * mx * my + C1, A2 = 2.f * sxy + C2;
    const float B1 = mx * mx + my * my + C1, B2 = sxx + syy + C2;
    const float iB1 = 1.f / B1, iB2 = 1.f / B2;
    const float s = A1 * A2 * iB1 * iB2;
    s_sum += s;
    // s as a function of (mu_x, E[xx], E[xy]) with the gt moments fixed:
    //   d s / d E[xx] = -s / B2,   d s / d E[xy] = 2 A1 / (B1 B2),
    //   d s / d mu_x  = 2 mu_y (A2 - A1) / (B1 B2) - 2 mu_x s / B1 + 2 mu_x s / B2
    const size_t o = ((size_t)ay * d.Wm + ax) * 3 + (c % 3);
    map_a[o] = 2.f * my * (A2 - A1) * iB1 * iB2 + 2.f * mx * s * (iB2 - iB1);
    map_b[o] = -s * iB2;
    map_c[o] = 2.f * A1 * iB1 * iB2;
  }
  // block sum of s (wave DPP reduce, then four partials through LDS): deterministic
  s_sum = wave_sum_uniform(s_sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_sum;
  __syncthreads();
  if (threadIdx.x == 0) partial_s[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// K2: transposed filter of the derivative maps -> d L / d pred (+ the L1 term), per-block sum of |gt - pred|
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_grad_kernel(LossDims d, Win win, const float* __restrict__ pred,
                                                        const float* __restrict__ gt, const float* __restrict__ map_a,
                                                        const float* __restrict__ map_b, const float* __restrict__ map_c,
                                                        float k_ssim /* -lambda / (3 Hm Wm) */,
                                                        float k_l1 /* (1 - lambda) / (3 H W) */,
                                                        float* __restrict__ v_pred, float* __restrict__ partial_l1) {
  __shared__ float sm[3][kPatchH][kRowPad];
  __shared__ float hp[3][kPatchH][kOutF];
  __shared__ float red[4];
  const int px0 = blockIdx.x * kLT_W, py0 = blockIdx.y * kLT_H;
  const bool with_ssim = k_ssim != 0.f;
  if (with_ssim) {
    // anchors (py0 - 10 .. py0 + 7) x (px0 - 10 .. px0 + 31); outside the map: zero
    for (int i = threadIdx.x; i < kPatchH * kRowF; i += 256) {
      const int r = i / kRowF, c = i - r * kRowF;
      const int ay = py0 - kHalo + r, ax = px0 - kHalo + c / 3;
      float a = 0.f, b = 0.f, cc = 0.f;
      if (ay >= 0 && ay < d.Hm && ax >= 0 && ax < d.Wm) {
        const size_t o = ((size_t)ay * d.Wm + ax) * 3 + (c % 3);
        a = map_a[o]; b = map_b[o]; cc = map_c[o];
      }
      sm[0][r][c] = a; sm[1][r][c] = b; sm[2][r][c] = cc;
    }
    __syncthreads();
    // out[t] = sum_d w[d] M[p - d] = sum_k w[k] patch[t + k] (symmetric window, patch index = anchor - (p0 - 10))
    for (int i = threadIdx.x; i < kPatchH * kOutF; i += 256) {
      const int r = i / kOutF, c = i - r * kOutF;
      float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        const float w = win.w[k];
        m0 += w * sm[0][r][c + 3 * k]; m1 += w * sm[1][r][c + 3 * k]; m2 += w * sm[2][r][c + 3 * k];
      }
      hp[0][r][c] = m0; hp[1][r][c] = m1; hp[2][r][c] = m2;
    }
    __syncthreads();
  }
  float l1 = 0.f;
  for (int i = threadIdx.x; i < kLT_H * kOutF; i += 256) {
    const int r = i / kOutF, c = i - r * kOutF;
    const int y = py0 + r, x3 = px0 * 3 + c;
    if (y >= d.H || x3 >= d.W * 3) continue;
    const size_t o = (size_t)y * d.W * 3 + x3;
    const float xp = pred[o], yg = gt[o];
    float g = 0.f;
    if (with_ssim) {
      float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        const float w = win.w[k];
        ga += w * hp[0][r + k][c]; gb += w * hp[1][r + k][c]; gc += w * hp[2][r + k][c];
      }
      g = k_ssim * (ga + 2.f * xp * gb + yg * gc);
    }
    const float df = xp - yg;
    l1 += fabsf(df);
    g += df > 0.f ? k_l1 : (df < 0.f ? -k_l1 : 0.f);                 // d |gt - pred| / d pred (0 at the kink, as torch)
    v_pred[o] = g;
  }
  l1 = wave_sum_uniform(l1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l1;
  __syncthreads();
  if (threadIdx.x == 0) partial_l1[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = loss, out[1] = mean |gt - pred|, out[2] = mean SSIM
__global__ __launch_bounds__(256) void loss_finalize_kernel(int n_s, const float* __restrict__ partial_s, int n_l1,
                                                            const float* __restrict__ partial_l1, double inv_map,
                                                            double inv_pix, float lambda, float* __restrict__ out) {
  __shared__ double red[2][4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n_s; i += 256) a += (double)partial_s[i];
  for (int i = threadIdx.x; i < n_l1; i += 256) b += (double)partial_l1[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ssim = n_s ? ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_map : 1.0;
    const double l1 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_pix;
    out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ssim));
    out[1] = (float)l1;
    out[2] = (float)ssim;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// multi-tensor Adam (torch.optim.Adam semantics: no weight decay, no amsgrad):
//   m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 8;
struct AdamArgs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  long long n[kAdamMaxTensors];
  unsigned block_end[kAdamMaxTensors];      // exclusive prefix of blocks per tensor
  float step_size[kAdamMaxTensors];         // lr / bias_correction1
  int count;
  float beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2;   // omb = 1 - beta, rounded from the DOUBLE difference like torch's scalars
};

constexpr int kAdamPerBlock = 256 * 4 * 4;  // 4 float4 per thread

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                         float eps, float isbc2, float step) {
  m = m + (g - m) * omb1;                   // torch: exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + omb2 * g * g;                // torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const float denom = sqrtf(v) * isbc2 + eps;
  p = p - step * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.block_end[t]) ++t;
  const unsigned b0 = t ? a.block_end[t - 1] : 0u;
  const long long n = a.n[t];
  float* __restrict__ P = a.p[t];
  const float* __restrict__ G = a.g[t];
  float* __restrict__ M = a.m[t];
  float* __restrict__ V = a.v[t];
  const float step = a.step_size[t];
  const long long base = (long long)(blockIdx.x - b0) * kAdamPerBlock;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
      float4 p = *reinterpret_cast<float4*>(P + i);
      const float4 g = *reinterpret_cast<const float4*>(G + i);
      float4 m = *reinterpret_cast<float4*>(M + i), v = *reinterpret_cast<float4*>(V + i);
      adam_one(p.x, g.x, m.x, v.x, a.omb1, a.beta2, a.omb2, a.eps, a.inv_sqrt_bc2, step);
      adam_one(p.y, g.y, m.y, v.y, a.omb1, a.beta2, a.omb2, a.eps, a.inv_sqrt_bc2, step);
      adam_one(p.z, g.z, m.z, v.z, a.omb1, a.beta2, a.omb2, a.eps, a.inv_sqrt_bc2, step);
      adam_one(p.w, g.w, m.w, v.w, a.omb1, a.beta2, a.omb2, a.eps, a.inv_sqrt_bc2, step);
      *reinterpret_cast<float4*>(P + i) = p;
      *reinterpret_cast<float4*>(M + i) = m;
      *reinterpret_cast<float4*>(V + i) = v;
    } else {
      for (long long j = i; j < n && j < i + 4; ++j) {
        float p = P[j], m = M[j], v = V[j];
        adam_one(p, G[j], m, v, a.omb1, a.beta2, a.omb2, a.eps, a.inv_sqrt_bc2, step);
        P[j] = p; M[j] = m; V[j] = v;
      }
    }
  }
}

}  // namespace gs

using namespace gs;

static Win make_window() {
  // pytorch_msssim / train_step.ssim: g[i] = exp(-(i - 5)^2 / (2 * 1.5^2)), normalised in float32
  Win w;
  float s = 0.f;
  for (int i = 0; i < kWin; ++i) { const float x = (float)i - 5.f; w.w[i] = expf(-(x * x) / (2.f * 1.5f * 1.5f)); s += w.w[i]; }
  for (int i = 0; i < kWin; ++i) w.w[i] /= s;
  return w;
}

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// workspace: three derivative maps [Hm, Wm, 3] + the per-block partial sums of both tile kernels
GS_EXPORT long long gs_image_loss_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const long long Hm = H >= kWin ? H - kHalo : 0, Wm = W >= kWin ? W - kHalo : 0;
  const long long maps = 3 * Hm * Wm * 3;
  const long long parts = (long long)cdiv(Wm > 0 ? Wm : 1, kLT_W) * cdiv(Hm > 0 ? Hm : 1, kLT_H) +
                          (long long)cdiv(W, kLT_W) * cdiv(H, kLT_H);
  return (maps + parts) * 4 + 256;
}

// Replaces splatfacto's loss + its autograd backward (nerfstudio 1.1.0 `get_loss_dict`: Ll1, simloss, ssim_lambda;
// reached from /root/reference/train.py:115-122).  pred / gt [H,W,3] fp32; v_pred [H,W,3] = d loss / d pred;
// loss_out[3] = {loss, mean |gt - pred|, mean SSIM} (device).  ssim_lambda == 0 skips the SSIM part (then any image
// size is allowed; with SSIM both sides must be >= 11).
GS_EXPORT int gs_image_loss_fwd_bwd(int H, int W, const float* pred, const float* gt, float ssim_lambda, float* v_pred,
                                    float* loss_out, void* workspace, long long workspace_bytes, void* stream) {
  if (H <= 0 || W <= 0 || !pred || !gt || !v_pred || !loss_out) return GS_ERR_INVALID;
  const bool with_ssim = ssim_lambda != 0.f;
  if (with_ssim && (H < kWin || W < kWin)) return GS_ERR_INVALID;
  if (workspace_bytes < gs_image_loss_workspace_bytes(H, W) || !workspace) return GS_ERR_WORKSPACE;
  LossDims d; d.H = H; d.W = W; d.Hm = with_ssim ? H - kHalo : 0; d.Wm = with_ssim ? W - kHalo : 0;
  const size_t map_n = (size_t)(H >= kWin ? H - kHalo : 0) * (size_t)(W >= kWin ? W - kHalo : 0) * 3;
  float* ws = reinterpret_cast<float*>(workspace);
  float* map_a = ws; float* map_b = ws + map_n; float* map_c = ws + 2 * map_n;
  float* part_s = ws + 3 * map_n;
  hipStream_t st = (hipStream_t)stream;
  const Win win = make_window();
  int n_s = 0;
  if (with_ssim) {
    dim3 g1(cdiv(d.Wm, kLT_W), cdiv(d.Hm, kLT_H));
    n_s = (int)(g1.x * g1.y);
    hipLaunchKernelGGL(ssim_stats_kernel, g1, dim3(256), 0, st, d, win, pred, gt, map_a, map_b, map_c, part_s);
  }
  float* part_l1 = part_s + (size_t)cdiv(d.Wm > 0 ? d.Wm : 1, kLT_W) * cdiv(d.Hm > 0 ? d.Hm : 1, kLT_H);
  dim3 g2(cdiv(W, kLT_W), cdiv(H, kLT_H));
  const double inv_pix = 1.0 / (3.0 * (double)H * (double)W);
  const double inv_map = with_ssim ? 1.0 / (3.0 * (double)d.Hm * (double)d.Wm) : 0.0;
  hipLaunchKernelGGL(loss_grad_kernel, g2, dim3(256), 0, st, d, win, pred, gt, map_a, map_b, map_c,
                     (float)(-(double)ssim_lambda * inv_map), (float)((1.0 - (double)ssim_lambda) * inv_pix), v_pred,
                     part_l1);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, n_s, part_s, (int)(g2.x * g2.y), part_l1, inv_map,
                     inv_pix, ssim_lambda, loss_out);
  return gs_launch_status();
}

// ONE Adam step over `count` (<= 8) tensors: params / grads / exp_avg / exp_avg_sq are host arrays of device
// pointers, numel / lr host arrays.  step = the 1-based step count AFTER this update (bias corrections
// 1 - beta^step, computed in double on the host like torch's Python scalars; betas / eps arrive as doubles for the
// same reason: torch rounds 1 - beta2 = 0.001 from the double difference, float(1) - float(0.999) is 4.7e-5 off).  Replaces torch.optim.Adam.step() of
// every parameter group (splatfacto's optimizers; eps 1e-15).
GS_EXPORT int gs_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const long long* numel, const float* lr, double beta1, double beta2,
                           double eps, int step, void* stream) {
  if (count <= 0 || count > kAdamMaxTensors || step <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr)
    return GS_ERR_INVALID;
  AdamArgs a;
  unsigned blocks = 0;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  int k = 0;
  for (int t = 0; t < count; ++t) {
    if (numel[t] <= 0) continue;
    if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]) return GS_ERR_INVALID;
    a.p[k] = params[t]; a.g[k] = grads[t]; a.m[k] = exp_avg[t]; a.v[k] = exp_avg_sq[t]; a.n[k] = numel[t];
    blocks += cdiv(numel[t], kAdamPerBlock);
    a.block_end[k] = blocks;
    a.step_size[k] = (float)((double)lr[t] / bc1);
    ++k;
  }
  if (k == 0) return GS_OK;
  a.count = k; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return gs_launch_status();
}
