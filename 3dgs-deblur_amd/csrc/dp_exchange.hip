// dp_exchange.hip — device side of the row-sparse data-parallel gradient exchange (dp.py).
//
// Early termination leaves ~1 % of the Gaussians with a gradient for a given view, so ranks exchange only
// their non-zero gradient ROWS.  The six Gaussian gradient tensors (means[N,3] scales[N,3] quats[N,4]
// opacities[N,1] features_dc[N,3] features_rest[N,15,3]: 59 floats per row at SH degree 3) are addressed
// through a small by-value table, so each step is ONE launch instead of one torch kernel per tensor:
//   row_mask     which rows hold any non-zero gradient                     (reads 236 B / row)
//   pack_rows    [M, wtot+1] payload: the rows' floats + the row index as the bit pattern of a last column
//   scatter_add  adds a payload into the tensors; a payload's row indices are unique, so there are no
//                colliding adds and no atomics — the caller applies the ranks' payloads in rank order, which
//                makes the sum bit-identical on every rank (replicas must not drift).
#include "gs_common.h"

namespace gs {

constexpr int kMaxDpTensors = 16;

struct DpTable {
  float* ptr[kMaxDpTensors];
  int width[kMaxDpTensors];     // floats per row
  int start[kMaxDpTensors];     // first payload column
  int n, wtot;
};

// Grid-stride over every tensor's flat element space: adjacent lanes read adjacent floats (a thread-per-row
// walk would touch 64 cache lines per load), a non-zero element stores 1 into its row's byte (benign race: every
// writer stores the same value).  mask must be zeroed by the caller (gs_dp_row_mask does it).
__global__ __launch_bounds__(256) void dp_row_mask_kernel(int N, DpTable tb, unsigned char* __restrict__ mask) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int t = 0; t < tb.n; ++t) {
    const float* __restrict__ p = tb.ptr[t];
    const unsigned w = (unsigned)tb.width[t];
    const size_t total = (size_t)N * w;
    for (size_t i = gid; i < total; i += stride)
      if (p[i] != 0.f) mask[i / w] = 1;
  }
}

// one thread per payload element; column wtot carries the row index
__global__ __launch_bounds__(256) void dp_pack_rows_kernel(long long M, const long long* __restrict__ idx, DpTable tb,
                                                           float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = tb.wtot + 1;
  if (e >= M * stride) return;
  const long long r = e / stride;
  const int col = (int)(e - r * stride);
  const long long row = idx[r];
  if (col == tb.wtot) { out[e] = __int_as_float((int)row); return; }
  int t = 0;
  while (t + 1 < tb.n && col >= tb.start[t + 1]) ++t;
  out[e] = tb.ptr[t][(size_t)row * tb.width[t] + (col - tb.start[t])];
}

__global__ __launch_bounds__(256) void dp_scatter_add_kernel(long long M, const float* __restrict__ payload,
                                                             DpTable tb, float scale) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = tb.wtot + 1;
  if (e >= M * stride) return;
  const long long r = e / stride;
  const int col = (int)(e - r * stride);
  if (col == tb.wtot) return;
  const int row = __float_as_int(payload[r * stride + tb.wtot]);
  int t = 0;
  while (t + 1 < tb.n && col >= tb.start[t + 1]) ++t;
  float* dst = tb.ptr[t] + (size_t)row * tb.width[t] + (col - tb.start[t]);
  *dst += payload[e] * scale;
}

// ---- fixed-capacity, sync-free form (round 2): nothing about the number of rows ever reaches the host ----------
// payload = [cap + 1][wtot + 1] floats; row 0 is a header (int bit patterns): [0] = rows with a gradient on the
// sending rank (may exceed cap: the receiver can tell the payload was truncated), [1] = rows actually packed.
// idx[pos[r]] = r for the masked rows whose rank pos[r] (exclusive scan of the mask) fits the capacity
__global__ __launch_bounds__(256) void dp_compact_rows_kernel(int N, const unsigned char* __restrict__ mask,
                                                              const int* __restrict__ pos, int cap,
                                                              int* __restrict__ idx, float* __restrict__ header) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  if (mask[r] && pos[r] < cap) idx[pos[r]] = r;
  if (r == N - 1) {
    const int total = pos[r] + (mask[r] ? 1 : 0);
    header[0] = __int_as_float(total);
    header[1] = __int_as_float(total < cap ? total : cap);
  }
}

// one thread per payload element of the cap data rows; the packed-row count is read from the header
__global__ __launch_bounds__(256) void dp_pack_rows_dev_kernel(int cap, const int* __restrict__ idx, DpTable tb,
                                                               float* __restrict__ payload) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = tb.wtot + 1;
  const int count = __float_as_int(payload[1]);
  if (e >= (long long)count * stride) return;
  const long long r = e / stride;
  const int col = (int)(e - r * stride);
  const int row = idx[r];
  float* out = payload + stride;                       // data rows start after the header row
  if (col == tb.wtot) { out[e] = __int_as_float(row); return; }
  int t = 0;
  while (t + 1 < tb.n && col >= tb.start[t + 1]) ++t;
  out[e] = tb.ptr[t][(size_t)row * tb.width[t] + (col - tb.start[t])];
}

__global__ __launch_bounds__(256) void dp_scatter_add_dev_kernel(int cap, const float* __restrict__ payload,
                                                                 DpTable tb, float scale) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = tb.wtot + 1;
  const int count = __float_as_int(payload[1]);
  if (e >= (long long)count * stride) return;
  const long long r = e / stride;
  const int col = (int)(e - r * stride);
  if (col == tb.wtot) return;
  const float* rows = payload + stride;
  const int row = __float_as_int(rows[r * stride + tb.wtot]);
  int t = 0;
  while (t + 1 < tb.n && col >= tb.start[t + 1]) ++t;
  float* dst = tb.ptr[t] + (size_t)row * tb.width[t] + (col - tb.start[t]);
  *dst += rows[e] * scale;
}

static bool make_table(int n, float* const* ptrs, const int* widths, DpTable& tb) {
  if (n <= 0 || n > kMaxDpTensors) return false;
  tb.n = n;
  int s = 0;
  for (int t = 0; t < n; ++t) {
    if (!ptrs[t] || widths[t] <= 0) return false;
    tb.ptr[t] = ptrs[t]; tb.width[t] = widths[t]; tb.start[t] = s;
    s += widths[t];
  }
  for (int t = n; t < kMaxDpTensors; ++t) { tb.ptr[t] = nullptr; tb.width[t] = 0; tb.start[t] = s; }
  tb.wtot = s;
  return true;
}

}  // namespace gs

using namespace gs;

// New design (the reference is single-GPU): row-sparse exchange of the Gaussian gradients.
GS_EXPORT int gs_dp_row_mask(int N, int n_tensors, float* const* grads /*host array of device pointers*/,
                             const int* widths /*host array: floats per row*/, unsigned char* mask /*N*/,
                             void* stream) {
  DpTable tb;
  if (N <= 0 || !make_table(n_tensors, grads, widths, tb)) return GS_ERR_INVALID;
  hipError_t me = hipMemsetAsync(mask, 0, (size_t)N, (hipStream_t)stream);
  if (me != hipSuccess) return 1000 + (int)me;
  hipLaunchKernelGGL(dp_row_mask_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, N, tb, mask);
  return gs_launch_status();
}

GS_EXPORT int gs_dp_pack_rows(long long M, const long long* row_idx /*M, device*/, int n_tensors,
                              float* const* grads, const int* widths, float* payload /*M*(wtot+1)*/,
                              void* stream) {
  DpTable tb;
  if (M <= 0 || !make_table(n_tensors, grads, widths, tb)) return GS_ERR_INVALID;
  const long long total = M * (tb.wtot + 1);
  hipLaunchKernelGGL(dp_pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M,
                     row_idx, tb, payload);
  return gs_launch_status();
}

GS_EXPORT int gs_dp_scatter_add_rows(long long M, const float* payload /*M*(wtot+1), unique row indices*/,
                                     int n_tensors, float* const* grads, const int* widths, float scale,
                                     void* stream) {
  DpTable tb;
  if (M <= 0 || !make_table(n_tensors, grads, widths, tb)) return GS_ERR_INVALID;
  const long long total = M * (tb.wtot + 1);
  hipLaunchKernelGGL(dp_scatter_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     M, payload, tb, scale);
  return gs_launch_status();
}

// ---- fixed-capacity, sync-free form: the row count stays on the device ------------------------------------------
// payload [(cap+1)*(wtot+1)]: row 0 = header {rows with a gradient, rows packed = min(that, cap)} as int bit patterns,
// rows 1.. = the packed rows.  pos = exclusive prefix sum of mask (int32 [N]); idx_ws is scratch for cap ints.
GS_EXPORT int gs_dp_pack_masked_rows(int N, const unsigned char* mask, const int* pos, int cap, int* idx_ws,
                                     int n_tensors, float* const* grads, const int* widths, float* payload,
                                     void* stream) {
  DpTable tb;
  if (N <= 0 || cap <= 0 || !make_table(n_tensors, grads, widths, tb)) return GS_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dp_compact_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, mask, pos, cap, idx_ws, payload);
  const long long total = (long long)cap * (tb.wtot + 1);
  hipLaunchKernelGGL(dp_pack_rows_dev_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cap, idx_ws, tb,
                     payload);
  return gs_launch_status();
}

// grads[t][row] += scale * row for the header[1] rows of a payload built by gs_dp_pack_masked_rows
GS_EXPORT int gs_dp_scatter_add_payload(int cap, const float* payload, int n_tensors, float* const* grads,
                                        const int* widths, float scale, void* stream) {
  DpTable tb;
  if (cap <= 0 || !make_table(n_tensors, grads, widths, tb)) return GS_ERR_INVALID;
  const long long total = (long long)cap * (tb.wtot + 1);
  hipLaunchKernelGGL(dp_scatter_add_dev_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cap, payload, tb, scale);
  return gs_launch_status();
}
