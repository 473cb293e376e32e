// gs_common.h — shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gs_math.h"

#define GS_OK 0
#define GS_ERR_INVALID 1
#define GS_ERR_WORKSPACE 3
// launch errors are returned as 1000 + hipError_t

#define GS_EXPORT extern "C" __attribute__((visibility("default")))

static inline int gs_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GS_OK : 1000 + (int)e;
}

namespace gs {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- wave64 DPP reduction: total lands in lane 63 ---------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}

// sum over the 64 lanes; result valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row = row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = wave total
  return v;
}

__device__ __forceinline__ float wave_sum_uniform(float v) {
  v = wave_sum_to_lane63(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o);
    v = v > t ? v : t;
  }
  return v;
}

// hardware fp32 atomic add, agent scope, no return
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XCD-aware remap: hardware places workgroup b on XCD b % 8 (speed only, never
// correctness).  Give each XCD a contiguous chunk of the logical index space so
// neighbouring tiles (which share Gaussians) hit the same L2.  Bijective for any n.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned X = 8;
  unsigned q = n / X, r = n % X;
  unsigned xcd = b % X, i = b / X;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

// ---- depth slices: which depth ranks of which sub-pose a slice holds --------------------------------------------
// passed BY VALUE in the kernel arguments (filled from host arrays): a slice is described on the host right
// after the plan read-back, and an upload would put a host->device copy on the critical path of every slice
constexpr int kMaxSubposes = 256;
struct SliceDesc {
  int begin[kMaxSubposes];        // first depth rank (absolute index into sorted_gi) of the slice in sub-pose p
  int prefix[kMaxSubposes + 1];   // prefix sums of the per-sub-pose slice lengths
  int P;
};

static inline bool make_slice_desc(int P, const int* begin, const int* prefix, SliceDesc& sd) {
  if (P <= 0 || P > kMaxSubposes || !begin || !prefix) return false;
  for (int p = 0; p < P; ++p) { sd.begin[p] = begin[p]; sd.prefix[p] = prefix[p]; }
  sd.prefix[P] = prefix[P];
  sd.P = P;
  return true;
}

__device__ __forceinline__ int slice_rank(const SliceDesc& sd, int j) {
  int p = 0;
  while (p + 1 < sd.P && j >= sd.prefix[p + 1]) ++p;
  return sd.begin[p] + (j - sd.prefix[p]);
}

}  // namespace gs
