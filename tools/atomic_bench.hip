// Micro-benchmark: fp32 atomic-add throughput on MI355X for the rasterize-backward pattern
// (each lane adds 9 consecutive floats of a random 48-byte record), agent vs workgroup scope,
// plus the plain-store rate of the same pattern for reference.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench && /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int SCOPE>
__global__ void k_atomic(float* buf, const unsigned* idx, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = buf + (size_t)idx[i] * 12;
#pragma unroll
  for (int c = 0; c < 9; ++c) __hip_atomic_fetch_add(p + c, 1.0f + c, __ATOMIC_RELAXED, SCOPE);
}
// SoA: component c of record r lives at buf[c*nrec + r]
__global__ void k_atomic_soa(float* buf, const unsigned* idx, size_t n, size_t nrec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = buf + idx[i];
#pragma unroll
  for (int c = 0; c < 9; ++c) __hip_atomic_fetch_add(p + (size_t)c * nrec, 1.0f + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one atomic per lane
__global__ void k_atomic_one(float* buf, const unsigned* idx, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  __hip_atomic_fetch_add(buf + (size_t)idx[i] * 12, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_store(float* buf, const unsigned* idx, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = buf + (size_t)idx[i] * 12;
#pragma unroll
  for (int c = 0; c < 9; ++c) p[c] = 1.0f + c;
}

int main() {
  const size_t nrec = 5u << 20;      // 5M records x 48 B = 240 MB (like P*N at 1M x 5)
  const size_t n = 64u << 20;        // 64M lanes -> 576M atomics
  float* buf; unsigned* idx;
  hipMalloc(&buf, nrec * 12 * sizeof(float));
  hipMalloc(&idx, n * sizeof(unsigned));
  std::vector<unsigned> h(n);
  unsigned s = 12345;
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: uniformly random records; mode 1: locally clustered (neighbouring lanes hit nearby records)
    for (size_t i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      h[i] = mode == 0 ? (s >> 8) % nrec : (unsigned)(((i / 64) * 37 + ((s >> 8) % 4096)) % nrec);
    }
    hipMemcpy(idx, h.data(), n * sizeof(unsigned), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int v = 0; v < 5; ++v) {
      hipMemset(buf, 0, nrec * 12 * sizeof(float));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        dim3 g((unsigned)((n + 255) / 256)), blk(256);
        if (v == 0) hipLaunchKernelGGL(k_atomic<__HIP_MEMORY_SCOPE_AGENT>, g, blk, 0, 0, buf, idx, n);
        else if (v == 1) hipLaunchKernelGGL(k_atomic<__HIP_MEMORY_SCOPE_WORKGROUP>, g, blk, 0, 0, buf, idx, n);
        else if (v == 2) hipLaunchKernelGGL(k_store, g, blk, 0, 0, buf, idx, n);
        else if (v == 3) hipLaunchKernelGGL(k_atomic_soa, g, blk, 0, 0, buf, idx, n, nrec);
        else hipLaunchKernelGGL(k_atomic_one, g, blk, 0, 0, buf, idx, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const char* nm[5] = {"atomic agent", "atomic workgroup", "plain store", "atomic SoA", "atomic 1/lane"};
      printf("%s %-17s: %8.3f ms  %7.2f G ops/s\n", mode == 0 ? "random   " : "clustered", nm[v], best,
             (double)n * (v == 4 ? 1 : 9) / best / 1e6);
    }
  }
  return 0;
}
