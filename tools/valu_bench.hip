// valu_bench.hip — issue-rate micro-benchmark of the VALU / LDS instructions the compositor kernels are made of
// (gfx950, wave64).  For every instruction kind: a loop of 64 independent instructions (8 accumulators) per
// iteration, timed per wave with s_memtime, at 1 / 2 / 4 / 8 waves per SIMD on every CU (one 64*k-thread block per
// CU, pinned by a 100 KB LDS request).  Prints shader cycles per wave-instruction per SIMD — the number the
// "VALU issue limit" arguments in DESIGN.md have to be priced against (MI355X_MICROARCH.md quotes 2 cycles for a
// plain wave64 VALU op; packed f32 and transcendental ops are not in its table).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA, FMA_S, PKFMA, PKMUL, PKADD, EXP, RCP, CNDVCC, CNDSGPR, CMPS, RDLANE, MINF, MULF, MIXFWD, DSW32, DSR128, NOPS };
static const char* kNames[] = {"v_fma_f32", "v_fma_f32 (sgpr src)", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                               "v_exp_f32", "v_rcp_f32", "v_cndmask (vcc)", "v_cndmask_e64 (sgpr mask)",
                               "v_cmp_lt_f32_e64 -> sgpr", "v_readlane_b32", "v_min_f32", "v_mul_f32",
                               "mix: 2 pk_fma + exp + mul + min + 2 cmp + cnd", "ds_write_b32", "ds_read_b128"};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ void bench(int iters, unsigned long long* cycles, float* sink, float sval) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i); p[i] = f2{a[i], a[i] + 1.f}; }
  const float b = 0.999f, c = 1e-4f;
  const f2 b2 = {b, b}, c2 = {c, c};
  float* myl = lds + threadIdx.x * 4;
  unsigned long long smask = 0x5555555555555555ull;
  int sacc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == FMA_S) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sval), "v"(c));
        REP8(X)
#undef X
      } else if (OP == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
        REP8(X)
#undef X
      } else if (OP == PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
        REP8(X)
#undef X
      } else if (OP == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        REP8(X)
#undef X
      } else if (OP == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == CNDVCC) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CNDSGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(smask));
        REP8(X)
#undef X
      } else if (OP == CMPS) {
#define X(i) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b)); sacc += (int)m; }
        REP8(X)
#undef X
      } else if (OP == RDLANE) {
#define X(i) { int s; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(a[i]), "s"(it & 63)); sacc += s; }
        REP8(X)
#undef X
      } else if (OP == MINF) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MULF) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MIXFWD) {
        // the forward compositor's per-pixel-pair mix: 8 "instructions" = 2 pk_fma + exp + mul + min + 2 cmp + cndmask
#define X(i) { unsigned long long m0, m1; \
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %0, %0, %4, %5\n\tv_exp_f32 %1, %1\n\tv_mul_f32 %1, %1, %6\n\t" \
                     "v_min_f32 %1, %1, %6\n\tv_cmp_lt_f32_e64 %2, %1, %6\n\tv_cmp_gt_f32_e64 %3, %1, %7\n\t" \
                     "v_cndmask_b32_e64 %1, %1, %6, %2" \
                     : "+v"(p[i]), "+v"(a[i]), "=&s"(m0), "=&s"(m1) : "v"(b2), "v"(c2), "v"(b), "v"(c)); sacc += (int)m1; }
        REP8(X)
#undef X
      } else if (OP == DSW32) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"((unsigned)(threadIdx.x * 4)), "v"(a[i]), "i"(i * 4096) : "memory");
        REP8(X)
#undef X
      } else if (OP == DSR128) {
#define X(i) { f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x * 16)), "i"(i * 4096) : "memory"); \
               asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); a[i] += v.x; }
        REP8(X)
#undef X
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (threadIdx.x == 0x7fffffff) sink[0] = s + (float)sacc + myl[0];
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run_one() {
  const int iters = 2000;
  const int per_iter = (OP == MIXFWD) ? 64 * 8 : 64;
  unsigned long long* d_cycles;
  float* d_sink;
  hipMalloc(&d_cycles, sizeof(unsigned long long) * 256 * 32);
  hipMalloc(&d_sink, 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  printf("%-50s", kNames[OP]);
  for (int wps : {1, 2, 4, 8}) {
    const int waves = 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(256), dim3(64 * waves), 100 * 1024, 0, 10, d_cycles, d_sink, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, dim3(256), dim3(64 * waves), 100 * 1024, 0, iters, d_cycles, d_sink, 0.999f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * waves);
    hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // cycles per wave-instruction per SIMD = wave cycles / (instructions per wave * waves per SIMD)
    const double cpi = med / ((double)iters * per_iter * wps);
    // wall-clock view: ms * 2.4e6 cycles / (instr per SIMD)
    const double cpi_wall = (double)ms * 2.4e6 / ((double)iters * per_iter * wps);
    printf("  w/SIMD=%d: %5.2f cyc (wall@2.4GHz %5.2f)", wps, cpi, cpi_wall);
  }
  printf("\n");
  hipFree(d_cycles); hipFree(d_sink);
}

int main() {
  run_one<FMA>(); run_one<FMA_S>(); run_one<PKFMA>(); run_one<PKMUL>(); run_one<PKADD>(); run_one<EXP>();
  run_one<RCP>(); run_one<CNDVCC>(); run_one<CNDSGPR>(); run_one<CMPS>(); run_one<RDLANE>(); run_one<MINF>();
  run_one<MULF>(); run_one<MIXFWD>(); run_one<DSW32>(); run_one<DSR128>();
  return 0;
}
