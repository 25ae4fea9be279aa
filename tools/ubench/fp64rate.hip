// Micro-benchmark: issue rate and dependent latency of v_add_f64 / v_mul_f64 / v_fma_f64 on gfx950,
// with 1, 2, 4 and 8 resident waves per SIMD (blocks of one wavefront; 1024 blocks = one wave per SIMD).
// Results of round 1: profiles/r01/fp64_issue_rate.txt.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64rate fp64rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// KIND 0: x += c1   1: x *= c2   2: x = fma(x, c2, c1)   3: x += c1; x *= c2 (dependent pair)   4: x = (c1 - x) * c2
template <int KIND, int CH>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, double c1, double c2, int iters) {
  double x[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = out[threadIdx.x + 64 * i];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {             // CH independent chains, interleaved
        if (KIND == 0) x[i] = x[i] + c1;
        if (KIND == 1) x[i] = x[i] * c2;
        if (KIND == 2) x[i] = __builtin_fma(x[i], c2, c1);
        if (KIND == 3) { x[i] = x[i] + c1; x[i] = x[i] * c2; }
        if (KIND == 4) { x[i] = (c1 - x[i]) * c2; }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < CH; ++i) out[threadIdx.x + 64 * i] = x[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int CH>
void run(const char *name, int ops_per_elem, int blocks) {
  double *out;
  unsigned long long *cyc;
  CHECK(hipMalloc(&out, 64 * 64 * sizeof(double)));
  CHECK(hipMemset(out, 0, 64 * 64 * sizeof(double)));
  CHECK(hipMalloc(&cyc, 16));
  const int iters = 2000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  k<KIND, CH><<<blocks, 64>>>(out, cyc, 1.0e-9, 1.0000001, 10);
  CHECK(hipEventRecord(a));
  k<KIND, CH><<<blocks, 64>>>(out, cyc, 1.0e-9, 1.0000001, iters);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  unsigned long long h = 0;
  CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  const double n = (double)iters * 8 * CH * ops_per_elem;
  const double waves_per_simd = blocks / 1024.0;
  printf("%-28s chains=%2d blocks=%5d  %.3f ms  per-wave memtime/instr=%.3f  per-SIMD cycles/instr (2.39 GHz)=%.3f\n", name, CH,
         blocks, ms, h / n, ms * 1e-3 * 2.39e9 / (n * waves_per_simd));
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
}

int main() {
  for (int blocks : {1024, 2048, 4096, 8192}) {
    run<0, 8>("add", 1, blocks);
    run<1, 8>("mul", 1, blocks);
    run<2, 8>("fma", 1, blocks);
    run<3, 8>("add+mul (dep pair)", 2, blocks);
    run<4, 8>("sub,mul (dep pair)", 2, blocks);
    run<0, 1>("add dependent", 1, blocks);
    run<2, 1>("fma dependent", 1, blocks);
    run<2, 2>("fma 2 chains", 1, blocks);
    run<2, 4>("fma 4 chains", 1, blocks);
    run<3, 1>("add,mul dependent", 2, blocks);
    run<3, 4>("add,mul 4 chains", 2, blocks);
  }
  return 0;
}
