// Where do the 8 wavefronts of a 512-thread workgroup land?  Prints (cu, simd) per wave for a few blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = hw;
  // keep the block alive for a while so that all blocks are co-resident
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 2000000ull) { }
}
int main() {
  const int blocks = 256;
  unsigned *d, h[blocks * 8];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 140 * 1024, 0, d);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 140 * 1024, 0, d); e = hipDeviceSynchronize(); }
  printf("status %s\n", hipGetErrorString(e));
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 0;
  for (int b = 0; b < blocks; ++b) {
    int simd[8], cu[8];
    for (int w = 0; w < 8; ++w) { simd[w] = (h[b * 8 + w] >> 4) & 3; cu[w] = (h[b * 8 + w] >> 8) & 15; }
    bool paired = true;
    for (int w = 0; w < 4; ++w) paired &= simd[w] == simd[w + 4];
    int cnt[4] = {0, 0, 0, 0};
    for (int w = 0; w < 8; ++w) cnt[simd[w]]++;
    bool two_each = cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2;
    ok += paired && two_each;
    if (b < 6) { printf("block %d: simd", b); for (int w = 0; w < 8; ++w) printf(" %d", simd[w]); printf("  cu"); for (int w = 0; w < 8; ++w) printf(" %d", cu[w]); printf("  hw0=%08x\n", h[b * 8]); }
  }
  printf("blocks with wave w and w+4 on the same SIMD and two waves per SIMD: %d of %d\n", ok, blocks);
  return 0;
}
