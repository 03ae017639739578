// Diagnostic (round 5): a kernel that merely OCCUPIES part of every CU for a while - `lds_bytes` of LDS per workgroup, 64 threads, a
// handful of registers - so that the workgroups of a kernel launched on another stream are placed beside it.  mode 0: sleeps;
// mode 1: hammers its LDS; mode 2: streams global memory (buf, n floats).  Used by tools/conv_corun.py and tests/test_gpu_corun.py to find out what a
// co-resident foreign workgroup does to the LDS-DMA convolution kernels (DESIGN_LOG.md, round 5).
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
__global__ __launch_bounds__(64) void k_squat(int mode, long long ticks, float* buf, long long n) {
  extern __shared__ float sm[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  long long i = threadIdx.x + (long long)blockIdx.x * 64;
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(32);
    } else if (mode == 1) {
      for (int k = 0; k < 64; k++) {
        sm[(threadIdx.x + 64 * k) & 255] = acc;
        acc += sm[(threadIdx.x * 7 + k) & 255];
      }
    } else {
      for (int k = 0; k < 16; k++) {
        acc += buf[i % n];
        i += 64 * 4096 + 17;
      }
    }
  }
  if (acc == 12345.678f) buf[0] = acc;
}
int squat(int grid, int lds_bytes, int mode, long long ticks, float* buf, long long n, hipStream_t s) {
  if (lds_bytes > 65536) (void)hipFuncSetAttribute((const void*)k_squat, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(k_squat, dim3(grid), dim3(64), lds_bytes < 1024 && mode == 1 ? 1024 : lds_bytes, s, mode, ticks, buf, n);
  return (int)hipGetLastError();
}
}
