// Debug aid: occupies `blocks` CUs with workgroups that hold `lds` bytes of LDS and spin for `cycles` clock ticks - a stand-in
// for a concurrently running collective kernel (RCCL) when studying how the persistent convolution kernels behave when some
// CUs cannot host their workgroups.  Built on the GPU box by tools/hog_test.py; not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void k_hog(long long cycles, int* sink) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = wall_clock64();
  int acc = 0;
  while (wall_clock64() - t0 < cycles) acc += sm[(threadIdx.x + acc) & 255];
  if (acc == 0x7fffffff) *sink = acc;
}
extern "C" int hog_launch(int blocks, int lds, long long cycles, int* sink, hipStream_t s) {
  hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(256), lds, s, cycles, sink);
  return (int)hipGetLastError();
}
