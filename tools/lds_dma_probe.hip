// Micro-probe (not part of the product): does LDS-DMA traffic (global_load_lds_dwordx4) slow the ds_read_b128 / MFMA stream of
// OTHER waves on the same CU?  One 8-wave workgroup per CU: waves 0..3 run `iters` rounds of 16 ds_read_b128 (+ 16 MFMAs when
// mode & 2), waves 4..7 stream `pieces` 1-KiB DMA pieces per round from a 2 MB L2-resident buffer when mode & 1.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void k_probe(const char* __restrict__ src, int iters, int pieces, int mode, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [0, 64K): read area, [64K, 128K): DMA landing area
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) ((float*)lds)[i] = (float)(i & 7);
  __syncthreads();
  if (wave >= 4) {
    if (!(mode & 1)) return;
    const char* g = src + ((size_t)(blockIdx.x & 31) * 65536) + lane * 16;
    for (int it = 0; it < iters; it++) {
      for (int p = 0; p < pieces; p++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * pieces + p) & 63) * 1024),
                                         (__attribute__((address_space(3))) void*)(lds + 65536 + (wave - 4) * 16384 + (p & 15) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  f32x16 acc[4];
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  const char* base = lds + wave * 16384 + lane * 16;
  for (int it = 0; it < iters; it++) {
    bf16x8 f[16];
#pragma unroll
    for (int k = 0; k < 16; k++) f[k] = *(const bf16x8*)(base + ((k * 1024 + it * 16) & 16383 & ~15));
    if (mode & 2) {
#pragma unroll
      for (int k = 0; k < 16; k++) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[k], f[(k + 1) & 15], acc[k & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) acc[k & 3][0] += (float)f[k][0];
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  if (s == 12345.678f) *sink = s;
}

int main() {
  char* src;
  float* sink;
  hipMalloc(&src, 4 << 20);
  hipMemset(src, 0, 4 << 20);
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  printf("per round: 4 waves x 16 ds_read_b128 (64 KB per CU) [+ 4 x 16 MFMA 32x32x16]; loaders: 4 waves x `pieces` KB per round\n");
  for (int mode : {0, 2}) {
    for (int pieces : {0, 2, 4, 8}) {
      const int m = mode | (pieces ? 1 : 0);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 131072, 0, src, iters, pieces ? pieces : 1, m, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s, DMA %2d KB/round/CU: %7.3f us per round  (reads %6.1f GB/s/CU, DMA %6.1f GB/s/CU)\n", mode & 2 ? "reads+MFMA" : "reads only",
             pieces * 4, ms * 1e3 / iters, 65536.0 / (ms * 1e-3 / iters) / 1e9, pieces * 4096.0 / (ms * 1e-3 / iters) / 1e9);
    }
  }
  return 0;
}
