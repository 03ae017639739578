// Micro-probe (not part of the product): what the conv3x3 multiply loop CAN reach on this part.
//   mode 0: bare v_mfma_f32_32x32x16_f16 chains, 4 independent accumulators per wave                      -> sustained MFMA peak
//   mode 1: the multiply step of k_conv3x3w: per 4 MFMAs 4 ds_read_b128 fragments (1 KiB of LDS per MFMA)  -> LDS-fed ceiling
//   mode 2: mode 1 + one s_barrier per 16 MFMAs (the per-tap barrier)
//   mode 3: 2 A + 1 B fragment per 2 MFMAs... (64 px x 32 cout per wave: k_conv3x3r's shape)
// W waves per workgroup (one workgroup per CU): 8 = two multiplying waves per SIMD, 4 = one.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(1024, 1) void k_probe(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 32768; i += blockDim.x) ((float*)lds)[i] = (float)(i & 3) * 0.25f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  const char* base = lds + (wave & 7) * 16384 + lane * 16;
  h8 a0 = *(const h8*)base, a1 = *(const h8*)(base + 1024), b0 = *(const h8*)(base + 2048), b1 = *(const h8*)(base + 3072);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      if (MODE >= 1) {
        const int o = ((it * 4 + kk) & 3) * 4096;
        a0 = *(const h8*)(base + o);
        a1 = *(const h8*)(base + o + 1024);
        b0 = *(const h8*)(base + o + 2048);
        if (MODE != 3) b1 = *(const h8*)(base + o + 3072);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc[1], 0, 0, 0);
      if (MODE != 3) {
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[3], 0, 0, 0);
      }
    }
    if (MODE == 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  if (s == 12345.678f) *sink = s;
}

// The conv3x3w step with its loaders: 8 multiplying waves (mode-2 loop: 16 ds_read_b128 + 16 MFMAs + one barrier per step) and 8
// loader waves that move PIECES one-KiB pieces per step into a ring of LDS slots and pass the same barrier behind a counted wait.
//   HOW 0: loaders only wait at the barrier (no traffic)          HOW 1: LDS-DMA (global_load_lds_dwordx4), as the kernel
//   HOW 2: register staging (global_load_dwordx4 + ds_write_b128)  HOW 3: LDS-DMA issued but never waited for until the end
template <int HOW, int PIECES>
__global__ __launch_bounds__(1024, 1) void k_step(const char* __restrict__ src, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [0, 64K): fragments read by the multipliers; [64K, 160K): DMA ring
  const int tid = threadIdx.x & 511, wave = tid >> 6, lane = tid & 63;
  const bool loader = threadIdx.x >= 512;
  for (int i = threadIdx.x; i < 16384; i += 1024) ((float*)lds)[i] = (float)(i & 3) * 0.25f;
  __syncthreads();
  if (loader) {
    // every CU reads the same 64 KB (the W tiles of one cout block) from L2, as the kernel's weight stream does
    const char* g = src + lane * 16 + wave * 1024;
    char* dst = lds + 65536 + wave * 1024;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int it = 0; it < iters; it++) {
      if (HOW == 1 || HOW == 3) {
#pragma unroll
        for (int p = 0; p < PIECES; p++)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * PIECES + p) & 7) * 8192),
                                           (__attribute__((address_space(3))) void*)(dst + ((it * PIECES + p) % 11) * 8192), 16, 0, 0);
        if (HOW == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");  // pieces of the two youngest steps stay in flight
      } else if (HOW == 2) {
        u4 v[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; p++) v[p] = *(const u4*)(g + ((it * PIECES + p) & 7) * 8192);
#pragma unroll
        for (int p = 0; p < PIECES; p++) *(u4*)(dst + ((it * PIECES + p) % 11) * 8192 + lane * 16) = v[p];
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  f32x16 acc[4];
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  const char* base = lds + wave * 8192 + lane * 16;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int o = ((it + kk) & 1) * 4096;
      const h8 a0 = *(const h8*)(base + o), a1 = *(const h8*)(base + o + 1024), b0 = *(const h8*)(base + o + 2048),
               b1 = *(const h8*)(base + o + 3072);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[3], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  if (s == 12345.678f) *sink = s;
}

template <int HOW, int PIECES>
static void run_step(const char* src, int iters, float* sink) {
  hipFuncSetAttribute((const void*)k_step<HOW, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_step<HOW, PIECES>), dim3(256), dim3(1024), 163840, 0, src, iters, sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_step<HOW, PIECES>), dim3(256), dim3(1024), 163840, 0, src, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double tf = 16.0 * iters * 8 * 256.0 * 32768.0 / (best * 1e-3) / 1e12;
  printf("step probe how %d, %d KiB staged per step and CU: %8.3f ms  %7.1f TFLOP/s  %.1f ns per step  (%.1f GB/s per CU)\n", HOW, PIECES * 8, best, tf,
         best * 1e6 / iters, PIECES * 8192.0 / (best * 1e6 / iters));
}

template <int MODE>
static void run(int waves, int iters, float* sink) {
  hipFuncSetAttribute((const void*)k_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(256), dim3(64 * waves), 131072, 0, iters, sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(256), dim3(64 * waves), 131072, 0, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mf = (MODE == 3 ? 8.0 : 16.0) * iters * waves * 256.0;  // MFMAs
  const double tf = mf * 32768.0 / (best * 1e-3) / 1e12;
  printf("mode %d waves/CU %2d iters %6d: %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)  %.1f ns per 16-MFMA step per wave\n", MODE, waves, iters, best, tf,
         tf / 2500.0, best * 1e6 / iters);
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  for (int iters : {20000}) {
    for (int waves : {8, 16}) {
      run<0>(waves, iters, sink);
      run<1>(waves, iters, sink);
      run<2>(waves, iters, sink);
      run<3>(waves, iters, sink);
    }
  }
  char* src;
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  const int it = 20000;
  run_step<0, 1>(src, it, sink);
  run_step<1, 1>(src, it, sink);
  run_step<1, 2>(src, it, sink);
  run_step<1, 3>(src, it, sink);
  run_step<1, 4>(src, it, sink);
  run_step<3, 3>(src, it, sink);
  run_step<2, 1>(src, it, sink);
  run_step<2, 2>(src, it, sink);
  run_step<2, 3>(src, it, sink);
  return 0;
}
