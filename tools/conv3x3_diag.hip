// Diagnostic harness (not part of the product): what one (segment, tap) step of k_conv3x3w is made of.
// Includes csrc/conv2d.hip with (C3P::diag switches parts of the kernel off) and times the kernel on a balanced
// workload (every workgroup gets the same number of items), per configuration.
//   hipcc --offload-arch=gfx950 -O3 -I mm2d3d_amd/csrc tools/conv3x3_diag.hip -o /tmp/conv3x3_diag && /tmp/conv3x3_diag
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "conv2d.hip"

static char g_err[512];
void mm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

template <int BN, int TW, int DIAG>
static float run(int B, int H, int W, int C, const u16* A, u16* O, const u16* Wp, int* steps_per_wg) {
  C3P p = {};
  p.A = A, p.B = B, p.H = H, p.W = W, p.Ca = C, p.lda = C, p.O = O, p.Cn = C, p.ldo = C, p.Wp = Wp, p.bias = nullptr, p.flip = 0;
  p.tiles_y = (H + 256 / TW - 1) / (256 / TW), p.tiles_x = (W + TW - 1) / TW;
  const int nitems = B * p.tiles_y * p.tiles_x * (C / BN);
  *steps_per_wg = (nitems + 255) / 256 * (C / 64) * 9;
  const size_t lds = (size_t)(2 * 344 * 64 + (BN == 128 ? 4 : 6) * BN * 64) * 2;
  hipFuncSetAttribute((const void*)k_conv3x3w<BN, TW, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_conv3x3w<BN, TW, DIAG>), dim3(256), dim3(1024), lds, 0, p);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_conv3x3w<BN, TW, DIAG>), dim3(256), dim3(1024), lds, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  struct Shape { int B, H, W, C; } shapes[] = {{128, 38, 60, 256}, {64, 76, 120, 128}, {128, 19, 30, 512}, {16, 38, 60, 256}, {16, 76, 120, 128}, {16, 19, 30, 512}};
  const char* names[256] = {};
  names[0] = "full kernel", names[1] = "no MFMA (fragment reads kept)", names[2] = "no fragment reads, no MFMA";
  names[4] = "W DMA from the zero line", names[8] = "halo DMA from the zero line", names[12] = "W + halo DMA from the zero line";
  names[16] = "no output stores", names[32] = "no epilogue", names[44] = "no epilogue, DMA from the zero line";
  names[33] = "no MFMA, no epilogue", names[34] = "no reads / MFMA, no epilogue", names[46] = "barriers + zero-line DMA only";
  names[45] = "reads only + zero-line DMA, no epilogue";
  names[128] = "no epilogue, MFMAs kept (asm use)", names[140] = "no epilogue, MFMAs kept, zero-line DMA";
  names[64] = "epilogue without its stores", names[76] = "no stores, DMA from the zero line";
  for (auto sh : shapes) {
    const size_t na = (size_t)sh.B * sh.H * sh.W * sh.C, nw = (size_t)sh.C * 9 * sh.C;
    u16 *A, *O, *Wp;
    hipMalloc(&A, na * 2);
    hipMalloc(&O, na * 2);
    hipMalloc(&Wp, nw * 2);
    hipMemset(A, 0, na * 2);
    hipMemset(Wp, 0, nw * 2);
    printf("%d -> %d @ %dx%d, B = %d\n", sh.C, sh.C, sh.H, sh.W, sh.B);
    const bool big = sh.H * sh.W >= 2000;
#define ONE(D)                                                                                                                       \
  {                                                                                                                                  \
    int steps = 0;                                                                                                                   \
    const float ms = big ? run<128, 16, D>(sh.B, sh.H, sh.W, sh.C, A, O, Wp, &steps) : run<128, 32, D>(sh.B, sh.H, sh.W, sh.C, A, O, Wp, &steps); \
    const double tf = 2.0 * sh.B * sh.H * sh.W * (double)sh.C * sh.C * 9 / (ms * 1e-3) / 1e12;                                       \
    printf("  diag %2d %-42s %8.1f us  %6.3f us per step (%d steps per workgroup)  %6.0f TFLOP/s nominal\n", D, names[D], ms * 1e3,  \
           ms * 1e3 / steps, steps, tf);                                                                                             \
  }
    ONE(0) ONE(2) ONE(12) ONE(128) ONE(140) ONE(46)
#undef ONE
    hipFree(A);
    hipFree(O);
    hipFree(Wp);
  }
  return 0;
}
