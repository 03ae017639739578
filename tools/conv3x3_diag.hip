// Diagnostic harness (not part of the product): what one (segment, tap) step of the 3x3 kernels is made of.
// Includes csrc/conv2d.hip (the DIAG template parameter switches parts of a kernel off at compile time) and times
// k_conv3x3w (round 2) and k_conv3x3v (round 6) per configuration, on balanced large-batch shapes and at the bench's item counts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mm2d3d_amd/csrc tools/conv3x3_diag.hip -o tools/_bin/conv3x3_diag && tools/_bin/conv3x3_diag
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

template <typename K>
static float time_kernel(K kern, int threads, C3P p) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 163840, 0, p);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 7; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 163840, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best;
}

#ifdef MM_DIAG_CLOCK
#include <algorithm>
#include <vector>
// in-kernel clock of the last launch (GHz): median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz
static double last_clock() {
  static unsigned long long h[3][1024][2];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
  std::vector<double> v;
  for (int b = 0; b < 256; b++)
    if (h[0][b][1] > 0) v.push_back((double)h[0][b][0] / (double)h[0][b][1] * 0.1);
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}
#else
static double last_clock() { return 0.0; }
#endif

static C3P make(int B, int H, int W, int C, int tw, const u16* A, u16* O, const u16* Wp) {
  C3P p = {};
  p.A = A, p.B = B, p.H = H, p.W = W, p.Ca = C, p.lda = C, p.O = O, p.Cn = C, p.ldo = C, p.Wp = Wp, p.bias = nullptr, p.flip = 0;
  p.B1 = B;
  p.tiles_y = (H + 256 / tw - 1) / (256 / tw), p.tiles_x = (W + tw - 1) / tw;
  return p;
}

int main() {
  struct Shape { int B, H, W, C; } shapes[] = {{128, 38, 60, 256}, {64, 76, 120, 128}, {32, 38, 60, 256}, {32, 76, 120, 128}, {32, 19, 30, 512}};
  for (auto sh : shapes) {
    const size_t na = (size_t)sh.B * sh.H * sh.W * sh.C, nw = (size_t)sh.C * 9 * sh.C;
    u16 *A, *O, *Wp;
    hipMalloc(&A, na * 2);
    hipMalloc(&O, na * 2);
    hipMalloc(&Wp, nw * 2);
    // pseudo-random 16-bit patterns of moderate magnitude (zero operands let the chip clock higher: MI355X_MICROARCH.md DVFS notes)
    {
      u16* h = (u16*)malloc((na > nw ? na : nw) * 2);
      unsigned s = 12345;
      for (size_t i = 0; i < na; i++) { s = s * 1664525u + 1013904223u; h[i] = (u16)(0x3000 + ((s >> 16) & 0x0FFF) + ((s >> 3) & 0x8000)); }
      hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice);
      for (size_t i = 0; i < nw; i++) { s = s * 1664525u + 1013904223u; h[i] = (u16)(0x2800 + ((s >> 16) & 0x07FF) + ((s >> 3) & 0x8000)); }
      hipMemcpy(Wp, h, nw * 2, hipMemcpyHostToDevice);
      free(h);
    }
    const bool big = sh.H * sh.W >= 2000;
    const int tw = big ? 16 : 32;
    const C3P p = make(sh.B, sh.H, sh.W, sh.C, tw, A, O, Wp);
    const int nitems = sh.B * p.tiles_y * p.tiles_x * (sh.C / 128);
    const double steps = (double)nitems / 256 * (sh.C / 64) * 9;  // average (segment, tap) steps per workgroup
    printf("%d -> %d @ %dx%d, B = %d: %d items, %.1f steps per workgroup\n", sh.C, sh.C, sh.H, sh.W, sh.B, nitems, steps);
#define ONE(D, what)                                                                                                                      \
  {                                                                                                                                        \
    const float mw = big ? time_kernel(k_conv3x3w<128, 16, D>, 1024, p) : time_kernel(k_conv3x3w<128, 32, D>, 1024, p);                   \
    const double cw = last_clock();                                                                                                        \
    const float mv = big ? time_kernel(k_conv3x3v<128, 16, D>, 512, p) : time_kernel(k_conv3x3v<128, 32, D>, 512, p);                     \
    const double cv = last_clock();                                                                                                        \
    const float ms = big ? time_kernel(k_conv3x3s<128, 16, D>, 512, p) : time_kernel(k_conv3x3s<128, 32, D>, 512, p);                     \
    const double cs = last_clock();                                                                                                        \
    printf("  diag %3d %-36s w %7.1f us %5.3f /step %4.2f GHz | v %7.1f us %5.3f /step %4.2f GHz | s %7.1f us %5.3f /step %4.2f GHz\n",  \
           D, what, mw * 1e3, mw * 1e3 / steps, cw, mv * 1e3, mv * 1e3 / steps, cv, ms * 1e3, ms * 1e3 / steps, cs);                       \
    fflush(stdout);                                                                                                                        \
  }
    ONE(0, "full kernel")
    ONE(32, "no epilogue")
    ONE(12, "W + halo DMA from the zero line")
    ONE(44, "zero-line DMA, no epilogue")
    ONE(4, "W DMA from the zero line")
    ONE(8, "halo DMA from the zero line")
    ONE(1, "no MFMA (fragment reads kept)")
    ONE(13, "no MFMA, zero-line DMA")
    ONE(2, "no fragment reads, no MFMA")
    ONE(46, "barriers + zero-line DMA issue only")
#undef ONE
    hipFree(A);
    hipFree(O);
    hipFree(Wp);
  }
  return 0;
}
