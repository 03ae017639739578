// Prints what v_permlane32_swap / v_permlane16_swap do on gfx950 (lane i holds 100 + i in a, 200 + i in b).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0], out[64 + l] = r[1];
  a = 100 + l, b = 200 + l;
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = s[0], out[192 + l] = s[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
  for (int v = 0; v < 4; v++) {
    printf("%s:", nm[v]);
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[v * 64 + i]);
    printf("\n");
  }
  return 0;
}
