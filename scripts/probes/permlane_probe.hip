// Prints the lane maps of v_permlane16_swap / v_permlane32_swap on gfx950 (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
  const int l = threadIdx.x;
  auto r16 = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
  auto r32 = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
  o[l] = r16[0];
  o[64 + l] = r16[1];
  o[128 + l] = r32[0];
  o[192 + l] = r32[1];
}
int main()
{
  int *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"p16.dst", "p16.src", "p32.dst", "p32.src"};
  for (int a = 0; a < 4; a++) {
    printf("%s:", names[a]);
    for (int l = 0; l < 64; l++) printf(" %d", h[64 * a + l]);
    printf("\n");
  }
  return 0;
}
