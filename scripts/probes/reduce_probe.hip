// Checks the cross-lane reductions of mtp_kernel_common.hpp against direct sums (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../lammps_mtp_kokkos_amd/csrc/mtp_kernel_common.hpp"
__global__ void k(double *o)
{
  const int lane = threadIdx.x;
  double v32[32], v16[16];
  for (int u = 0; u < 32; u++) v32[u] = 1000.0 * u + lane;
  for (int u = 0; u < 16; u++) v16[u] = 1000.0 * u + lane;
  Butterfly<32>::run(v32, lane);
  Butterfly<16>::run(v16, lane);
  o[lane] = v32[0];
  o[64 + lane] = v16[0];
  o[128 + lane] = pair_sum16((double) lane);
  o[192 + lane] = pair_sum32((double) lane);
  o[256 + lane] = wave_sum((double) lane);
  o[320 + lane] = partner_f64<16>((double) lane, lane);
  o[384 + lane] = partner_f64<32>((double) lane, lane);
}
int main()
{
  double *d, h[448];
  (void) hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void) hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int half = l & 32, row = l & 48;
    double e32 = 0, e16 = 0;
    for (int m = 0; m < 32; m++) e32 += 1000.0 * (l & 31) + (half + m);
    for (int m = 0; m < 16; m++) e16 += 1000.0 * (l & 15) + (row + m);
    if (h[l] != e32) { bad++; printf("B32 lane %d got %g want %g\n", l, h[l], e32); }
    if (h[64 + l] != e16) { bad++; printf("B16 lane %d got %g want %g\n", l, h[64 + l], e16); }
    if (h[128 + l] != (double) (l + (l ^ 16))) { bad++; printf("ps16 lane %d got %g\n", l, h[128 + l]); }
    if (h[192 + l] != (double) (l + (l ^ 32))) { bad++; printf("ps32 lane %d got %g\n", l, h[192 + l]); }
    if (h[256 + l] != 2016.0) { bad++; printf("wsum lane %d got %g\n", l, h[256 + l]); }
    if (h[320 + l] != (double) (l ^ 16)) { bad++; printf("p16 lane %d got %g\n", l, h[320 + l]); }
    if (h[384 + l] != (double) (l ^ 32)) { bad++; printf("p32 lane %d got %g\n", l, h[384 + l]); }
  }
  printf("reduce_probe: %d mismatches\n", bad);
  return bad != 0;
}
