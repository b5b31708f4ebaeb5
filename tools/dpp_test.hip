#include <hip/hip_runtime.h>
#include <cstdio>
#define MPDATA_NS tst
#include "../codesign-kernels_amd/csrc/mpdata_kernel_v2_body.h"
__global__ void k(double* o) {
  const int lane = threadIdx.x;
  double x = 100.0 + lane;
  o[lane] = tst::v2::shift_dn_clamped(x, (lane % 32) == 0);
  o[64 + lane] = tst::v2::shift_dn(x);
  o[128 + lane] = tst::v2::shift_up(x);
  o[192 + lane] = tst::v2::shift_dn_clamped(x, (lane % 16) == 0);
  o[256 + lane] = tst::v2::shift_dn_clamped(x, lane == 0);
  const bool ge = (lane % 32) >= 26;
  o[320 + lane] = tst::v2::shift_up_clamped(x, ge);
  double y = x * 2.0 + o[lane];  // VALU-produced value feeding DPP
  o[384 + lane] = tst::v2::shift_up_clamped(y, ge) - o[lane];
}
int main() {
  double* d; hipMalloc(&d, 448 * 8); hipMemset(d, 0, 448 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[448]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[7] = {"dn_c32", "dn", "up", "dn_c16", "dn_c64", "up_c", "up_c2"};
  for (int t = 0; t < 7; ++t) { printf("%-7s:", names[t]); for (int l = 0; l < 64; ++l) printf(" %d", (int)h[t * 64 + l] - 100); printf("\n"); }
  return 0;
}
