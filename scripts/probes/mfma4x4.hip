// Register layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products), found with one-hot
// operands: for every (lane of A, lane of B) the lanes of D that become 1.
// hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma4x4.hip -o scripts/probes/_mfma4x4 && scripts/probes/_mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      if (d != 0.0) out[la * 64 + lb] = l;
    }
}
int main() {
  int *d, h[4096];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0xff, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf("  B%d->D%d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
