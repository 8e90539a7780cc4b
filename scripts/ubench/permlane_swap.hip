// What v_permlane32_swap / v_permlane16_swap (gfx950) return for vdst = src = lane id: used for the cross-row steps of
// wave butterflies without ds_bpermute.   hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  const int x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  o[threadIdx.x] = a[0];
  o[64 + threadIdx.x] = a[1];
  o[128 + threadIdx.x] = b[0];
  o[192 + threadIdx.x] = b[1];
}
int main() {
  int* d;
  hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32[0]", "swap32[1]", "swap16[0]", "swap16[1]"};
  for (int r = 0; r < 4; r++) {
    printf("%s:", names[r]);
    for (int l = 0; l < 64; l += 8) printf(" l%d=%d", l, h[64 * r + l]);
    printf("\n");
  }
  return 0;
}
