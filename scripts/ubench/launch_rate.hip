// Dependent-launch throughput of hipGraphs on one and on several streams (MI355X): how many kernel boundaries per
// second the device sustains when 1, 2, 4, 8 independent chains run concurrently.  The kernels do (almost) nothing,
// with the grid of the per-iteration kernels of the optimiser (1264 blocks x 128 threads).
// Build: hipcc --offload-arch=gfx950 -O3 launch_rate.hip -o launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_nop(int* p, int work) {
  if (work && threadIdx.x == 0) {  // `work` dependent ~1 us round trips per block
    int v = blockIdx.x;
    for (int i = 0; i < work; i++) v = p[v & 1023];
    if (v == -123) p[0] = v;
  }
}
int main() {
  int* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  const int CHAIN = 64;
  for (int blocks : {16, 1264}) {
    for (int work : {0, 8}) {
      for (int S : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(S);
        std::vector<hipGraphExec_t> ge(S);
        for (int s = 0; s < S; s++) {
          hipStreamCreate(&st[s]);
          hipGraph_t g;
          hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
          for (int i = 0; i < CHAIN; i++) hipLaunchKernelGGL(k_nop, dim3(blocks), dim3(128), 0, st[s], d, work);
          hipStreamEndCapture(st[s], &g);
          hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
          hipGraphDestroy(g);
        }
        for (int s = 0; s < S; s++) hipGraphLaunch(ge[s], st[s]);
        hipDeviceSynchronize();
        const int REP = 20;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; r++)
          for (int s = 0; s < S; s++) hipGraphLaunch(ge[s], st[s]);
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("blocks %4d work %d streams %d: %.2f us per launch per stream, %.2f us per launch overall\n", blocks, work, S,
               us / (REP * CHAIN), us / (REP * CHAIN * S));
        for (int s = 0; s < S; s++) {
          hipGraphExecDestroy(ge[s]);
          hipStreamDestroy(st[s]);
        }
      }
    }
  }
  return 0;
}
