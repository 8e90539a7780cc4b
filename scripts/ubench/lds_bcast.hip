// LDS broadcast-read throughput on gfx950: every lane of a wave reads the same 16-byte (or 4-byte) word.
// Build: hipcc --offload-arch=gfx950 -O3 lds_bcast.hip -o lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int stride) {
  __shared__ f32x4 buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  float acc = 0;
  int idx = wave * 8;
  for (int i = 0; i < ITER; i++) {
    if (MODE == 0) {  // 4 x b128, uniform address
      const f32x4 a = buf[idx & 1023], b = buf[(idx + 1) & 1023], c = buf[(idx + 2) & 1023], d = buf[(idx + 3) & 1023];
      acc += a.x + b.y + c.z + d.w;
    } else if (MODE == 1) {  // 16 lanes read one dword each (64 B), others idle
      if ((threadIdx.x & 63) < 16) acc += ((const float*)buf)[((idx * 4) & 4095) + (threadIdx.x & 15)];
    } else {  // 4 x b128, per-lane distinct addresses (conflict-free streaming)
      const int l = threadIdx.x & 63;
      const f32x4 a = buf[(idx + l) & 1023], b = buf[(idx + 64 + l) & 1023], c = buf[(idx + 128 + l) & 1023], d = buf[(idx + 192 + l) & 1023];
      acc += a.x + b.y + c.z + d.w;
    }
    idx += stride;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, threads>>>(out, 4);
  hipEventRecord(a);
  k<MODE><<<256, threads>>>(out, 4);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: (threads/64) waves x ITER iterations x (4 reads | 1 read)
  const double instr = (double)(threads / 64) * ITER * (MODE == 1 ? 1 : 4);
  printf("%-28s waves/CU=%2d  %.3f ms  %.2f ns per LDS instr per CU (%.1f clk @2.4GHz)\n", name, threads / 64, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
  hipFree(out);
}
int main() {
  for (int t : {256, 512, 1024}) {
    run<0>("b128 broadcast", t);
    run<1>("b32 x16 lanes", t);
    run<2>("b128 distinct", t);
  }
  return 0;
}
