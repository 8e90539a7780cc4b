// Micro-benchmark of the gfx950 issue rates that bound k_scan: v_fma_f32, v_pk_fma_f32, v_cmp->SGPR,
// SALU (s_or_b64) and mixes.  Build: hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k_fma(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkfma(float* out, float a, float b) {
  f32x2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f32x2 aa = {a, a}, bb = {b, b};
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
  }
  f32x2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// pk_fma with an SGPR-pair scalar operand broadcast (as the compiler emits for k_scan)
__global__ void k_pkfma_sgpr(float* out, float a, float b) {
  f32x2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f32x2 bb = {b, b};
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n"
                 "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"((double)a), "v"(bb));
  }
  f32x2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_cmp(float* out, float a) {
  float x = threadIdx.x;
  unsigned long long acc = 0;
  for (int i = 0; i < ITER; i++) {
    unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
    asm volatile("v_cmp_lt_f32 %0, %8, %9\n v_cmp_lt_f32 %1, %8, %9\n v_cmp_lt_f32 %2, %8, %9\n v_cmp_lt_f32 %3, %8, %9\n"
                 "v_cmp_lt_f32 %4, %8, %9\n v_cmp_lt_f32 %5, %8, %9\n v_cmp_lt_f32 %6, %8, %9\n v_cmp_lt_f32 %7, %8, %9\n"
                 : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7) : "v"(x), "v"(a));
    acc |= m0 ^ m7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
__global__ void k_salu(float* out, unsigned long long a) {
  unsigned long long s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3;
  for (int i = 0; i < ITER; i++) {
    asm volatile("s_or_b64 %0, %0, %4\n s_or_b64 %1, %1, %4\n s_or_b64 %2, %2, %4\n s_or_b64 %3, %3, %4\n"
                 "s_or_b64 %0, %0, %4\n s_or_b64 %1, %1, %4\n s_or_b64 %2, %2, %4\n s_or_b64 %3, %3, %4\n"
                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s0 + s1 + s2 + s3);
}
// 8 VALU + n SALU interleaved
template <int NS>
__global__ void k_mix(float* out, float a, float b, unsigned long long sa) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long s0 = sa, s1 = sa + 1, s2 = sa + 2, s3 = sa + 3;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#pragma unroll
    for (int q = 0; q < NS / 2; q++) asm volatile("s_or_b64 %0, %0, %2\n s_or_b64 %1, %1, %2\n" : "+s"(s0), "+s"(s1) : "s"(sa));
    asm volatile("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#pragma unroll
    for (int q = 0; q < NS / 2; q++) asm volatile("s_or_b64 %0, %0, %2\n s_or_b64 %1, %1, %2\n" : "+s"(s2), "+s"(s3) : "s"(sa));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3);
}

template <typename F>
float time_it(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 8192);
  for (int wpc : {4, 8, 16, 32}) {   // waves per CU
    int blocks = 256 * wpc / 4;      // 256-thread blocks = 4 waves
    double winst = (double)blocks * 4 * ITER * 8;  // wave-instructions of the measured type
    float t;
    t = time_it([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("waves/CU=%2d  v_fma_f32      : %.3f ms  %.2f wave-instr/clk/CU(@2.4GHz)  %.1f TFLOP/s\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9, winst * 64 * 2 / (t * 1e-3) / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("waves/CU=%2d  v_pk_fma_f32   : %.3f ms  %.2f wave-instr/clk/CU  %.1f TFLOP/s\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9, winst * 64 * 4 / (t * 1e-3) / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(k_pkfma_sgpr, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("waves/CU=%2d  v_pk_fma_f32(s): %.3f ms  %.2f wave-instr/clk/CU  %.1f TFLOP/s\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9, winst * 64 * 4 / (t * 1e-3) / 1e12);
    t = time_it([&] { hipLaunchKernelGGL(k_cmp, dim3(blocks), dim3(256), 0, 0, out, 3.f); });
    printf("waves/CU=%2d  v_cmp->sgpr    : %.3f ms  %.2f wave-instr/clk/CU\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9);
    t = time_it([&] { hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(256), 0, 0, out, 3ull); });
    printf("waves/CU=%2d  s_or_b64       : %.3f ms  %.2f wave-instr/clk/CU\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9);
    t = time_it([&] { hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3ull); });
    printf("waves/CU=%2d  8 fma + 4 salu : %.3f ms  %.2f valu-instr/clk/CU\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9);
    t = time_it([&] { hipLaunchKernelGGL(k_mix<8>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3ull); });
    printf("waves/CU=%2d  8 fma + 8 salu : %.3f ms  %.2f valu-instr/clk/CU\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9);
    t = time_it([&] { hipLaunchKernelGGL(k_mix<16>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3ull); });
    printf("waves/CU=%2d  8 fma + 16 salu: %.3f ms  %.2f valu-instr/clk/CU\n", wpc, t, winst / (t * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
