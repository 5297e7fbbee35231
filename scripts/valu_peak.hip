// valu_peak.hip — the VALU issue ceilings DESIGN.md prices the VALU-bound kernels against, measured on the box:
// wave-instructions per second of v_fma_f32 (full rate), v_rcp_f32 (quarter rate) and v_pk_fma_f32 (two fp32 per
// lane).  Build + run: scripts/valu_peak.sh (hipcc --offload-arch=gfx950, no other dependency).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ITER = 4096, UNROLL = 16;

template <int KIND> __global__ __launch_bounds__(256) void k(float *out, float a, float b) {
  float x[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; i++) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
    }
    if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < UNROLL; i += 2)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<float2 *>(&x[i])) : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < UNROLL; i++) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND> static double run(float *d, int blocks, const char *name, double instr_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 1e-3f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 1e-3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4, winstr = waves * ITER * instr_per_iter;
  const double rate = winstr / (ms * 1e-3);
  printf("%-14s %8.3f ms  %.3e wave-instructions/s  (%.1f T lane-ops/s)\n", name, ms, rate, rate * 64 / 1e12);
  return rate;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  float *d;
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  printf("# %s, %d CUs, clock %d MHz, %d workgroups x 256 threads, %d x %d instructions per thread\n", p.gcnArchName,
         p.multiProcessorCount, p.clockRate / 1000, blocks, ITER, UNROLL);
  const double f = run<0>(d, blocks, "v_fma_f32", UNROLL);
  const double r = run<1>(d, blocks, "v_rcp_f32", UNROLL);
  const double q = run<2>(d, blocks, "v_pk_fma_f32", UNROLL / 2);
  printf("rcp / fma issue ratio %.2f; pk_fma lane-ops vs fma %.2fx\n", r / f, 2 * q / f);
  hipFree(d);
  return 0;
}
