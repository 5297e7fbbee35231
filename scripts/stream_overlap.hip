// scripts/stream_overlap.hip — do kernels of two HIP streams run side by side on this box?  (a) two 1-workgroup spinners, (b) a spinner
// that fills 5 workgroups per CU with 96 VGPRs + 30 KB LDS beside a store-only kernel, (c) the same two back to back on one stream.
//   hipcc -O3 --offload-arch=gfx950 scripts/stream_overlap.hip -o /tmp/so && /tmp/so
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(long long cycles, float *out) {
  const long long t0 = __builtin_readcyclecounter();
  float a = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < cycles) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = a;
}
__global__ __launch_bounds__(256, 5) void k_heavy(long long cycles, float *out) {
  __shared__ float lds[30 * 256];
  float v[60];
  for (int i = 0; i < 60; i++) v[i] = threadIdx.x * 0.001f + i;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles)
    for (int i = 0; i < 60; i++) v[i] = v[i] * 1.0001f + v[(i + 7) % 60];
  float a = 0; for (int i = 0; i < 60; i++) a += v[i];
  lds[threadIdx.x] = a; __syncthreads();
  if (lds[(threadIdx.x + 1) & 255] == 12345.f) out[0] = a;
}
__global__ __launch_bounds__(256) void k_store(float *p, size_t n_per_wg) {
  float *q = p + (size_t)blockIdx.x * n_per_wg;
  for (size_t i = threadIdx.x; i < n_per_wg; i += 256) q[i] = 1.f;
}
int main() {
  float *out, *buf; hipMalloc(&out, 64); const size_t N = (size_t)600 << 20; hipMalloc(&buf, N * 4);
  hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  auto ms = [&](hipEvent_t x, hipEvent_t y) { float m; hipEventElapsedTime(&m, x, y); return m; };
  const long long CY = 100 * 1000 * 10;   // ~1 ms at 100 MHz counter? (readcyclecounter = s_memtime, 100 MHz)
  for (int rep = 0; rep < 2; rep++) {
    hipDeviceSynchronize();
    hipEventRecord(a, s1);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 100000, out);
    hipEventRecord(b, s1); hipEventSynchronize(b);
    const float one = ms(a, b);
    hipEventRecord(a, s1);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 100000, out);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, 100000, out);
    hipEventRecord(c, s2); hipStreamWaitEvent(s1, c, 0);
    hipEventRecord(b, s1); hipEventSynchronize(b);
    printf("spinner alone %.3f ms; two spinners on two streams %.3f ms\n", one, ms(a, b));
    // heavy alone, store alone, both on one stream, both on two streams
    const unsigned NH = 1280 * 2, NS = 4096;
    hipEventRecord(a, s1); hipLaunchKernelGGL(k_heavy, dim3(NH), dim3(256), 0, s1, 30000, out); hipEventRecord(b, s1); hipEventSynchronize(b);
    const float th = ms(a, b);
    hipEventRecord(a, s1); hipLaunchKernelGGL(k_store, dim3(NS), dim3(256), 0, s1, buf, N / NS); hipEventRecord(b, s1); hipEventSynchronize(b);
    const float ts = ms(a, b);
    hipEventRecord(a, s1);
    hipLaunchKernelGGL(k_heavy, dim3(NH), dim3(256), 0, s1, 30000, out);
    hipLaunchKernelGGL(k_store, dim3(NS), dim3(256), 0, s1, buf, N / NS);
    hipEventRecord(b, s1); hipEventSynchronize(b);
    const float t1 = ms(a, b);
    hipEventRecord(a, s1);
    hipEventRecord(c, s1); hipStreamWaitEvent(s2, c, 0);
    hipLaunchKernelGGL(k_heavy, dim3(NH), dim3(256), 0, s1, 30000, out);
    hipLaunchKernelGGL(k_store, dim3(NS), dim3(256), 0, s2, buf, N / NS);
    hipEventRecord(c, s2); hipStreamWaitEvent(s1, c, 0);
    hipEventRecord(b, s1); hipEventSynchronize(b);
    printf("heavy (2 rounds of 5 wg/CU) %.3f ms, store 2.5 GB %.3f ms (%.2f TB/s); one stream %.3f ms; two streams %.3f ms\n", th, ts, N * 4 / 1e9 / ts, t1, ms(a, b));
  }
  return 0;
}
