// scripts/hbm_write.hip — how fast does the MI355X take the stores of k_update_z's predicted planes?  Six fp32 fields of 512^3 (3.2 GB),
// (a) written front to back (float4 per lane), (b) in k_update_z's order: a workgroup of 64 x 4 columns marching 64 planes, a 256-byte
// row segment per wave, field and plane; (c) the same with 8 rows per workgroup; (d) with non-temporal stores.
//   hipcc -O3 --offload-arch=gfx950 scripts/hbm_write.hip -o /tmp/hbm_write && /tmp/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 512;
constexpr size_t FS = (size_t)N * N * (N + 6);
__global__ __launch_bounds__(256) void k_linear(float4 *p, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
template <int ROWS, bool NT> __global__ __launch_bounds__(64 * ROWS) void k_march(float *p, float v) {
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int nbx = N / 64, nby = N / ROWS;
  unsigned b = blockIdx.x;
  b = (b & ~7u) | ((b + (b >> 3)) & 7u);
  const int bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
  const size_t col = (size_t)(by * ROWS + ly) * N + bx * 64 + lx;
  for (int z = bz * 64; z < bz * 64 + 64; z++)
#pragma unroll
    for (int m = 0; m < 6; m++) {
      float *q = p + m * FS + (size_t)(z + 3) * N * N + col;
      if (NT) __builtin_nontemporal_store(v + m, q); else *q = v + m;
    }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); f();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < 5; i++) f();
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  float *p; if (hipMalloc(&p, 6 * FS * 4) != hipSuccess) return 1;
  const double gb = 6.0 * N * N * N * 4 / 1e9;
  float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, (float4 *)p, 6 * FS / 4, 1.f); });
  printf("linear float4, 6 x %zu floats          %7.3f ms  %6.2f TB/s\n", FS, ms, 6.0 * FS * 4 / 1e9 / ms);
  ms = timeit([&] { hipLaunchKernelGGL((k_march<4, false>), dim3(8 * 128 * 8), dim3(256), 0, 0, p, 1.f); });
  printf("march 64 x 4 columns x 64 planes         %7.3f ms  %6.2f TB/s\n", ms, gb / ms);
  ms = timeit([&] { hipLaunchKernelGGL((k_march<4, true>), dim3(8 * 128 * 8), dim3(256), 0, 0, p, 1.f); });
  printf("march 64 x 4, non-temporal               %7.3f ms  %6.2f TB/s\n", ms, gb / ms);
  ms = timeit([&] { hipLaunchKernelGGL((k_march<8, false>), dim3(8 * 64 * 8), dim3(512), 0, 0, p, 1.f); });
  printf("march 64 x 8 columns x 64 planes         %7.3f ms  %6.2f TB/s\n", ms, gb / ms);
  ms = timeit([&] { hipLaunchKernelGGL((k_march<16, false>), dim3(8 * 32 * 8), dim3(1024), 0, 0, p, 1.f); });
  printf("march 64 x 16 columns x 64 planes        %7.3f ms  %6.2f TB/s\n", ms, gb / ms);
  return 0;
}
