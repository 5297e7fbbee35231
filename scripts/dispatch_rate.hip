// scripts/dispatch_rate.hip — how fast does the MI355X start workgroups that do (almost) nothing?  The split 3D step's k_flux_xy is one
// 512-thread workgroup per 32 x 16 tile; a tile whose staging finds it uniform leaves at once, and the launch is then bound by this.
//   hipcc -O3 --offload-arch=gfx950 scripts/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int LDS, int VG> __global__ __launch_bounds__(NT) void k_empty(const unsigned *flag, float *out) {
  __shared__ float s[LDS > 0 ? LDS / 4 : 1];
  if (*flag != 0u) {   // never: keeps the LDS array and a few registers alive
    float v[VG];
    for (int i = 0; i < VG; i++) v[i] = out[threadIdx.x + i * NT];
    s[threadIdx.x % (LDS > 0 ? LDS / 4 : 1)] = v[0];
    __syncthreads();
    float a = s[(threadIdx.x * 7) % (LDS > 0 ? LDS / 4 : 1)];
    for (int i = 0; i < VG; i++) a = a * v[i] + v[(i + 1) % VG];
    out[blockIdx.x * NT + threadIdx.x] = a;
  }
}
template <int NT, int LDS, int VG> void run(const char *name, unsigned nwg, const unsigned *flag, float *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_empty<NT, LDS, VG>), dim3(nwg), dim3(NT), 0, 0, flag, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_empty<NT, LDS, VG>), dim3(nwg), dim3(NT), 0, 0, flag, out);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  printf("%-34s %7u wg x %4d thr  lds %6d  %8.1f us  %6.2f ns/wg  %6.3f ns/wave\n", name, nwg, NT, LDS, ms * 1e3, ms * 1e6 / nwg, ms * 1e6 / nwg / (NT / 64));
}
int main() {
  unsigned *flag; float *out;
  hipMalloc(&flag, 4); hipMemset(flag, 0, 4); hipMalloc(&out, 1 << 28);
  const unsigned N = 262144;
  run<64, 0, 4>("64 thr, no LDS", N * 8, flag, out);
  run<128, 0, 4>("128 thr, no LDS", N * 4, flag, out);
  run<256, 0, 4>("256 thr, no LDS", N * 2, flag, out);
  run<512, 0, 4>("512 thr, no LDS", N, flag, out);
  run<1024, 0, 4>("1024 thr, no LDS", N / 2, flag, out);
  run<256, 19000, 4>("256 thr, 19 KB LDS", N * 2, flag, out);
  run<512, 38000, 4>("512 thr, 38 KB LDS", N, flag, out);
  run<512, 38000, 48>("512 thr, 38 KB LDS, ~64 VGPR", N, flag, out);
  run<1024, 38000, 4>("1024 thr, 38 KB LDS", N / 2, flag, out);
  run<1024, 64000, 4>("1024 thr, 64 KB LDS", N / 2, flag, out);
  run<256, 38000, 4>("256 thr, 38 KB LDS", N, flag, out);
  run<64, 38000, 4>("64 thr, 38 KB LDS", N, flag, out);
  run<64, 0, 4>("64 thr, no LDS, N wg", N, flag, out);
  return 0;
}
