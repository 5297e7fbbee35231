// lbm.hip — D2Q9 BGK lattice-Boltzmann collide-and-stream for gfx950 (MI355X); SURVEY §8f row 4.
//
// Reference (tau_lbm.cu:94-132): one thread per cell, 16x16 blocks, push scheme — read the cell's nine
// populations, relax towards equilibrium, write each to the q-neighbour (x periodic) unless that neighbour is
// solid or beyond the y walls (then keep it in the cell's own opposite slot: on-link bounce-back); a solid cell
// reflects its nine values in place.  Every slot of fout has exactly one writer, so the result is schedule
// independent and — compiled without FMA contraction, in the reference's association order — bit-identical
// to the CPU oracle.
//
// 72 B/cell compulsory (9 fp32 in, 9 fp32 out, + 1 mask byte), ~150 flops: pure HBM streaming.  What matters
// here is the shape of the accesses, not the arithmetic:
//   * a workgroup is 256 consecutive cells of ONE row (the reference's 16x16 tile makes every wave touch 4
//     rows x 64 B); each of the 9 loads and 9 stores of a wave is one contiguous 256-B run;
//   * the +-1 x-shift of six of the stores makes a run straddle a 128-B line shared with the neighbouring
//     workgroup: the linear block id is remapped so that each XCD walks a contiguous range of workgroups and the
//     two halves of such a line meet in the same L2 before they go to HBM;
//   * the nine loads are streaming (non-temporal) loads — nothing is ever read twice — which leaves L2 to the
//     stores that do meet there: 5.16 -> 5.54 TB/s.  Non-temporal STORES lose (4.7 TB/s): the straddling halves no
//     longer merge;
//   * no per-cell integer modulo (the periodic wrap is two selects), no constant-memory index tables.
#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <new>
#include <vector>

namespace lbm {

struct Args {
  const float *fin;
  float *fout;
  const uint8_t *solid;
  int nx, ny, nbx;          // nbx = workgroups per row
  size_t cells;
  float omega, drive;
};

__device__ __forceinline__ float feq(int ex, int ey, float w, float rho, float ux, float uy) { // :67-71
  const float cu = 3.0f * (ex * ux + ey * uy);
  const float u2 = ux * ux + uy * uy;
  return w * rho * (1.0f + cu + 0.5f * cu * cu - 1.5f * u2);
}

__global__ __launch_bounds__(256) void k_collide_stream(const Args A) {
  constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};   // :57-59
  constexpr int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
  constexpr int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
  constexpr float WQ[9] = {4.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f,
                           1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f};
  const unsigned b = tau::xcd_swizzle(blockIdx.x, gridDim.x);
  const int j = (int)(b / (unsigned)A.nbx);
  const int i = (int)(b - (unsigned)j * (unsigned)A.nbx) * 256 + (int)threadIdx.x;
  if (i >= A.nx) return;
  const size_t p = (size_t)j * A.nx + i;
  float local[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) local[q] = __builtin_nontemporal_load(&A.fin[q * A.cells + p]);
  if (A.solid[p]) {
#pragma unroll
    for (int q = 0; q < 9; ++q) A.fout[OPP[q] * A.cells + p] = local[q];
    return;
  }
  float rho = 0.0f, ux = 0.0f, uy = 0.0f;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    rho += local[q];
    ux += local[q] * EX[q];
    uy += local[q] * EY[q];
  }
  rho = fmaxf(rho, 1.0e-6f);
  ux = ux / rho + A.drive;
  uy /= rho;
  const int im = (i == 0) ? A.nx - 1 : i - 1, ip = (i == A.nx - 1) ? 0 : i + 1;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const float post = local[q] - A.omega * (local[q] - feq(EX[q], EY[q], WQ[q], rho, ux, uy));
    const int ni = (EX[q] == 0) ? i : (EX[q] > 0 ? ip : im);
    const int nj = j + EY[q];
    const size_t np = (size_t)nj * A.nx + ni;
    const bool blocked = (EY[q] != 0 && (nj < 0 || nj >= A.ny)) || (q != 0 && A.solid[(nj < 0 || nj >= A.ny) ? p : np]);
    if (blocked) A.fout[OPP[q] * A.cells + p] = post;
    else A.fout[q * A.cells + np] = post;
  }
}

// init_kernel, :72-92 — the shear profile (one sinf per row) comes from the host so that the start state is
// bit-identical to a libm evaluation
__global__ __launch_bounds__(256) void k_init(float *f, uint8_t *solid, const float *shear_row, int nx, int ny, int obstacle,
                                              float radius, float rho0) {
  constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
  constexpr int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
  constexpr float WQ[9] = {4.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f, 1.0f / 9.0f,
                           1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f, 1.0f / 36.0f};
  const size_t cells = (size_t)nx * ny;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < cells; p += (size_t)gridDim.x * 256) {
    const int j = (int)(p / nx), i = (int)(p - (size_t)j * nx);
    const float cx = 0.28f * nx, cy = 0.5f * ny;
    const float dx = i - cx, dy = j - cy;
    const bool wall = (j == 0 || j == ny - 1);
    const bool cyl = obstacle && (dx * dx + dy * dy < radius * radius);
    solid[p] = (wall || cyl) ? 1 : 0;
    const float shear = shear_row[j];
#pragma unroll
    for (int q = 0; q < 9; ++q) f[q * cells + p] = feq(EX[q], EY[q], WQ[q], rho0, shear, 0.0f);
  }
}

// render_kernel, :134-153
__global__ __launch_bounds__(256) void k_speed(const float *f, const uint8_t *solid, float *speed, size_t cells) {
  constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
  constexpr int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < cells; p += (size_t)gridDim.x * 256) {
    if (solid[p]) { speed[p] = -1.0f; continue; }
    float rho = 0.0f, ux = 0.0f, uy = 0.0f;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const float fq = f[q * cells + p];
      rho += fq;
      ux += fq * EX[q];
      uy += fq * EY[q];
    }
    speed[p] = hypotf(ux / rho, uy / rho);
  }
}

} // namespace lbm

// =====================================================================================
// C-ABI
// =====================================================================================
struct taulbm {
  taulbm_params p;
  int device;
  hipStream_t stream;
  bool own_stream;
  float *f[2];
  float *speed;
  uint8_t *solid;
  int cur;
  long step;
};

extern "C" void taulbm_params_default(taulbm_params *P) { // tau_lbm.cu:43-55
  P->nx = 512; P->ny = 256; P->obstacle = 1; P->tau = 0.56f; P->drive = 1.0e-6f; P->rho0 = 1.0f;
  P->obstacle_radius = 32.0f;
}

extern "C" int taulbm_create(taulbm_t **out, const taulbm_params *P, int device, void *stream) {
  if (!out || !P) return tau::fail("taulbm_create: null argument");
  if (P->nx < 1 || P->ny < 1) return tau::fail("taulbm_create: bad grid %dx%d", P->nx, P->ny);
  if (!(P->tau > 0.5f)) return tau::fail("taulbm_create: tau must exceed 0.5 (got %g)", (double)P->tau);
  TAU_HIP(hipSetDevice(device));
  taulbm *h = new (std::nothrow) taulbm();
  if (!h) return tau::fail("taulbm_create: out of host memory");
  tau::HandleGuard<taulbm> guard{h, taulbm_destroy};
  h->p = *P; h->device = device;
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  const size_t cells = (size_t)P->nx * P->ny;
  TAU_HIP(hipMalloc(&h->f[0], 9 * cells * sizeof(float)));
  TAU_HIP(hipMalloc(&h->f[1], 9 * cells * sizeof(float)));
  TAU_HIP(hipMalloc(&h->speed, cells * sizeof(float)));
  TAU_HIP(hipMalloc(&h->solid, cells));
  TAU_HIP(hipMemsetAsync(h->solid, 0, cells, h->stream));
  *out = guard.release();
  return 0;
}
extern "C" void taulbm_destroy(taulbm_t *h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  hipFree(h->f[0]); hipFree(h->f[1]); hipFree(h->speed); hipFree(h->solid);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int taulbm_init(taulbm_t *h) { // init_kernel :72-92 + the D2D copy :247
  TAU_HIP(hipSetDevice(h->device));
  const taulbm_params &P = h->p;
  std::vector<float> shear((size_t)P.ny);
  for (int j = 0; j < P.ny; j++) shear[j] = 0.015f * sinf(2.0f * 3.14159265f * j / (P.ny > 1 ? P.ny - 1 : 1));
  float *d = nullptr;
  TAU_HIP(hipMalloc(&d, shear.size() * sizeof(float)));
  TAU_HIP(hipMemcpyAsync(d, shear.data(), shear.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(lbm::k_init, dim3(2048), dim3(256), 0, h->stream, h->f[0], h->solid, (const float *)d, P.nx, P.ny, P.obstacle,
                     P.obstacle_radius, P.rho0);
  TAU_LAUNCH_CHECK("lbm::k_init");
  const size_t cells = (size_t)P.nx * P.ny;
  TAU_HIP(hipMemcpyAsync(h->f[1], h->f[0], 9 * cells * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  TAU_HIP(hipFree(d));
  h->cur = 0; h->step = 0;
  return 0;
}

extern "C" int taulbm_upload(taulbm_t *h, const float *f9, const uint8_t *solid) {
  TAU_HIP(hipSetDevice(h->device));
  const size_t cells = (size_t)h->p.nx * h->p.ny;
  if (f9) TAU_HIP(hipMemcpyAsync(h->f[h->cur], f9, 9 * cells * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (solid) TAU_HIP(hipMemcpyAsync(h->solid, solid, cells, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int taulbm_download(taulbm_t *h, float *f9, uint8_t *solid) {
  TAU_HIP(hipSetDevice(h->device));
  const size_t cells = (size_t)h->p.nx * h->p.ny;
  if (f9) TAU_HIP(hipMemcpyAsync(f9, h->f[h->cur], 9 * cells * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (solid) TAU_HIP(hipMemcpyAsync(solid, h->solid, cells, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int taulbm_state_ptrs(taulbm_t *h, float **f9, uint8_t **solid) {
  if (f9) *f9 = h->f[h->cur];
  if (solid) *solid = h->solid;
  return 0;
}
extern "C" int taulbm_set_drive(taulbm_t *h, float drive) { h->p.drive = drive; return 0; } // keys + / -, :283-284

extern "C" int taulbm_step_async(taulbm_t *h, int nsteps) { // loop body :262-265
  TAU_HIP(hipSetDevice(h->device));
  lbm::Args A;
  A.nx = h->p.nx; A.ny = h->p.ny; A.nbx = (A.nx + 255) / 256; A.cells = (size_t)A.nx * A.ny;
  A.omega = 1.0f / h->p.tau; A.drive = h->p.drive; A.solid = h->solid;
  for (int s = 0; s < nsteps; s++) {
    A.fin = h->f[h->cur]; A.fout = h->f[h->cur ^ 1];
    hipLaunchKernelGGL(lbm::k_collide_stream, dim3((unsigned)(A.nbx * A.ny)), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("lbm::k_collide_stream");
    h->cur ^= 1;   // std::swap(d_f0, d_f1), :264
    h->step++;
  }
  return 0;
}
extern "C" int taulbm_sync(taulbm_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int taulbm_step(taulbm_t *h, int nsteps) {
  if (taulbm_step_async(h, nsteps)) return 1;
  return taulbm_sync(h);
}
extern "C" int taulbm_speed(taulbm_t *h, float *host_speed) { // render_kernel + D2H, :267-269
  TAU_HIP(hipSetDevice(h->device));
  const size_t cells = (size_t)h->p.nx * h->p.ny;
  hipLaunchKernelGGL(lbm::k_speed, dim3(2048), dim3(256), 0, h->stream, (const float *)h->f[h->cur], (const uint8_t *)h->solid,
                     h->speed, cells);
  TAU_LAUNCH_CHECK("lbm::k_speed");
  if (host_speed) TAU_HIP(hipMemcpyAsync(host_speed, h->speed, cells * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int64_t taulbm_steps_done(taulbm_t *h) { return (int64_t)h->step; }
