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
#include <type_traits>
#include <cstdlib>
#include <new>
#include <vector>

namespace lbm {

struct Args {
  const float *fin;
  float *fout;
  const uint8_t *solid;
  int nx, ny, nbx;          // nbx = workgroups per row
  int nstrips, nchunks, rows; // fused kernel: strips of 64 - 2K owned columns, chunks of `rows` output rows
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

// -------------------------------------------------------------------------------------------------
// K steps per pass (temporal fusion, as st2::k_fused for the stencils): 72 B per update is the single-step floor and
// leaves 60 % of the VALU idle.  The push scheme is restated as a PULL so that a level can be assembled row by row
// from the post-collision rows of the level below:
//     out_q(n) = post_q(n - e_q)        if n - e_q is inside the y range and neither it nor n is solid
//              = post_opp(q)(n)         otherwise          (post = f for a solid cell: it reflects in place)
// which is the same single-writer assignment the push makes (a blocked push IS the second line).  A wave holds one
// column per lane, 64 - 2K of them owned, the outer K lanes on each side being the halo of the K steps; per time
// level it keeps the post-collision populations of three consecutive rows (+ their mask) in registers.  Each trip
// loads one row of level t, collides it once, and every stage assembles one row of the next level, collides it and
// hands it up; only level t+K is written.  Collision and assembly use the very expressions of the single-step kernel
// (no FMA contraction), so the result is bit-identical to K single steps.
struct LRow { float p[9]; int m; };   // post-collision populations of one row (this lane's column); m: 0 fluid, 1 solid, 2 no such row

__device__ __forceinline__ void lbm_collide(const Args &A, LRow &r) {   // in: populations, out: post-collision
  if (r.m != 0) return;
  float *p = r.p;
  // Moments and equilibria of :113-128, in the reference's association order but without the operations a strict
  // IEEE compiler may not drop and that cannot change a bit of a FINITE state: products with the 0 / +-1 lattice components,
  // additions of the resulting signed zeros, and — since e_opp = -e_q — the second evaluation of cu^2/2 for the
  // opposite direction ((0.5 * -cu) * -cu is the same number).  The pass is VALU bound, so this is ~25 % of it.
  float rho = 0.0f + p[0];
  rho += p[1]; rho += p[2]; rho += p[3]; rho += p[4]; rho += p[5]; rho += p[6]; rho += p[7]; rho += p[8];
  float ux = p[1];                      // (0 + 0*p0) + p1
  ux -= p[3]; ux += p[5]; ux -= p[6]; ux -= p[7]; ux += p[8];
  float uy = p[2];
  uy -= p[4]; uy += p[5]; uy += p[6]; uy -= p[7]; uy -= p[8];
  rho = fmaxf(rho, 1.0e-6f);
  ux = ux / rho + A.drive;
  uy /= rho;
  const float u2 = ux * ux + uy * uy, t = 1.5f * u2;
  const float w0 = (4.0f / 9.0f) * rho, w1 = (1.0f / 9.0f) * rho, w2 = (1.0f / 36.0f) * rho;
  auto relax = [&](int q, float feq) { p[q] = p[q] - A.omega * (p[q] - feq); };
  relax(0, w0 * (1.0f - t));                                            // cu = 0: (1 + 0 + 0) - 1.5 u2
  auto pair = [&](int qa, int qb, float w, float e) {                    // directions +e and -e
    const float cu = 3.0f * e, h = 0.5f * cu * cu;
    relax(qa, w * (1.0f + cu + h - t));
    relax(qb, w * (1.0f - cu + h - t));
  };
  pair(1, 3, w1, ux);
  pair(2, 4, w1, uy);
  pair(5, 7, w2, ux + uy);
  pair(8, 6, w2, ux - uy);              // e_8 = (1, -1): 1*ux + (-1)*uy ; e_6 = (-1, 1) is its negative
}

// populations of the middle row at the next level, from the post-collision rows above (j-1), at (j) and below (j+1)
__device__ __forceinline__ void lbm_assemble(const LRow &up, const LRow &cu, const LRow &dn, LRow &o) {
  constexpr int EX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
  constexpr int EY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
  constexpr int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
  const int mm = up.m | (cu.m << 2) | (dn.m << 4);           // the three masks of this column in one word
  const int ml = __shfl_up(mm, 1, 64), mr = __shfl_down(mm, 1, 64);
  o.m = cu.m;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const LRow &src = (EY[q] > 0) ? up : (EY[q] < 0 ? dn : cu);             // row of c = n - e_q
    const int sh = (EY[q] > 0) ? 0 : (EY[q] < 0 ? 4 : 2);
    float v = src.p[q];
    int mc = mm;
    if (EX[q] > 0) { v = __shfl_up(v, 1, 64); mc = ml; }                     // c is the column to the left
    if (EX[q] < 0) { v = __shfl_down(v, 1, 64); mc = mr; }
    const bool from_c = (((mc >> sh) & 3) == 0) && (cu.m == 0);              // c exists and is fluid, n is fluid
    o.p[q] = from_c ? v : cu.p[OPP[q]];
  }
}

template <int K>
__global__ __launch_bounds__(64 * 4) void k_fused(const Args A) {
  static_assert(K >= 2 && K <= 4, "");
  constexpr int STRIDE = 64 - 2 * K;                                         // owned columns per wave
  const int lane = threadIdx.x & 63;
  const unsigned nwork = (unsigned)(A.nstrips * A.nchunks);
  const unsigned wid = tau::xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (wid >= nwork) return;
  const int strip = (int)(wid % (unsigned)A.nstrips);
  const int chunk = (int)(wid / (unsigned)A.nstrips);
  const int xc = strip * STRIDE - K + lane;
  int xw = xc;
  while (xw < 0) xw += A.nx;
  while (xw >= A.nx) xw -= A.nx;
  const bool owner = lane >= K && lane < 64 - K && xc < A.nx && xc < (strip + 1) * STRIDE;
  const int j0 = chunk * A.rows, j1 = min(j0 + A.rows, A.ny);

  LRow st[K][3];
#pragma unroll
  for (int s = 0; s < K; s++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      st[s][k].m = 2;
#pragma unroll
      for (int q = 0; q < 9; ++q) st[s][k].p[q] = 0.f;
    }
  // trip r: level-t row r enters; stage s then holds post-collision rows (r-2-s, r-1-s, r-s) of level t+s; the row of
  // level t+K produced by the trip is r-K.  The first 2K trips only fill the pipeline (nothing is stored before j0).
  // The three rows of a stage live in a ROTATING window (newest in slot PH, oldest in PH+1, middle in PH+2, mod 3; the
  // loop is unrolled over the three phases), so no row is ever copied: shifting cost ~36 register moves per update.
  auto trip = [&](auto ph, int r) {
    constexpr int PH = decltype(ph)::value;
    constexpr int NEW = PH % 3, OLD = (PH + 1) % 3, MID = (PH + 2) % 3;
    {
      LRow &n = st[0][NEW];
      if (r >= 0 && r < A.ny) {
        const size_t p = (size_t)r * A.nx + xw;
        n.m = A.solid[p];
#pragma unroll
        for (int q = 0; q < 9; ++q) n.p[q] = __builtin_nontemporal_load(&A.fin[q * A.cells + p]);
      } else {
        n.m = 2;
      }
      lbm_collide(A, n);
    }
#pragma unroll
    for (int s = 0; s < K; s++) {
      if (s + 1 < K) {
        LRow &o = st[s + 1][NEW];                                              // overwrites that stage's oldest row
        lbm_assemble(st[s][OLD], st[s][MID], st[s][NEW], o);                   // level t+s+1, row r-1-s
        lbm_collide(A, o);
      } else {
        LRow out;
        lbm_assemble(st[s][OLD], st[s][MID], st[s][NEW], out);
        const int j = r - K;
        if (owner && j >= j0) {
          const size_t p = (size_t)j * A.nx + xc;
#pragma unroll
          for (int q = 0; q < 9; ++q) A.fout[q * A.cells + p] = out.p[q];
        }
      }
    }
  };
  const int r0 = j0 - K, r1 = j1 + K;
  for (int r = r0; r < r1; r += 3) {
    trip(std::integral_constant<int, 0>{}, r);
    if (r + 1 < r1) trip(std::integral_constant<int, 1>{}, r + 1);
    if (r + 2 < r1) trip(std::integral_constant<int, 2>{}, r + 2);
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
  // passes of K fused time levels.  Measured: 4 levels pay from ~8 M cells (8192^2: 77 -> 167 G updates/s), 3 levels
  // from ~2 M (2048^2: 82 -> 127); below that a pass is a long serial march of few waves and the single-step push
  // kernel (one dispatch per step) is faster.  TAU_LBM_LEVELS = 1..4 overrides.
  static const int kenv = getenv("TAU_LBM_LEVELS") ? atoi(getenv("TAU_LBM_LEVELS")) : 0;
  const int kmax = kenv > 0 ? kenv : (A.cells >= ((size_t)1 << 23) ? 4 : (A.cells >= ((size_t)1 << 21) ? 3 : 1));
  static const int frows = getenv("TAU_LBM_FROWS") ? atoi(getenv("TAU_LBM_FROWS")) : 0;
  int s = 0;
  while (s < nsteps) {
    A.fin = h->f[h->cur]; A.fout = h->f[h->cur ^ 1];
    const int left = nsteps - s;
    const int K = (kmax < 2 || kmax > 4) ? 1 : (left >= kmax ? kmax : (left >= 2 ? left : 1));
    if (K >= 2) {
      A.nstrips = (A.nx + (64 - 2 * K) - 1) / (64 - 2 * K);
      long r = frows > 0 ? frows : (long)A.ny * A.nstrips / 4096;
      A.rows = frows > 0 ? frows : (r >= 32 ? 32 : (r < 4 ? 4 : (int)r));
      A.nchunks = (A.ny + A.rows - 1) / A.rows;
      const unsigned nwork = (unsigned)(A.nstrips * A.nchunks), nb = (nwork + 3) / 4;
      if (K == 2) hipLaunchKernelGGL(lbm::k_fused<2>, dim3(nb), dim3(256), 0, h->stream, A);
      else if (K == 3) hipLaunchKernelGGL(lbm::k_fused<3>, dim3(nb), dim3(256), 0, h->stream, A);
      else hipLaunchKernelGGL(lbm::k_fused<4>, dim3(nb), dim3(256), 0, h->stream, A);
      TAU_LAUNCH_CHECK("lbm::k_fused");
    } else {
      hipLaunchKernelGGL(lbm::k_collide_stream, dim3((unsigned)(A.nbx * A.ny)), dim3(256), 0, h->stream, A);
      TAU_LAUNCH_CHECK("lbm::k_collide_stream");
    }
    h->cur ^= 1;   // std::swap(d_f0, d_f1), :264 — once per pass
    h->step += K;
    s += K;
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
