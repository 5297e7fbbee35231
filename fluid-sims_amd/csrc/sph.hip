// sph.hip — 2D weakly-compressible SPH sub-step for gfx950 (MI355X).
//
// Reference (tau_sph.cu): per sub-step k_clear_heads -> k_build_cells (linked lists through
// atomicExch: nondeterministic order, pointer chasing) -> k_density_pressure_cell -> k_forces_cell ->
// k_integrate; every neighbour visit is a dependent global load.
//
// Here the uniform grid is rebuilt by a DETERMINISTIC counting sort instead of linked lists (k_clear_heads / k_build_cells,
// :159-176; all hand-written, no sort library):
//   1. k_count     cell index of every particle (same float divide + floor + clamp as grid_x/grid_y, :141-157 — the
//                  integer index is bit-exact) and a slot inside its cell: one atomicAdd per (wave, cell) — the lanes of a
//                  wave that share a cell are found with ballots and ranked with a popcount of the lanes below
//   2. k_tile_sums / k_scan   exclusive prefix of the per-cell counts -> cellStart[M + 1] (two small launches, no
//                  inter-workgroup waiting); the counts are zeroed for the next sub-step on the way
//   3. k_scatter   particle id -> cellStart[cell] + slot: every cell's particles are now contiguous, in the order the
//                  atomics happened to run
//   4. k_rank_gather   restores the order a stable sort would give: a particle's place in its cell is the number of
//                  ids in that cell's segment below its own (n ~ 16 loads per particle, all lanes of a cell reading the
//                  same words); the records are gathered into that place (16-B + 8-B, cell order)
//   So cells ascend, ids ascend inside a cell — bit-identical lists run after run and identical to the stable radix sort
//   this replaces (rounds 1-2: hipcub::DeviceRadixSort, six rocPRIM dispatches per sub-step).
//   4. k_density   one lane per (sorted) particle; the three cells of a grid row are ONE contiguous
//                  record range, so a particle walks 3 ranges of ~48 contiguous 16-B records instead of
//                  9 linked lists; the 64 lanes of a wave sit in ~4 neighbouring cells and read the
//                  same ranges (L1/LDS-free broadcast of the same lines)
//   5. k_forces    same walk over two 16-B records per neighbour, + symplectic-Euler integrate fused in
// State arrays stay in the reference's layout and order (pos, vel, acc as float2 AoS, s, press).
// Only the summation order differs from a linked-list run (ascending id inside row-ordered cells),
// which is a rounding-level difference (compare at 1e-5, cell indices exactly).

#include "../../include/taueng.h"
#include "tau_common.h"
#include <cmath>
#include <new>
#include <random>
#include <vector>

namespace sph {

struct Args {
  int N, Gx, Gy, M;
  float cell, h, mass, rho0, c0, gammaEOS, viscAlpha, gx, gy, boxX, boxY, dt;
  float alpha;        // 10 / (7 pi h^2), evaluated as the reference does (fp64 product, rounded once)
  int useVisc, useGrav;
  float2 *pos, *vel, *acc;
  float *s, *press;
  int *cellOf;
  unsigned *keys, *ids, *keys_s, *ids_s;   // per particle: cell, slot in the cell; per sorted place: cell, particle id
  unsigned *tmpKey, *tmpId;                // per scattered place (cells contiguous, order inside a cell not yet fixed)
  unsigned *cellCount, *tileSum;           // particles per cell (M + 1 padded to whole scan tiles; zero between sub-steps), sums per scan tile
  int *cellStart;     // M + 1
  float4 *recA;       // sorted: x, y, vx, vy
  float2 *recB;       // sorted: p / rho^2, rho
  float2 *recP;       // sorted: x, y once more (the scan only needs 8 B per candidate)
  unsigned *nbrMask;  // [NW][N] in-range bitmasks, written by k_density, read by k_forces
  unsigned *ovfMask;  // [3][OVW][N] the same for the 32-candidate blocks beyond the ROWCAP candidates a row's masks cover (dense states)
  float4 *recA2;      // sorted: x, y, vx, vy AFTER the integrate (only when XSPH is on)
  int *rainWinner;    // per particle: highest drop index that picked it this launch, else -1 (only with rain)
  float xsphEps;
  int countNext;      // k_forces also counts its moved particles into the cells of the NEXT build (k_count's work, fused)
};

__device__ __forceinline__ int grid_c(float x, float cell, int G) { // grid_x / grid_y, :141-157
  int g = (int)floorf(x / cell);
  g = g < 0 ? 0 : g;
  return g >= G ? G - 1 : g;
}
__device__ __forceinline__ float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsqf(float x) { return __builtin_amdgcn_rsqf(x); }

// ---- cell build: counting sort --------------------------------------------------------------------
constexpr int SCAN_T = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_T * SCAN_ITEMS;   // cells per scan workgroup

// A particle's slot inside its cell for the next scatter: one atomicAdd per RUN of consecutive lanes in the same cell (lanes are
// consecutive particle ids in k_count, consecutive sorted places in k_forces: neighbouring lanes mostly share a cell): run heads
// from a compare with the lane below, a ballot of the heads, and bit arithmetic on it give every lane its run's head lane, its
// rank in the run and the run's length — no loop, whatever the particle order (scattered cells degrade to one atomic per lane,
// never to more work per lane).  Two runs of one cell in a wave are simply two atomics.  The valid lanes must be a prefix of
// the wave's active lanes; call it from converged code.
// STRIDE: the active lanes are every STRIDE-th lane (k_forces with four lanes per particle leaves lane 0 of each quad).
template <int STRIDE>
__device__ __forceinline__ unsigned cell_slot(const Args &A, unsigned c, bool valid) {
  const unsigned lane = __lane_id();
  const unsigned cprev = (unsigned)__shfl_up((int)c, STRIDE, 64);
  const bool head = valid && (lane < (unsigned)STRIDE || c != cprev);
  const unsigned long long heads = __ballot(head), vmask = __ballot(valid);
  const unsigned long long below = heads & (~0ull >> (63u - lane));            // heads at or below this lane
  const int hl = below ? 63 - __clzll((long long)below) : 0;
  const unsigned long long above = (lane == 63u) ? 0ull : (heads >> (lane + 1u)) << (lane + 1u);   // heads above this lane
  const int nexth = above ? (__ffsll((long long)above) - 1) : (63 - __clzll((long long)(vmask | 1ull)) + STRIDE);   // (the last run ends behind the last valid lane)
  unsigned base = 0;
  if (head) base = atomicAdd(&A.cellCount[c], (unsigned)((nexth - (int)lane) / STRIDE));
  base = (unsigned)__shfl((int)base, hl, 64);
  return base + (lane - (unsigned)hl) / (unsigned)STRIDE;
}

__global__ __launch_bounds__(256) void k_count(const Args A) { // k_build_cells' index arithmetic, :170-174, + the slot
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < A.N;
  unsigned c = 0xFFFFFFFFu;
  if (valid) {
    const float2 p = A.pos[i];
    c = (unsigned)(grid_c(p.y, A.cell, A.Gy) * A.Gx + grid_c(p.x, A.cell, A.Gx));
    A.keys[i] = c;
  }
  const unsigned slot = cell_slot<1>(A, c, valid);
  if (valid) A.ids[i] = slot;
}

// exclusive prefix of cellCount[0 .. M] (M + 1 entries: entry M is always 0, so cellStart[M] = N).  Every workgroup of k_scan
// first adds up what lies below its 4096-cell tile — the raw counts themselves while there are at most DIRECT_TILES tiles
// (one launch: 65 536 particles are two tiles), else the tile sums of a k_tile_sums launch — and then scans its tile.  No
// workgroup ever waits for another.  (k_rank_gather clears the counts for the next sub-step.)
constexpr int DIRECT_TILES = 16;
__device__ __forceinline__ unsigned block_sum_256(unsigned v, unsigned *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const unsigned t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}
__global__ __launch_bounds__(SCAN_T) void k_tile_sums(const Args A) {
  __shared__ unsigned sh[4];
  const uint4 *src = (const uint4 *)(A.cellCount + (size_t)blockIdx.x * SCAN_TILE);
  unsigned v = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS / 4; k++) { const uint4 q = src[k * SCAN_T + threadIdx.x]; v += q.x + q.y + q.z + q.w; }
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) A.tileSum[blockIdx.x] = v;
}
__global__ __launch_bounds__(SCAN_T) void k_scan(const Args A, int ntiles) {
  __shared__ unsigned sh[4];
  __shared__ unsigned swave[4];
  const int tid = threadIdx.x, tile = blockIdx.x;
  unsigned before = 0;                                   // particles in the tiles below this one
  if (ntiles > DIRECT_TILES) {
    for (int t = tid; t < tile; t += SCAN_T) before += A.tileSum[t];
    before = block_sum_256(before, sh);
  } else if (tile > 0) {
    const uint4 *lowc = (const uint4 *)A.cellCount;
    for (int t = tid; t < tile * (SCAN_TILE / 4); t += SCAN_T) { const uint4 q = lowc[t]; before += q.x + q.y + q.z + q.w; }
    before = block_sum_256(before, sh);
  }
  // 16 consecutive cells per thread
  const uint4 *cnt = (const uint4 *)(A.cellCount + (size_t)tile * SCAN_TILE) + tid * (SCAN_ITEMS / 4);
  unsigned v[SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS / 4; k++) {
    const uint4 q = cnt[k];
    v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
  }
  unsigned tsum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) tsum += v[k];
  // exclusive scan of the 256 thread sums: inclusive wave scan, then the wave totals
  unsigned inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned n = (unsigned)__shfl_up((int)inc, o, 64); if ((tid & 63) >= o) inc += n; }
  if ((tid & 63) == 63) swave[tid >> 6] = inc;
  __syncthreads();
  unsigned wbase = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) wbase += (w < (tid >> 6)) ? swave[w] : 0u;
  unsigned run = before + wbase + (inc - tsum);
  const size_t c0 = (size_t)tile * SCAN_TILE + (size_t)tid * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (c0 + k <= (size_t)A.M) A.cellStart[c0 + k] = (int)run;
    run += v[k];
  }
}

__global__ __launch_bounds__(256) void k_scatter(const Args A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.N) return;
  const unsigned c = A.keys[i];
  const unsigned p = (unsigned)A.cellStart[c] + A.ids[i];
  A.tmpKey[p] = c;
  A.tmpId[p] = (unsigned)i;
}

// place p of the scattered order -> place s + (number of ids of the cell below its own): ascending ids inside the cell,
// what k_build_cells' lists would hold if its atomics ran in id order (:165-176).  Also the gather of the records.
__global__ __launch_bounds__(256) void k_rank_gather(const Args A, int ncount) {
  __shared__ unsigned sId[256];
  const int p0 = blockIdx.x * 256, p = p0 + threadIdx.x;
  for (int q = p; q < ncount; q += gridDim.x * 256) A.cellCount[q] = 0u;   // the counts have been scanned: zero for the next build
  const bool valid = p < A.N;
  unsigned c = 0, id = 0;
  if (valid) { c = A.tmpKey[p]; id = A.tmpId[p]; }
  sId[threadIdx.x] = id;
  __syncthreads();
  if (!valid) return;
  // the cell's segment [s, e): the part inside this workgroup's 256 places comes from LDS (a cell is ~16 consecutive places, so
  // almost all of it), the rest from memory
  const int s = A.cellStart[c], e = A.cellStart[c + 1];
  const int a = max(s, p0), b = min(e, p0 + 256);
  int r = 0;
  for (int q = s; q < a; q++) r += (A.tmpId[q] < id) ? 1 : 0;
  for (int q = a; q < b; q++) r += (sId[q - p0] < id) ? 1 : 0;
  for (int q = max(b, s); q < e; q++) r += (A.tmpId[q] < id) ? 1 : 0;
  const int k = s + r;
  A.keys_s[k] = c;
  A.ids_s[k] = id;
  const float2 pp = A.pos[id], v = A.vel[id];
  A.recA[k] = make_float4(pp.x, pp.y, v.x, v.y);
  A.recP[k] = pp;
}

// cellOf[particle] of the last build, from its sorted arrays (tausph_download: the integer cell index gy * Gx + gx, :170-174)
__global__ __launch_bounds__(256) void k_cell_of(const Args A) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < A.N) A.cellOf[A.ids_s[k]] = (int)A.keys_s[k];
}

// before any build: the cell index of the current positions (same arithmetic as k_count)
__global__ __launch_bounds__(256) void k_cell_of_pos(const Args A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.N) return;
  const float2 p = A.pos[i];
  A.cellOf[i] = grid_c(p.y, A.cell, A.Gy) * A.Gx + grid_c(p.x, A.cell, A.Gx);
}

// ordered pairs (i, j), i != j, closer than 2h among the records of the last build — what the density and force passes of
// that sub-step evaluated (diagnostic: bench.py's pair-interactions/s; not part of a step)
__global__ __launch_bounds__(256) void k_count_pairs(const Args A, unsigned long long *out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  if (k < A.N) {
    const float2 me = A.recP[k];
    const int c = (int)A.keys_s[k];
    const int gy = c / A.Gx, gx = c - gy * A.Gx;
    const int cxlo = max(gx - 1, 0), cxhi = min(gx + 1, A.Gx - 1);
    const float twoh = 2.f * A.h, twoh2 = twoh * twoh;
    for (int oy = -1; oy <= 1; ++oy) {
      const int cy = gy + oy;
      if ((unsigned)cy >= (unsigned)A.Gy) continue;
      const int j0 = A.cellStart[cy * A.Gx + cxlo], j1 = A.cellStart[cy * A.Gx + cxhi + 1];
      for (int j = j0; j < j1; j++) {
        const float2 o = A.recP[j];
        const float dx = me.x - o.x, dy = me.y - o.y;
        n += (j != k && dx * dx + dy * dy < twoh2) ? 1ull : 0ull;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

__device__ __forceinline__ float W_cubic(float r, float ih, float alpha) { // :105-116
  float q = r * ih;
  float q2 = q * q;
  float t = 2.f - q;
  const float w1 = 1.f - 1.5f * q2 + 0.75f * q2 * q;
  const float w2 = 0.25f * t * t * t;
  const float w12 = (q < 1.0f) ? w1 : w2;      // both pieces are a few multiplies: selects, no exec-mask branches in the pair loop
  return alpha * ((q < 2.0f) ? w12 : 0.f);
}

// Pair evaluations of the two neighbour passes.  The constant factors are taken OUT of the sums — rho = (m alpha) sum s(q),
// a = (-m alpha / h) sum (p_i/rho_i^2 + p_j/rho_j^2 + Pi_ij) s'(q) / r (x_i - x_j) — and the constants a pair still needs sit
// in VGPRs: a VALU instruction with an SGPR operand issues at half rate on gfx950 (profiles/r02/valu_calib.txt), and the
// reference's order (m * (alpha * s) per pair, :105-116, 196-201) spent four such multiplies per density pair.  A reassociation
// at rounding level (the sums already run in another order than the reference's lists).
__device__ __forceinline__ float vreg(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
// W_cubic / alpha for a pair the range test admitted (r2 < (2h)^2: q < 2 up to rounding, where the q >= 2 branch of :113-115
// differs from the cubic by less than 1e-20)
struct DensK { float ih, c075; };
__device__ __forceinline__ DensK dens_k(const Args &A) { return DensK{vreg(1.0f / A.h), vreg(0.75f)}; }
__device__ __forceinline__ float W_shape(const DensK &K, float q) {
  const float q2 = q * q, t = 2.f - q;
  const float w1 = fmaf(q2, fmaf(K.c075, q, -1.5f), 1.f);   // 1 - 1.5 q^2 + 0.75 q^3 (one literal per instruction: 0.75 in a VGPR)
  const float w2 = (0.25f * t) * (t * t);
  return (q < 1.0f) ? w1 : w2;
}
__device__ __forceinline__ void dens_pair(const DensK &K, float2 me, float2 o, bool on, float &rho) {
  const float dx = me.x - o.x, dy = me.y - o.y;
  const float w = W_shape(K, __builtin_amdgcn_sqrtf(dx * dx + dy * dy) * K.ih);
  rho += on ? w : 0.f;
}
__device__ __forceinline__ float dens_scale(const Args &A) { return A.mass * A.alpha; }

struct ForceK { float ih, eps2, cv; };
__device__ __forceinline__ ForceK force_k(const Args &A) {
  return ForceK{vreg(1.0f / A.h), vreg(0.01f * A.h * A.h), vreg(-2.f * A.viscAlpha * A.c0 * A.h)};
}
// One neighbour inside the support.  Straight-line on purpose: the reference's skips (self and coincident particles :231-233 —
// both have r2 = 0 —, gradW's r guard :119, receding pairs :254) become selects that contribute exact zeros: a few VALU ops, no
// exec-mask juggling inside the hottest loop of the pass.  Two transcendentals per pair instead of four: 1/r from rsq with
// r = r2 / r, and the two divisions of the viscosity term (:256-258) as one reciprocal of the product.
template <bool VISC>
__device__ __forceinline__ void force_pair(const ForceK &K, float4 me, float2 meB, float4 o, float2 oB, bool on, float &ax, float &ay) {
  const float dx = me.x - o.x, dy = me.y - o.y;
  const float r2 = dx * dx + dy * dy;
  const float ir = rsqf(r2), r = r2 * ir;
  const bool valid = on & (r2 > 1e-16f) & (r > 1e-8f);
  const float q = r * K.ih, t = 2.0f - q;                // gradW_cubic / alpha, :118-133
  const float dWa = q * (2.25f * q - 3.0f), dWb = -0.75f * t * t;
  const float s0 = ((q < 1.0f) ? dWa : dWb) * ir;
  const float sg = valid ? s0 : 0.f;                     // (a select of two VALUES: `valid ? expr : 0` becomes an exec-mask branch)
  float pv = meB.x + oB.x;                               // p_i/rho_i^2 + p_j/rho_j^2
  if (VISC) {   // (A.useVisc, resolved outside the loop: a uniform branch between the two evaluations of a trip would order them)
    const float dvx = me.z - o.z, dvy = me.w - o.w;
    const float dot = fminf(dvx * dx + dvy * dy, 0.f);   // receding pairs: mu = 0, Pi = 0
    // mu = h dot / (r2 + eps2), Pi = -alpha_v c0 mu / (0.5 (rho_i + rho_j))
    pv = fmaf(K.cv * dot, rcpf((r2 + K.eps2) * (meB.y + oB.y)), pv);
  }
  const float gp = sg * pv;
  ax = fmaf(gp, dx, ax);
  ay = fmaf(gp, dy, ay);
}
__device__ __forceinline__ float force_scale(const Args &A) { return -A.mass * A.alpha * (1.0f / A.h); }

// ---- neighbour passes ------------------------------------------------------------------------------
// A particle's candidates are the 3 x 3 cells around it = three contiguous record ranges (~96 records
// each at the dam's packing), of which only the ones inside the 2h disc (~35 %) contribute.  Walking the
// ranges and branching on the distance does not skip anything on a 64-wide wave (some lane is always in
// range), so every candidate costs the full kernel + gradient evaluation.  Instead:
//   stage  a workgroup = 256 consecutive sorted particles = a run of cells of one grid row, so the union
//          of their candidate ranges is again three contiguous ranges (the run widened by one cell each
//          side): copied into LDS once with coalesced loads.  The vector-memory address unit is the
//          measured limiter otherwise (TA busy 87 % with every neighbour visit a global gather).
//   scan   (density pass only) one cheap loop over the candidates' positions builds a per-particle
//          bitmask of the in-range ones: WPR words per row range, kept in LDS and written to nbrMask
//   eval   both passes iterate the SET bits only (ctz / clear-lowest), so the expensive part runs
//          ~max-over-lanes(hits) times instead of ~max-over-lanes(candidates); the forces pass reads the
//          masks back (positions do not move between the two passes) and never scans
// Candidates beyond ROWCAP per row (pathological packing) are evaluated directly, and a workgroup whose
// particles straddle two grid rows or whose union overflows CAP walks global memory: any state is handled.
constexpr int WPR = 4;              // mask words per row range
constexpr int ROWCAP = 32 * WPR;    // candidates per row range covered by the mask
constexpr int NW = 3 * WPR;
constexpr int OVW = 60;             // overflow blocks per row range whose hit masks the density pass hands to the force pass:
                                    // (WPR + OVW) * 32 = 2048 candidates per row (the collapsed dam: ~1200); beyond that both scan
constexpr int CAP = 3 * (256 + 128); // staged records: 3 rows x (the workgroup's run + one cell either side)

// per lanes-per-particle configuration: particles per workgroup and the staged records (LPP = 2: half the run, so half the
// union: 24 KB instead of 40 per workgroup in the force pass — six workgroups per CU instead of four)
template <int LPP> struct Cfg {
  static constexpr int PPW = 256 / LPP;
  static constexpr int CAPL = (LPP == 2) ? 3 * (128 + 128) : CAP;
};
struct Walk {                        // one lane's three candidate ranges
  int jb[3], jn[3];
};
__device__ __forceinline__ Walk make_walk(const Args &A, int k) {
  const int c = (int)A.keys_s[k];
  const int gy = c / A.Gx, gx = c - gy * A.Gx;
  const int cxlo = max(gx - 1, 0), cxhi = min(gx + 1, A.Gx - 1);
  Walk wk;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int cy = gy + r - 1;
    const bool ok = (unsigned)cy < (unsigned)A.Gy;
    const int j0 = ok ? A.cellStart[cy * A.Gx + cxlo] : 0;
    const int j1 = ok ? A.cellStart[cy * A.Gx + cxhi + 1] : 0;
    wk.jb[r] = j0; wk.jn[r] = j1 - j0;
  }
  return wk;
}

// The three record ranges a workgroup stages, and the shift from sorted index to LDS slot per row.
struct Stage {
  int base[3], len[3], delta[3];
  int spread[3];   // how far the last particle's range starts behind the first one's (records), per row
  bool on;      // the three ranges fit the stage
  bool row;     // the workgroup's particles lie in one grid row (the ranges are defined at all)
};
__device__ __forceinline__ Stage make_stage(const Args &A, int k0, int ppw, int cap) {
  Stage st;
  const int kl = min(k0 + ppw - 1, A.N - 1);
  const int c0 = (int)A.keys_s[k0], c1 = (int)A.keys_s[kl];
  const int gy = c0 / A.Gx;
  const int cxa = max(c0 - gy * A.Gx - 1, 0), cxb = min(c1 - gy * A.Gx + 1, A.Gx - 1);
  int total = 0;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int cy = gy + r - 1;
    const bool ok = (unsigned)cy < (unsigned)A.Gy && cxb >= cxa;
    st.base[r] = ok ? A.cellStart[cy * A.Gx + cxa] : 0;
    st.len[r] = ok ? A.cellStart[cy * A.Gx + min(cxb, A.Gx - 1) + 1] - st.base[r] : 0;
    st.spread[r] = ok ? A.cellStart[cy * A.Gx + max(c1 - gy * A.Gx - 1, 0)] - st.base[r] : 0;
    st.delta[r] = total - st.base[r];
    total += st.len[r];
  }
  st.row = c1 / A.Gx == gy;
  st.on = st.row && total <= cap;
  return st;
}

// Lanes per particle.  LPP = 1: one lane walks all NW mask words of its particle.  LPP = 4 (small N: 65 536
// particles are one wave per SIMD otherwise, and a lone wave issues its dependent chain at a fraction of the VALU
// rate): four consecutive lanes share a particle, lane `sub` owns word `sub` of each of the three row ranges
// (words sub, sub + 4, sub + 8) and every fourth 32-candidate block of the overflow; the partial sums meet in a
// quad reduction.  The masks in LDS and in nbrMask are indexed by particle either way.
//
// iterate the set bits of this lane's mask words (column `pl` of sM) in ascending candidate order
#if !defined(TAU_EXPERIMENT) && (defined(TAUSPH_NH_D) || defined(TAUSPH_NH_F))
#error "tuning overrides need -DTAU_EXPERIMENT (scripts/variant_build_file.sh sets it)"
#endif
#ifndef TAUSPH_NH_D
#define TAUSPH_NH_D 2   // pair evaluations per trip, density pass
#endif
#ifndef TAUSPH_NH_F
#define TAUSPH_NH_F 2   // forces pass
#endif
template <int LPP, int NH, int PW, class F>
__device__ __forceinline__ void for_each_hit(const unsigned (*sM)[PW], int pl, int sub, const Walk &wk, F &&body) {
  int w = sub;
  unsigned m = sM[w][pl];
  const int b0 = wk.jb[0], d1 = wk.jb[1] - wk.jb[0], d2 = wk.jb[2] - wk.jb[1];
  int wbase = b0 + ((w & (WPR - 1)) << 5);
  // One straight-line step per trip: a lane whose word ran dry fetches its next word (an empty word costs
  // that lane one idle trip), then every lane holding a bit evaluates it.  No inner loop: the other lanes
  // of the wave would only wait for it.
  while (m != 0u || w + LPP < NW) {
    if (m == 0u) {
      w += LPP;
      m = sM[w][pl];
      // (two independent selects: a chained select over jb[] is turned into a 3-entry table in scratch memory,
      // and the load sits on the critical path of every word fetch)
      wbase = b0 + (w >= WPR ? d1 : 0) + (w >= 2 * WPR ? d2 : 0) + ((w & (WPR - 1)) << 5);
    }
    if (m != 0u) {
      // NH hits per trip where the word holds as many: the pair evaluations are independent until their sums, so the second
      // one's LDS read and its rsq / rcp run in the shadow of the first's (a pass is a chain of ~100 dependent trips per wave
      // at four to seven waves per SIMD: latency, not issue, is what a trip costs).  `on` = false: the body adds exact zeros.
      int j[NH];
      bool on[NH];
#pragma unroll
      for (int e = 0; e < NH; e++) {
        on[e] = m != 0u;
        j[e] = on[e] ? wbase + __builtin_ctz(on[e] ? m : 1u) : j[0];
        m &= m - 1u;                      // (0 stays 0)
      }
#pragma unroll
      for (int e = 0; e < NH; e++) body(j[e], on[e]);
    }
  }
}
// the candidates of over-full rows that the mask does not cover (32-candidate blocks dealt round-robin to the lanes).
// A block is SCANNED first (32 cheap distance tests -> one mask word in a register) and then only its set bits are
// evaluated: walking every candidate with an `if (in range)` costs the full evaluation per candidate on a 64-wide wave
// (some lane is always in range), which is what made the collapsed dam (~1200 candidates per row, ~35 % in range) 2.8x
// more expensive than its pair count.  `inrange(j)` is the cheap test, `body(j)` the evaluation of a hit.
// MODE 0: scan every block (no mask storage); 1: scan and STORE the block masks (density pass); 2: LOAD them instead of scanning
// (force pass: positions have not moved since the density pass).  Round 4: on the collapsed dam the force pass spent two thirds
// of its time re-scanning ~3 600 candidates per particle for the ~350 it then evaluates.
template <int LPP, int MODE, class T, class F>
__device__ __forceinline__ void for_each_overflow(const Args &A, int k, const Walk &wk, int sub, T &&inrange, F &&body) {
#pragma unroll
  for (int r = 0; r < 3; r++)
    for (int blk = WPR + sub; blk * 32 < wk.jn[r]; blk += LPP) {
      const int j0 = wk.jb[r] + blk * 32, cnt = min(32, wk.jn[r] - blk * 32);
      const int ow = blk - WPR;
      const bool kept = MODE != 0 && ow < OVW;
      unsigned m = 0u;
      if (MODE == 2 && kept) {
        m = A.ovfMask[((size_t)(r * OVW + ow)) * A.N + k];
      } else {
#pragma unroll 4
        for (int b = 0; b < cnt; b++) m = m + m + (inrange(j0 + b) ? 1u : 0u);
        m = __builtin_bitreverse32(m) >> (32 - cnt);                 // candidate b -> bit b (cnt >= 1 here)
        if (MODE == 1 && kept) A.ovfMask[((size_t)(r * OVW + ow)) * A.N + k] = m;
      }
      while (m != 0u) {                                            // two hits per trip, as in for_each_hit
        const int ja = j0 + __builtin_ctz(m);
        m &= m - 1u;
        const bool two = m != 0u;
        const int jb = two ? j0 + __builtin_ctz(two ? m : 1u) : ja;
        m &= m - 1u;
        body(ja, true);
        body(jb, two);
      }
    }
}
template <int LPP> __device__ __forceinline__ float quad_sum(float v) {
  if (LPP >= 2) v += __shfl_xor(v, 1, 64);
  if (LPP == 4) v += __shfl_xor(v, 2, 64);
  return v;
}

// density of one particle; P indexes candidate positions (LDS slots or sorted records — wk is in the same space)
template <int LPP, int PW, class PosArr>
__device__ __forceinline__ float density_of(const Args &A, unsigned (*sM)[PW], int pl, int sub, int k, const Walk &wk,
                                            float2 me, PosArr P) {
  const float twoh = 2.f * A.h, twoh2 = twoh * twoh;
  // scan: bit b of word (r, w) <=> candidate jb[r] + 32 w + b is inside the support
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll 1
    for (int w = sub; w < WPR; w += LPP) {
      const int base = wk.jb[r] + 32 * w;
      const int cnt = min(max(wk.jn[r] - 32 * w, 0), 32);
      unsigned m = 0u;                     // built MSB-last: m = 2 m + hit is one add-with-carry per candidate
#pragma unroll 4
      for (int b = 0; b < cnt; b++) {
        const float2 o = P[base + b];
        const float dx = me.x - o.x, dy = me.y - o.y;
        const float r2 = dx * dx + dy * dy;
        m = m + m + ((r2 < twoh2) ? 1u : 0u);
      }
      m = cnt ? (__builtin_bitreverse32(m) >> (32 - cnt)) : 0u;   // candidate b -> bit b
      sM[r * WPR + w][pl] = m;
      A.nbrMask[(size_t)(r * WPR + w) * A.N + k] = m;
    }
  }
  float rho = 0.f;
  const DensK K = dens_k(A);
  auto add = [&](int j, bool on) { dens_pair(K, me, P[j], on, rho); };
  for_each_hit<LPP, TAUSPH_NH_D, PW>(sM, pl, sub, wk, add);
  if (A.ovfMask) for_each_overflow<LPP, 1>(A, k, wk, sub, [&](int j) {
    const float2 o = P[j];
    const float dx = me.x - o.x, dy = me.y - o.y;
    return dx * dx + dy * dy < twoh2;
  }, add);
  else for_each_overflow<LPP, 0>(A, k, wk, sub, [&](int j) {
    const float2 o = P[j];
    const float dx = me.x - o.x, dy = me.y - o.y;
    return dx * dx + dy * dy < twoh2;
  }, add);
  return dens_scale(A) * quad_sum<LPP>(rho);
}

// entropy variable, pressure and the force pass's record of sorted place k from its summed density (:204-212)
__device__ __forceinline__ void density_store(const Args &A, int k, float rho) {
  const float si = logf(fmaxf(rho, 1e-6f));
  rho = expf(si);
  const float ratio = rho / A.rho0;
  float p = (A.c0 * A.c0) * A.rho0 * (powf(ratio, A.gammaEOS) - 1.0f) / A.gammaEOS;
  p = fmaxf(p, 0.0f);
  const unsigned id = A.ids_s[k];
  A.s[id] = si;
  A.press[id] = p;
  A.recB[k] = make_float2(p / (rho * rho), rho);
}

// ---- dense states: a workgroup whose candidate ranges do not fit the stage walks them in ROUNDS through it ------------------
// (the collapsed dam: 100-400 particles per cell, 1 500-3 600 candidates per particle against the stage's 1 152 — round 3
// walked those from global memory: every distance test a broadcast load, every hit an L2 gather.)  Round t of a row takes the
// 32-candidate blocks [t TB, (t + 1) TB) of EVERY lane's own range — the lanes of a workgroup have ranges of similar length, so
// they all have work in every round but the last — and stages the records those blocks can touch: from the first lane's block
// t TB to the last lane's block (t + 1) TB, i.e. 32 TB + `spread` records.  A lane scans its blocks into sM (density pass: and
// into nbrMask / ovfMask) or fetches their masks (force pass), then iterates the set bits on its own, two hits per trip, all
// from LDS.  One lane per particle only (LPP = 1: the large-N configuration).
// (Tiles of consecutive records with each lane taking the blocks that START in the tile were measured first: lanes whose range
// begins mid-tile idle, 51 % of the lanes active per VALU instruction, no faster than the global walk.)
constexpr int TBLK = NW;   // blocks per lane and round: the mask words sM holds
__device__ __forceinline__ bool rounds_ok(const Stage &st, int cap) {   // every row gets at least four blocks per round
  return st.row && !st.on && max(max(st.spread[0], st.spread[1]), st.spread[2]) + 4 * 32 <= cap;
}
template <class F>
__device__ __forceinline__ void tile_hits(const unsigned (*sM)[256], int pl, int nbl, int s0, F &&body) {
  if (nbl <= 0) return;
  int w = 0, wbase = s0;
  unsigned m = sM[0][pl];
  while (m != 0u || w + 1 < nbl) {
    if (m == 0u) { w++; m = sM[w][pl]; wbase += 32; }
    if (m != 0u) {
      const int ja = wbase + __builtin_ctz(m);
      m &= m - 1u;
      const bool two = m != 0u;
      const int jb = two ? wbase + __builtin_ctz(two ? m : 1u) : ja;
      m &= m - 1u;
      body(ja, true);
      body(jb, two);
    }
  }
}
template <class PosOf>
__device__ __forceinline__ unsigned tile_scan(int s, int cnt, float2 me, float twoh2, PosOf &&pos) {
  unsigned m = 0u;
#pragma unroll 4
  for (int b = 0; b < cnt; b++) {
    const float2 o = pos(s + b);
    const float dx = me.x - o.x, dy = me.y - o.y;
    m = m + m + ((dx * dx + dy * dy < twoh2) ? 1u : 0u);
  }
  return __builtin_bitreverse32(m) >> (32 - cnt);   // candidate b -> bit b (cnt >= 1)
}
__device__ __forceinline__ float density_tiled(const Args &A, const Stage &st, unsigned (*sM)[256], float2 *sP, int cap, int tid,
                                               int k, bool active, const Walk &wk, float2 me) {
  const float twoh = 2.f * A.h, twoh2 = twoh * twoh;
  float rho = 0.f;
  const DensK K = dens_k(A);
  auto add = [&](int j, bool on) { dens_pair(K, me, sP[j], on, rho); };
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int lo = st.base[r], u1 = lo + st.len[r];
    const int tb = min(TBLK, (cap - st.spread[r]) >> 5);
    const int off = wk.jb[r] - lo, jn = active ? wk.jn[r] : 0, nb = (jn + 31) >> 5;
    for (int t = 0; __syncthreads_or(nb > t * tb); t++) {   // (the barrier also frees the stage of the round before)
      const int w0 = lo + 32 * tb * t, n = min(u1, w0 + st.spread[r] + 32 * tb) - w0;
      for (int i = tid; i < n; i += 256) sP[i] = A.recP[w0 + i];
      __syncthreads();
      const int nbl = min(nb - t * tb, tb);
      for (int w = 0; w < nbl; w++) {
        const int sl = off + 32 * w, blk = t * tb + w;
        const unsigned m = tile_scan(sl, min(32, jn - 32 * blk), me, twoh2, [&](int j) { return sP[j]; });
        sM[w][tid] = m;
        if (blk < WPR) A.nbrMask[(size_t)(r * WPR + blk) * A.N + k] = m;
        else if (A.ovfMask && blk - WPR < OVW) A.ovfMask[((size_t)(r * OVW + blk - WPR)) * A.N + k] = m;
      }
      tile_hits(sM, tid, nbl, off, add);
    }
  }
  return dens_scale(A) * rho;
}

template <int LPP>
__global__ __launch_bounds__(256) void k_density(const Args A) { // k_density_pressure_cell, :178-213
  constexpr int PPW = Cfg<LPP>::PPW;      // particles per workgroup
  __shared__ unsigned sM[NW][PPW];
  __shared__ float2 sP[Cfg<LPP>::CAPL];
  const int tid = threadIdx.x, pl = tid / LPP, sub = tid % LPP, k0 = blockIdx.x * PPW, k = k0 + pl;
  const Stage st = make_stage(A, k0, PPW, Cfg<LPP>::CAPL);
  if (st.on) {
#pragma unroll
    for (int r = 0; r < 3; r++)
      for (int i = tid; i < st.len[r]; i += 256) sP[st.base[r] + st.delta[r] + i] = A.recP[st.base[r] + i];
    __syncthreads();
  }
  if constexpr (LPP == 1) if (rounds_ok(st, Cfg<LPP>::CAPL)) {   // dense state (workgroup-uniform branch)
    const bool active = k < A.N;
    const int kc = active ? k : A.N - 1;
    const float2 me = A.recP[kc];
    const Walk wk = make_walk(A, kc);
    const float rho = density_tiled(A, st, sM, sP, Cfg<LPP>::CAPL, tid, kc, active, wk, me);
    if (active) density_store(A, k, rho);
    return;
  }
  if (k >= A.N) return;                // whole quads leave together (k is the same for the LPP lanes)
  const float2 me = A.recP[k];
  Walk wk = make_walk(A, k);
  float rho;
  if (st.on) {
#pragma unroll
    for (int r = 0; r < 3; r++) wk.jb[r] += st.delta[r];
    rho = density_of<LPP, PPW>(A, sM, pl, sub, k, wk, me, (const float2 *)sP);
  } else {
    rho = density_of<LPP, PPW>(A, sM, pl, sub, k, wk, me, (const float2 *)A.recP);
  }
  if (sub != 0) return;
  density_store(A, k, rho);
}

// acceleration of one particle; RA / RB index the candidates' records in wk's index space, kg = own sorted place
template <int LPP, bool VISC, int PW, class ArrA, class ArrB>
__device__ __forceinline__ float2 accel_of(const Args &A, unsigned (*sM)[PW], int pl, int sub, int kg, const Walk &wk,
                                           float4 me, float2 meB, ArrA RA, ArrB RB) {
  const float twoh = 2.f * A.h, twoh2 = twoh * twoh;
  float ax = 0.f, ay = 0.f;
  const ForceK K = force_k(A);
  auto add = [&](int j, bool on) { force_pair<VISC>(K, me, meB, RA[j], RB[j], on, ax, ay); };
  for_each_hit<LPP, TAUSPH_NH_F, PW>(sM, pl, sub, wk, add);
  if (A.ovfMask) for_each_overflow<LPP, 2>(A, kg, wk, sub, [&](int j) {
    const float4 o = RA[j];
    const float dx = me.x - o.x, dy = me.y - o.y;
    return dx * dx + dy * dy < twoh2;
  }, add);
  else for_each_overflow<LPP, 0>(A, kg, wk, sub, [&](int j) {
    const float4 o = RA[j];
    const float dx = me.x - o.x, dy = me.y - o.y;
    return dx * dx + dy * dy < twoh2;
  }, add);
  const float sc = force_scale(A);
  return make_float2(sc * quad_sum<LPP>(ax), sc * quad_sum<LPP>(ay));
}

// the force pass in rounds (see density_tiled): the masks come from the density pass — nbrMask, ovfMask — and only blocks
// beyond what those hold (more than (WPR + OVW) * 32 candidates in a row) are scanned again
template <bool VISC>
__device__ __forceinline__ float2 accel_tiled(const Args &A, const Stage &st, unsigned (*sM)[256], float4 *sA, float2 *sB, int cap,
                                              int tid, int k, bool active, const Walk &wk, float4 me, float2 meB) {
  const float twoh = 2.f * A.h, twoh2 = twoh * twoh;
  float ax = 0.f, ay = 0.f;
  const ForceK K = force_k(A);
  auto add = [&](int j, bool on) { force_pair<VISC>(K, me, meB, sA[j], sB[j], on, ax, ay); };
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int lo = st.base[r], u1 = lo + st.len[r];
    const int tb = min(TBLK, (cap - st.spread[r]) >> 5);
    const int off = wk.jb[r] - lo, jn = active ? wk.jn[r] : 0, nb = (jn + 31) >> 5;
    for (int t = 0; __syncthreads_or(nb > t * tb); t++) {
      const int w0 = lo + 32 * tb * t, n = min(u1, w0 + st.spread[r] + 32 * tb) - w0;
      for (int i = tid; i < n; i += 256) { sA[i] = A.recA[w0 + i]; sB[i] = A.recB[w0 + i]; }
      const int nbl = min(nb - t * tb, tb);
      bool rescan = false;
      for (int w = 0; w < nbl; w++) {                    // (global loads: in flight across the barrier below)
        const int blk = t * tb + w;
        unsigned m = 0u;
        if (blk < WPR) m = A.nbrMask[(size_t)(r * WPR + blk) * A.N + k];
        else if (A.ovfMask && blk - WPR < OVW) m = A.ovfMask[((size_t)(r * OVW + blk - WPR)) * A.N + k];
        else rescan = true;
        sM[w][tid] = m;
      }
      __syncthreads();
      if (rescan)
        for (int w = 0; w < nbl; w++) {
          const int blk = t * tb + w;
          if (blk >= WPR && !(A.ovfMask && blk - WPR < OVW))
            sM[w][tid] = tile_scan(off + 32 * w, min(32, jn - 32 * blk), make_float2(me.x, me.y), twoh2,
                                   [&](int j) { const float4 o = sA[j]; return make_float2(o.x, o.y); });
        }
      tile_hits(sM, tid, nbl, off, add);
    }
  }
  const float sc = force_scale(A);
  return make_float2(sc * ax, sc * ay);
}

template <int LPP>
__global__ __launch_bounds__(256) void k_forces(const Args A) { // k_forces_cell :215-272 + k_integrate :324-355
  constexpr int PPW = Cfg<LPP>::PPW;
  __shared__ unsigned sM[NW][PPW];
  __shared__ float4 sA[Cfg<LPP>::CAPL];
  __shared__ float2 sB[Cfg<LPP>::CAPL];
  const int tid = threadIdx.x, pl = tid / LPP, sub = tid % LPP, k0 = blockIdx.x * PPW, k = k0 + pl;
  const Stage st = make_stage(A, k0, PPW, Cfg<LPP>::CAPL);
  if (st.on) {
#pragma unroll
    for (int r = 0; r < 3; r++)
      for (int i = tid; i < st.len[r]; i += 256) {
        sA[st.base[r] + st.delta[r] + i] = A.recA[st.base[r] + i];
        sB[st.base[r] + st.delta[r] + i] = A.recB[st.base[r] + i];
      }
    __syncthreads();
  }
  float4 me;
  float2 a;
  bool tiled = false;
  if constexpr (LPP == 1) if (rounds_ok(st, Cfg<LPP>::CAPL)) {   // dense state (workgroup-uniform branch)
    const bool active = k < A.N;
    const int kc = active ? k : A.N - 1;
    me = A.recA[kc];
    const float2 meB = A.recB[kc];
    const Walk wk = make_walk(A, kc);
    a = A.useVisc ? accel_tiled<true>(A, st, sM, sA, sB, Cfg<LPP>::CAPL, tid, kc, active, wk, me, meB)
                  : accel_tiled<false>(A, st, sM, sA, sB, Cfg<LPP>::CAPL, tid, kc, active, wk, me, meB);
    if (!active) return;
    tiled = true;
  }
  if (!tiled) {
    if (k >= A.N) return;
#pragma unroll
    for (int w = sub; w < NW; w += LPP) sM[w][pl] = A.nbrMask[(size_t)w * A.N + k];   // only the words this lane walks
    me = A.recA[k];
    const float2 meB = A.recB[k];
    Walk wk = make_walk(A, k);
    if (st.on) {
#pragma unroll
      for (int r = 0; r < 3; r++) wk.jb[r] += st.delta[r];
      a = A.useVisc ? accel_of<LPP, true, PPW>(A, sM, pl, sub, k, wk, me, meB, (const float4 *)sA, (const float2 *)sB)
                    : accel_of<LPP, false, PPW>(A, sM, pl, sub, k, wk, me, meB, (const float4 *)sA, (const float2 *)sB);
    } else {
      a = A.useVisc ? accel_of<LPP, true, PPW>(A, sM, pl, sub, k, wk, me, meB, (const float4 *)A.recA, (const float2 *)A.recB)
                    : accel_of<LPP, false, PPW>(A, sM, pl, sub, k, wk, me, meB, (const float4 *)A.recA, (const float2 *)A.recB);
    }
  }
  if (sub != 0) return;
  float ax = a.x, ay = a.y;
  if (A.useGrav) { ax += A.gx; ay += A.gy; }
  const unsigned id = A.ids_s[k];
  A.acc[id] = make_float2(ax, ay);
  // k_integrate
  float vx = me.z + ax * A.dt, vy = me.w + ay * A.dt;
  float x = me.x + vx * A.dt, y = me.y + vy * A.dt;
  const float e = 0.2f;
  if (x < 0.f) { x = 0.f; vx = -e * vx; }
  if (x > A.boxX) { x = A.boxX; vx = -e * vx; }
  if (y < 0.f) { y = 0.f; vy = -e * vy; }
  if (y > A.boxY) { y = A.boxY; vy = -e * vy; }
  A.pos[id] = make_float2(x, y);
  A.vel[id] = make_float2(vx, vy);
  if (A.recA2) A.recA2[k] = make_float4(x, y, vx, vy);
  if (A.countNext) {   // the moved particle's cell and slot for the NEXT sub-step's build: k_count without its launch and its read of pos
    const unsigned c = (unsigned)(grid_c(y, A.cell, A.Gy) * A.Gx + grid_c(x, A.cell, A.Gx));
    A.keys[id] = c;
    // the wave's remaining lanes are consecutive sorted places (every LPP-th lane, a prefix of them): runs of one cell
    A.ids[id] = cell_slot<LPP>(A, c, true);
  }
}

// XSPH velocity smoothing, k_xsph_cell + k_apply_xsph (:274-322; launched after the integrate, :698-704).
// The reference walks the cell lists built BEFORE the integrate with the positions and velocities from AFTER
// it; the sorted arrays are those lists, recA2 the moved records.  Off by default: a plain range walk.
__global__ __launch_bounds__(256) void k_xsph(const Args A) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= A.N) return;
  const float4 me = A.recA2[k];
  const float rhoi = A.recB[k].y;
  const int gx = grid_c(me.x, A.cell, A.Gx), gy = grid_c(me.y, A.cell, A.Gy);   // the cell it moved to
  const int cxlo = max(gx - 1, 0), cxhi = min(gx + 1, A.Gx - 1);
  const float twoh = 2.f * A.h, twoh2 = twoh * twoh, ih = 1.0f / A.h;
  float dvx = 0.f, dvy = 0.f;
  for (int oy = -1; oy <= 1; ++oy) {
    const int cy = gy + oy;
    if ((unsigned)cy >= (unsigned)A.Gy) continue;
    const int j0 = A.cellStart[cy * A.Gx + cxlo], j1 = A.cellStart[cy * A.Gx + cxhi + 1];
    for (int j = j0; j < j1; j++) {
      const float4 o = A.recA2[j];
      const float dx = me.x - o.x, dy = me.y - o.y;
      const float r2 = dx * dx + dy * dy;
      const float w = (j != k && r2 < twoh2) ? W_cubic(__builtin_amdgcn_sqrtf(r2), ih, A.alpha) : 0.f;
      const float c = (A.mass / (0.5f * (rhoi + A.recB[j].y))) * w;
      dvx += c * (o.z - me.z);
      dvy += c * (o.w - me.w);
    }
  }
  const unsigned id = A.ids_s[k];
  const float2 d = make_float2(A.xsphEps * dvx, A.xsphEps * dvy);
  A.acc[id] = d;                                           // the reference parks dvel in acc, :699
  A.vel[id] = make_float2(me.z + d.x, me.w + d.y);
}

// Rain, k_rain (:377-392).  Two drops may pick the same particle; the reference leaves the winner to the
// hardware.  Here the HIGHEST drop index wins (what an in-order launch would give): claim, then apply.
__device__ __forceinline__ unsigned rain_draw(unsigned seed, int k, float boxX, float boxY, float &x, float &y) {
#pragma clang fp contract(off)   // drop positions are compared bit for bit with the host restatement
  unsigned s = seed ^ ((unsigned)k * 1664525u + 1013904223u);
  s = s * 1664525u + 1013904223u;
  const float rx = (s & 0x00FFFFFF) / 16777216.f;
  s = s * 1664525u + 1013904223u;
  x = rx * (boxX * 0.8f) + 0.1f * boxX;
  const float ry = (s & 0x00FFFFFF) / 16777216.f;
  y = boxY * (0.9f + 0.08f * ry);
  return s;
}
__global__ void k_rain_claim(const Args A, int nspawn, unsigned seed) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nspawn) return;
  float x, y;
  const unsigned s = rain_draw(seed, k, A.boxX, A.boxY, x, y);
  atomicMax(&A.rainWinner[s % (unsigned)A.N], k);
}
__global__ void k_rain_apply(const Args A, int nspawn, unsigned seed) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nspawn) return;
  float x, y;
  const unsigned s = rain_draw(seed, k, A.boxX, A.boxY, x, y);
  const int i = (int)(s % (unsigned)A.N);
  if (A.rainWinner[i] != k) return;
  A.pos[i] = make_float2(x, y);
  A.vel[i] = make_float2(0.f, -0.5f * A.c0);
  A.rainWinner[i] = -1;
}

// k_rasterize, :363-374: particle counts on a W x 2H raster, y flipped
__global__ void k_rasterize(const float2 *__restrict__ pos, int N, int *grid2, int W, int H, float boxX, float boxY) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float2 p = pos[i];
  const int cx = (int)(p.x / boxX * (W - 1));
  const int sy = (int)((boxY - p.y) / boxY * (2 * H - 1));
  if ((unsigned)cx < (unsigned)W && (unsigned)sy < (unsigned)(2 * H)) atomicAdd(&grid2[sy * W + cx], 1);
}

} // namespace sph

// =====================================================================================
// C-ABI
// =====================================================================================
struct tausph {
  tausph_params p;
  int device;
  hipStream_t stream;
  bool own_stream;
  sph::Args a;
  int ntiles;        // scan tiles of the cell counts
  bool built;        // a sub-step has built the cell arrays (keys_s / ids_s / cellStart are defined)
  bool counted;      // the cells of the current positions are already counted (by the last k_forces): the build starts at the scan
  bool fuse_count;   // k_forces counts for the next build (TAU_SPH_FUSE_COUNT=0 switches it off)
  unsigned long long *pairs;   // device word of tausph_count_pairs (lazy)
  float tau, t;
  long step;
  float rain_carry;
  long rain_spawned;
  int *raster;
  size_t raster_n;
  int lpp;           // lanes per particle of the density / force passes (1 or 4)
};

extern "C" void tausph_params_default(tausph_params *P, int N) { // tau_sph.cu:49-85
  P->N = N; P->boxX = 1.0f; P->boxY = 1.0f; P->dTau = 1.0f; P->t0 = 1.0f; P->CFL = 1.0f;
  P->rho0 = 1.0f; P->c0 = 1.0f; P->gammaEOS = 1.0f; P->hMul = 2.0f; P->viscAlpha = 0.25f; P->gravity = 9.81f;
  P->useVisc = 1; P->useGrav = 1; P->viscSub = 1; P->seed = 69420;
  P->useXSPH = 0; P->xsphEps = 0.25f; P->rain = 0;   // rain: see tau_params.h (the reference's own default is on)
}

extern "C" int tausph_create(tausph_t **out, const tausph_params *P, int device, void *stream) {
  if (!out || !P) return tau::fail("tausph_create: null argument");
  if (P->N < 1) return tau::fail("tausph_create: N must be positive");
  if (!(P->boxX > 0.f) || !(P->boxY > 0.f) || !(P->hMul > 0.f)) return tau::fail("tausph_create: bad box / hMul");
  TAU_HIP(hipSetDevice(device));
  tausph *h = new (std::nothrow) tausph();
  if (!h) return tau::fail("tausph_create: out of host memory");
  tau::HandleGuard<tausph> guard{h, tausph_destroy};
  h->p = *P; h->device = device;
  h->own_stream = (stream == nullptr);
  if (h->own_stream) TAU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  else h->stream = (hipStream_t)stream;
  sph::Args &A = h->a;
  memset(&A, 0, sizeof(A));
  const float area = P->boxX * P->boxY;             // :573-576
  A.N = P->N; A.mass = (P->rho0 * area) / P->N;
  A.h = P->hMul * sqrtf(area / P->N);
  A.alpha = (float)(10.0f / (7.0f * M_PI * A.h * A.h)); // W_cubic's constant, :107
  A.cell = 2.0f * A.h;                               // ensure_cell_buffers, :512-521
  A.Gx = (int)ceilf(P->boxX / A.cell); A.Gy = (int)ceilf(P->boxY / A.cell);
  if (A.Gx < 1) A.Gx = 1;
  if (A.Gy < 1) A.Gy = 1;
  A.M = A.Gx * A.Gy;
  A.rho0 = P->rho0; A.c0 = P->c0; A.gammaEOS = P->gammaEOS; A.viscAlpha = P->viscAlpha;
  A.gx = 0.f; A.gy = -(P->useGrav ? P->gravity : 0.f);
  A.boxX = P->boxX; A.boxY = P->boxY; A.useVisc = P->useVisc; A.useGrav = P->useGrav;
  const size_t N = (size_t)P->N;
  TAU_HIP(hipMalloc(&A.pos, N * sizeof(float2))); TAU_HIP(hipMalloc(&A.vel, N * sizeof(float2)));
  TAU_HIP(hipMalloc(&A.acc, N * sizeof(float2)));
  TAU_HIP(hipMalloc(&A.s, N * sizeof(float))); TAU_HIP(hipMalloc(&A.press, N * sizeof(float)));
  TAU_HIP(hipMalloc(&A.cellOf, N * sizeof(int)));
  TAU_HIP(hipMalloc(&A.keys, N * 4)); TAU_HIP(hipMalloc(&A.ids, N * 4));
  TAU_HIP(hipMalloc(&A.keys_s, N * 4)); TAU_HIP(hipMalloc(&A.ids_s, N * 4));
  TAU_HIP(hipMalloc(&A.cellStart, ((size_t)A.M + 1) * sizeof(int)));
  TAU_HIP(hipMalloc(&A.tmpKey, N * 4)); TAU_HIP(hipMalloc(&A.tmpId, N * 4));
  h->ntiles = (int)(((size_t)A.M + 1 + sph::SCAN_TILE - 1) / sph::SCAN_TILE);
  TAU_HIP(hipMalloc(&A.cellCount, (size_t)h->ntiles * sph::SCAN_TILE * sizeof(unsigned)));
  TAU_HIP(hipMalloc(&A.tileSum, (size_t)h->ntiles * sizeof(unsigned)));
  TAU_HIP(hipMemsetAsync(A.cellCount, 0, (size_t)h->ntiles * sph::SCAN_TILE * sizeof(unsigned), h->stream));   // k_scan leaves it zero again
  TAU_HIP(hipMalloc(&A.recA, N * sizeof(float4))); TAU_HIP(hipMalloc(&A.recB, N * sizeof(float2)));
  TAU_HIP(hipMalloc(&A.recP, N * sizeof(float2)));
  TAU_HIP(hipMalloc(&A.nbrMask, N * sizeof(unsigned) * 12));
  // overflow-block masks (720 B per particle — 3 GB at 4 M particles; written and read only where a row holds more than 128
  // candidates).  An optimisation, not a requirement: without them the force pass scans the overflow blocks again (MODE 0), same
  // results — so a device that cannot spare the buffer runs without it instead of failing the create.
  if (!(getenv("TAU_SPH_OVFMASK") && atoi(getenv("TAU_SPH_OVFMASK")) == 0)) {
    if (hipMalloc(&A.ovfMask, N * sizeof(unsigned) * 3 * sph::OVW) != hipSuccess) {
      (void)hipGetLastError();
      A.ovfMask = nullptr;
    }
  }
  A.xsphEps = P->xsphEps;
  if (P->useXSPH && P->xsphEps > 0.f) TAU_HIP(hipMalloc(&A.recA2, N * sizeof(float4)));
  if (P->rain) {
    TAU_HIP(hipMalloc(&A.rainWinner, N * sizeof(int)));
    TAU_HIP(hipMemsetAsync(A.rainWinner, 0xff, N * sizeof(int), h->stream));
  }
  TAU_HIP(hipMemsetAsync(A.acc, 0, N * sizeof(float2), h->stream));
  TAU_HIP(hipMemsetAsync(A.s, 0, N * sizeof(float), h->stream));
  TAU_HIP(hipMemsetAsync(A.press, 0, N * sizeof(float), h->stream));
  // one lane per particle does not fill 1 024 SIMDs below ~100 k particles (lattice sub-step, 1 vs 4 lanes: 4 096:
  // 110 vs 69 us, 65 536: 134 vs 112 us, 262 144: 226 vs 289 us; the reference's compressed default run at 65 536:
  // 0.64-0.87 vs 0.29-0.45 ms)
  h->counted = false;
  h->built = false;
  h->fuse_count = !(getenv("TAU_SPH_FUSE_COUNT") && atoi(getenv("TAU_SPH_FUSE_COUNT")) == 0);
  h->lpp = (P->N < (1 << 17)) ? 4 : 1;
  if (const char *e = getenv("TAU_SPH_LPP")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) h->lpp = v; }

  h->tau = 0.f; h->t = P->t0 * expf(h->tau); h->step = 0; // :577-578
  *out = guard.release();
  return 0;
}
extern "C" void tausph_destroy(tausph_t *h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  sph::Args &A = h->a;
  hipFree(A.pos); hipFree(A.vel); hipFree(A.acc); hipFree(A.s); hipFree(A.press); hipFree(A.cellOf);
  hipFree(A.keys); hipFree(A.ids); hipFree(A.keys_s); hipFree(A.ids_s); hipFree(A.cellStart);
  hipFree(A.recA); hipFree(A.recB); hipFree(A.recP); hipFree(A.nbrMask); hipFree(A.ovfMask); hipFree(A.recA2); hipFree(A.rainWinner); hipFree(h->raster);
  hipFree(A.tmpKey); hipFree(A.tmpId); hipFree(A.cellCount); hipFree(A.tileSum); hipFree(h->pairs);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int tausph_upload(tausph_t *h, const float *pos_xy, const float *vel_xy) {
  TAU_HIP(hipSetDevice(h->device));
  if (h->counted) {   // the positions the last k_forces counted are being replaced
    TAU_HIP(hipMemsetAsync(h->a.cellCount, 0, (size_t)h->ntiles * sph::SCAN_TILE * sizeof(unsigned), h->stream));
    h->counted = false;
  }
  size_t b = (size_t)h->p.N * sizeof(float2);
  TAU_HIP(hipMemcpyAsync(h->a.pos, pos_xy, b, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipMemcpyAsync(h->a.vel, vel_xy, b, hipMemcpyHostToDevice, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int tausph_reset_particles(tausph_t *h) { // reset_particles, :493-510 (host) + H2D :567-570
  const tausph_params &P = h->p;
  std::vector<float> pos(2 * (size_t)P.N), vel(2 * (size_t)P.N, 0.f);
  std::mt19937 rng(P.seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  int nSide = (int)sqrtf((float)P.N);
  int nx = nSide, ny = (P.N + nSide - 1) / nSide;
  float padX = 0.05f * P.boxX, padY = 0.05f * P.boxY;
  float width = P.boxX - 2 * padX, height = 0.6f * P.boxY - padY;
  for (int i = 0; i < P.N; ++i) {
    int ix = i % nx, iy = i / nx;
    float fx = (ix + 0.5f) / nx, fy = (iy + 0.5f) / ny;
    float x = padX + fx * width, y = padY + fy * height;
    x += (U(rng) - 0.5f) * 0.2f * width / nx;
    y += (U(rng) - 0.5f) * 0.2f * height / ny;
    pos[2 * (size_t)i] = x; pos[2 * (size_t)i + 1] = y;
  }
  h->tau = 0.f; h->t = P.t0 * expf(h->tau); h->step = 0;
  return tausph_upload(h, pos.data(), vel.data());
}

extern "C" int tausph_download(tausph_t *h, float *pos_xy, float *vel_xy, float *acc_xy, float *s, float *press,
                               int32_t *cellOf) {
  TAU_HIP(hipSetDevice(h->device));
  size_t N = (size_t)h->p.N;
  if (pos_xy) TAU_HIP(hipMemcpyAsync(pos_xy, h->a.pos, N * 8, hipMemcpyDeviceToHost, h->stream));
  if (vel_xy) TAU_HIP(hipMemcpyAsync(vel_xy, h->a.vel, N * 8, hipMemcpyDeviceToHost, h->stream));
  if (acc_xy) TAU_HIP(hipMemcpyAsync(acc_xy, h->a.acc, N * 8, hipMemcpyDeviceToHost, h->stream));
  if (s) TAU_HIP(hipMemcpyAsync(s, h->a.s, N * 4, hipMemcpyDeviceToHost, h->stream));
  if (press) TAU_HIP(hipMemcpyAsync(press, h->a.press, N * 4, hipMemcpyDeviceToHost, h->stream));
  if (cellOf) {
    if (h->built) hipLaunchKernelGGL(sph::k_cell_of, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->a);
    else hipLaunchKernelGGL(sph::k_cell_of_pos, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->a);   // no sub-step yet: nothing sorted
    TAU_LAUNCH_CHECK("sph::k_cell_of");
    TAU_HIP(hipMemcpyAsync(cellOf, h->a.cellOf, N * 4, hipMemcpyDeviceToHost, h->stream));
  }
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tausph_state_written(tausph_t *h);
extern "C" int tausph_state_ptrs(tausph_t *h, float **pos, float **vel, float **acc, float **s, float **press) {
  // Whoever holds these pointers may move particles behind the engine's back, and the force pass's fused count of the NEXT
  // build's cells (k_forces, A.countNext) would then scatter by stale cells while the records gather the new positions —
  // neighbour pairs missed without a word.  From the first hand-out on, this handle counts the cells from the positions at
  // every sub-step again (k_count: ~28 us of a 1.2 ms sub-step at 4 M particles), as the reference's k_build_cells reads them
  // every sub-step; tausph_state_written is then only needed for the sub-step already enqueued.
  if (pos && h->fuse_count) {
    h->fuse_count = false;
    if (tausph_state_written(h)) return 1;
  }
  if (pos) *pos = (float *)h->a.pos;
  if (vel) *vel = (float *)h->a.vel;
  if (acc) *acc = (float *)h->a.acc;
  if (s) *s = h->a.s;
  if (press) *press = h->a.press;
  return 0;
}
extern "C" int tausph_state_written(tausph_t *h) {   // the caller moved particles through tausph_state_ptrs: count the cells again
  if (!h) return tau::fail("tausph_state_written: null handle");
  TAU_HIP(hipSetDevice(h->device));
  if (h->counted) {
    TAU_HIP(hipMemsetAsync(h->a.cellCount, 0, (size_t)h->ntiles * sph::SCAN_TILE * sizeof(unsigned), h->stream));
    h->counted = false;
  }
  return 0;
}
extern "C" int tausph_grid(tausph_t *h, int *Gx, int *Gy, float *cell, float *hh, float *mass) {
  if (Gx) *Gx = h->a.Gx;
  if (Gy) *Gy = h->a.Gy;
  if (cell) *cell = h->a.cell;
  if (hh) *hh = h->a.h;
  if (mass) *mass = h->a.mass;
  return 0;
}

extern "C" int tausph_substep_async(tausph_t *h, float dt) { // the five launches of :676-701
  TAU_HIP(hipSetDevice(h->device));
  sph::Args A = h->a;
  A.dt = dt;
  const unsigned gs = (unsigned)((A.N + 255) / 256);
  // the cell build (k_clear_heads + k_build_cells, :159-176): counting sort, four or five small launches
  if (!h->counted) {
    hipLaunchKernelGGL(sph::k_count, dim3(gs), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_count");
  }
  A.countNext = h->fuse_count ? 1 : 0;
  h->counted = h->fuse_count;
  h->built = true;
  if (h->ntiles > sph::DIRECT_TILES) {
    hipLaunchKernelGGL(sph::k_tile_sums, dim3((unsigned)h->ntiles), dim3(sph::SCAN_T), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_tile_sums");
  }
  hipLaunchKernelGGL(sph::k_scan, dim3((unsigned)h->ntiles), dim3(sph::SCAN_T), 0, h->stream, A, h->ntiles);
  TAU_LAUNCH_CHECK("sph::k_scan");
  hipLaunchKernelGGL(sph::k_scatter, dim3(gs), dim3(256), 0, h->stream, A);
  TAU_LAUNCH_CHECK("sph::k_scatter");
  hipLaunchKernelGGL(sph::k_rank_gather, dim3(gs), dim3(256), 0, h->stream, A, h->ntiles * sph::SCAN_TILE);
  TAU_LAUNCH_CHECK("sph::k_rank_gather");
  // four lanes per particle while there are too few particles to fill the chip with one
  if (h->lpp == 4) {
    const unsigned gq = (unsigned)((A.N + 63) / 64);
    hipLaunchKernelGGL(sph::k_density<4>, dim3(gq), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_density");
    hipLaunchKernelGGL(sph::k_forces<4>, dim3(gq), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_forces");
  } else if (h->lpp == 2) {
    const unsigned gh = (unsigned)((A.N + 127) / 128);
    hipLaunchKernelGGL(sph::k_density<2>, dim3(gh), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_density");
    hipLaunchKernelGGL(sph::k_forces<2>, dim3(gh), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_forces");
  } else {
    hipLaunchKernelGGL(sph::k_density<1>, dim3(gs), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_density");
    hipLaunchKernelGGL(sph::k_forces<1>, dim3(gs), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_forces");
  }
  if (A.recA2) { // :698-704
    hipLaunchKernelGGL(sph::k_xsph, dim3(gs), dim3(256), 0, h->stream, A);
    TAU_LAUNCH_CHECK("sph::k_xsph");
  }
  if (h->p.rain) { // :706-716
    h->rain_carry += 0.02f * h->p.N * dt;
    int nspawn = (int)h->rain_carry;
    h->rain_carry -= nspawn;
    if (nspawn > 0) {
      if (h->counted) {   // drops move particles after they were counted: the next build counts again
        TAU_HIP(hipMemsetAsync(A.cellCount, 0, (size_t)h->ntiles * sph::SCAN_TILE * sizeof(unsigned), h->stream));
        h->counted = false;
      }
      const unsigned seed = (unsigned)(h->p.seed + (int)h->step), gr = (unsigned)((nspawn + 127) / 128);
      hipLaunchKernelGGL(sph::k_rain_claim, dim3(gr), dim3(128), 0, h->stream, A, nspawn, seed);
      TAU_LAUNCH_CHECK("sph::k_rain_claim");
      hipLaunchKernelGGL(sph::k_rain_apply, dim3(gr), dim3(128), 0, h->stream, A, nspawn, seed);
      TAU_LAUNCH_CHECK("sph::k_rain_apply");
      h->rain_spawned += nspawn;
    }
  }
  return 0;
}

extern "C" float tausph_dt(tausph_t *h) { // :666-669
  float dt_try = h->t * h->p.dTau;
  float dt_cfl = h->p.CFL * h->a.h / (h->p.c0 * (1.0f + 2.0f * h->p.viscAlpha));
  return fminf(dt_try, dt_cfl);
}

extern "C" int tausph_step_async(tausph_t *h, int nsteps) { // host loop body, :665-721
  for (int n = 0; n < nsteps; n++) {
    int K = (h->p.viscSub > 0 ? h->p.viscSub : 1);
    float dt_sub = tausph_dt(h) / K;
    float dTau_accum = 0.f;
    for (int k = 0; k < K; ++k) {
      if (tausph_substep_async(h, dt_sub)) return 1;
      float dTau_actual = dt_sub / fmaxf(h->t, 1e-9f);
      dTau_accum += dTau_actual;
      h->t = h->p.t0 * expf(h->tau + dTau_accum);
    }
    h->tau += dTau_accum;
    h->step++;
  }
  return 0;
}
extern "C" int tausph_rasterize(tausph_t *h, int W, int H, int32_t *host_grid2) { // k_clear_grid + k_rasterize, :357-374
  if (W < 1 || H < 1 || !host_grid2) return tau::fail("tausph_rasterize: bad raster %dx%d", W, H);
  TAU_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)W * 2 * (size_t)H;
  if (h->raster_n < n) { hipFree(h->raster); h->raster = nullptr; TAU_HIP(hipMalloc(&h->raster, n * sizeof(int))); h->raster_n = n; }
  TAU_HIP(hipMemsetAsync(h->raster, 0, n * sizeof(int), h->stream));
  hipLaunchKernelGGL(sph::k_rasterize, dim3((unsigned)((h->p.N + 255) / 256)), dim3(256), 0, h->stream, (const float2 *)h->a.pos,
                     h->p.N, h->raster, W, H, h->p.boxX, h->p.boxY);
  TAU_LAUNCH_CHECK("sph::k_rasterize");
  TAU_HIP(hipMemcpyAsync(host_grid2, h->raster, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int64_t tausph_rain_spawned(tausph_t *h) { return (int64_t)h->rain_spawned; }
extern "C" int tausph_count_pairs(tausph_t *h, int64_t *ordered_pairs) {
  if (!h || !ordered_pairs) return tau::fail("tausph_count_pairs: null argument");
  if (!h->built) return tau::fail("tausph_count_pairs: no sub-step has built the cells yet");
  TAU_HIP(hipSetDevice(h->device));
  if (!h->pairs) TAU_HIP(hipMalloc(&h->pairs, sizeof(unsigned long long)));
  TAU_HIP(hipMemsetAsync(h->pairs, 0, sizeof(unsigned long long), h->stream));
  hipLaunchKernelGGL(sph::k_count_pairs, dim3((unsigned)((h->a.N + 255) / 256)), dim3(256), 0, h->stream, h->a, h->pairs);
  TAU_LAUNCH_CHECK("sph::k_count_pairs");
  unsigned long long v = 0;
  TAU_HIP(hipMemcpyAsync(&v, h->pairs, sizeof v, hipMemcpyDeviceToHost, h->stream));
  TAU_HIP(hipStreamSynchronize(h->stream));
  *ordered_pairs = (int64_t)v;
  return 0;
}
extern "C" int tausph_sync(tausph_t *h) {
  TAU_HIP(hipSetDevice(h->device));
  TAU_HIP(hipStreamSynchronize(h->stream));
  return 0;
}
extern "C" int tausph_step(tausph_t *h, int nsteps) {
  if (tausph_step_async(h, nsteps)) return 1;
  return tausph_sync(h);
}
extern "C" int tausph_get_clock(tausph_t *h, float *t, float *tau, int64_t *step) {
  if (t) *t = h->t;
  if (tau) *tau = h->tau;
  if (step) *step = h->step;
  return 0;
}
