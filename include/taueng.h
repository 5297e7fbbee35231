/* taueng.h — C-ABI of libtaueng, the MI355X (gfx950) explicit time-stepping engine.
 *
 * The reference (seanwevans/fluid-sims) has no library or FFI: every simulator is a
 * stand-alone program whose `main` owns the device buffers and launches the step
 * kernels itself.  This header is the seam a reference `main` (or any FFI: cgo, JNI,
 * ctypes) binds instead of those launches.  Each entry point names the reference
 * code it replaces (file:line in the reference tree).
 *
 * Conventions:
 *   - opaque handle per simulator; the library owns device buffers, the caller owns
 *     host buffers; plain pointers and sizes only, no C++/torch types;
 *   - every function returns 0 on success, non-zero on error; tau_last_error() gives
 *     the message (the reference prints to stderr and exits: `ck`,
 *     tau_hypersonic_3d_cuda.cu:62-67 — the thin C drivers reproduce that);
 *   - state arrays keep the reference layout: row-major y*W+x (2D),
 *     (z*ny+y)*nx+x (3D), ping-pong by pointer swap inside the handle;
 *   - a call is synchronous-looking to the caller unless its name ends in _async;
 *     all work of a handle runs on that handle's stream.
 */
#ifndef TAUENG_H
#define TAUENG_H

#include "tau_params.h"
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *tau_last_error(void);
/* 1 if a gfx950 device is visible to the HIP runtime, else 0 (never fails) */
int tau_device_available(void);
/* number of devices the HIP runtime shows (0 with an error text when there is no runtime / no device) */
int tau_device_count(int *n);
int tau_version(void);
/* Test seam (host only, no device): the guided chunk schedule of the row marches (2D Euler, Burgers, shallow water;
 * csrc/tau_common.hip tau::guided_chunks).  H rows in 8 bands, each cut into chunks of descending length: a chunk is the band's
 * remaining rows x nstrips shared out over slots_per_xcd resident waves, clamped to [lmin, lmax].  Writes *nchunks + 1 row starts
 * (chunk c = rows [starts[c], starts[c + 1]), possibly empty) if `starts` is non-null and `cap` entries suffice. */
int tau_guided_chunks(int H, int nstrips, int slots_per_xcd, int lmin, int lmax, int *starts, int cap, int *nchunks);

/* =====================================================================
 * 3D two-temperature hypersonic Euler — replaces the launches in
 * tau_hypersonic_3d_cuda.cu:1559-1605 (setup) and :1678-1713 (step loop).
 *
 * Z-slab form: a handle owns local planes [z0, z0+nzl) of the global nz
 * (nzl = nz, z0 = 0 for a single GPU) plus a 3-plane halo on each side
 * (WENO_HALO, :58).  Field f of the state is one contiguous device array of
 * (nzl+6)*ny*nx floats, halo planes first; tau3d_state_ptrs() returns the
 * pointer to local plane 0, i.e. the reference's (z*ny+y)*nx+x array.
 * Field order: xi, phix, phiy, phiz, lam, zet (:1564-1565).
 * ===================================================================== */
typedef struct tau3d tau3d_t;

void tau3d_params_default(tau3d_params *p, int nx, int ny, int nz); /* :1531-1557 */

/* alloc + constants upload (replaces cudaMalloc x14 + cudaMemcpyToSymbol(P), :1559-1589).
 * stream: a hipStream_t to run on, or NULL for a private stream. */
int tau3d_create(tau3d_t **out, const tau3d_params *p, int z0, int nzl, int device, void *stream);
void tau3d_destroy(tau3d_t *h);

/* k_build_solid_mask + k_init (:1601-1605); resets the clock to t=1e-5, d_tau=1e-3 (:1635-1636).
 * mode 0 = reference quiescent start; mode 1 = synthetic developed flow (every fluid cell at
 * the full inflow state, SURVEY §8d input (ii)). */
int tau3d_init(tau3d_t *h, int mode);

/* host <-> device copies of the nzl interior planes, reference layout, 6 fields */
int tau3d_upload_state(tau3d_t *h, const float *const host[6]);
int tau3d_download_state(tau3d_t *h, float *const host[6]);
int tau3d_download_solid(tau3d_t *h, uint8_t *host);
/* same for a range of local planes [zl_lo, zl_hi), halo planes allowed: -3 <= zl_lo < zl_hi <= nzl+3 */
int tau3d_upload_planes(tau3d_t *h, int zl_lo, int zl_hi, const float *const host[6]);
int tau3d_download_planes(tau3d_t *h, int zl_lo, int zl_hi, float *const host[6]);

/* device pointers (current ping-pong side): interior plane 0 of each field, the solid
 * mask (interior plane 0) and the device clock block */
int tau3d_state_ptrs(tau3d_t *h, float *dptr[6], uint8_t **solid);

int tau3d_get_clock(tau3d_t *h, tau3d_clock *out);
int tau3d_set_clock(tau3d_t *h, const tau3d_clock *in);

/* n full steps = the reference loop body :1680-1711 (log-time clock, k_step, d_tau
 * controller, swap), single domain only (nzl == nz): halos are filled periodically on
 * device.  The controller runs on the device — no per-step host round trip. */
int tau3d_step(tau3d_t *h, int nsteps, tau3d_clock *out);
int tau3d_step_async(tau3d_t *h, int nsteps);   /* same, returns after enqueueing */

/* One k_step launch with an explicit dt / inflow_gain (the kernel call at :1689-1691 alone),
 * for parity tests: halos periodic (single domain), result swapped in, *maxs = max wavespeed. */
int tau3d_step_explicit(tau3d_t *h, float dt, float inflow_gain, float *maxs);

/* ---- multi-GPU pieces (no reference counterpart: SURVEY §8e).  One step on rank r is
 *   tau3d_clock_begin_async            t*=exp(d_tau), dt, gain; zero the max word
 *   tau3d_step_edges_async(E)          planes [0,E) and [nzl-E,nzl), E >= 3, one launch — need the halos
 *   <caller: exchange tau3d_halo_send_ptr -> neighbour's tau3d_halo_recv_ptr>
 *   tau3d_step_range_async(interior)   planes [E,nzl-E)
 *   <caller: all-reduce(max) the TWO words at tau3d_max_ptr, as floats>
 *   tau3d_clock_end_async              d_tau controller, swap
 * The caller (bench.py / the driver) orders these on streams it owns. */
int tau3d_clock_begin_async(tau3d_t *h);
int tau3d_step_range_async(tau3d_t *h, int zl_lo, int zl_hi, void *stream);
/* both Z-slab edges, planes [0, depth) and [nzl-depth, nzl), in ONE launch (twice the workgroups of an edge launch:
 * the 768 resident slots of the chip fill better); the whole slab if 2*depth >= nzl.  depth >= 3. */
int tau3d_step_edges_async(tau3d_t *h, int depth, void *stream);
int tau3d_clock_end_async(tau3d_t *h);
/* The same Z-slab step in five dispatches (four with TAU3D_SLAB_XY_FIRST=1: the x/y flux kernel then runs once, over all
 * planes, inside the edges piece — but the exchange has less to hide behind) and two collectives (what fluid-sims_amd/slab.py drives):
 *   tau3d_slab_begin_async        ONE kernel: d_tau controller of the previous step (if one is pending: it needs the
 *                                 all-reduced max word), clock of this step, received halos unpacked into the current state
 *   tau3d_slab_edges_async(E)     planes [0,E) and [nzl-E,nzl): [large planes: their x/y flux kernel, then] one launch that
 *                                 steps both edges and also writes the new boundary planes into the packed send buffers
 *                                 (no separate pack kernel)
 *   <caller: post the send / recv of tau3d_halo_buf_ptr(kind, side)>
 *   tau3d_slab_interior_async(E)  planes [E, nzl-E) [x/y flux kernel + z kernel]: overlaps the exchange
 *   <caller: all-reduce(max) the two words at tau3d_max_ptr>
 *   tau3d_slab_end_async          swap; host bookkeeping only — the controller rides on the next tau3d_slab_begin_async
 *                                 (or on tau3d_get_clock, which flushes it) */
int tau3d_slab_begin_async(tau3d_t *h);
int tau3d_slab_edges_async(tau3d_t *h, int depth);
int tau3d_slab_interior_async(tau3d_t *h, int depth);
int tau3d_slab_end_async(tau3d_t *h);
/* The pipelined step of the direct-halo ring (csrc/ring.hip): begin, tau3d_slab_xy_async (x/y fluxes of ALL planes: they read
 * no halo plane, so the halos of the step before may still be arriving), <halos landed>, tau3d_slab_z_async (z fluxes + update
 * of all planes; with the fused small-plane kernel: the whole step), <all-reduce(max); copy the boundary planes out>, end. */
int tau3d_slab_clock_async(tau3d_t *h);   /* tau3d_slab_begin_async without the unpack (packed transports: tau3d_unpack_halos_async follows before the z piece) */
int tau3d_slab_xy_async(tau3d_t *h);
int tau3d_slab_z_async(tau3d_t *h);
/* Round 5: the x/y fluxes may also be issued AHEAD of the step's clock — tau3d_slab_xy_async, <all-reduce of the step before
 * landed>, begin / clock, tau3d_slab_xy_fix_async, z.  The x/y flux kernel reads one word of the clock block, the field range that
 * picks the WENO weight form; ahead of the clock it sees the range of the step before, and the clock's commit records whether the
 * new range is on the other side of the limit.  tau3d_slab_xy_fix_async repeats the fluxes in that case (never in a sane run: one
 * near-empty launch) — the step's results do not depend on when the first launch ran.  The ring's pipelined steps use it to take
 * the all-reduce off the critical path. */
int tau3d_slab_xy_fix_async(tau3d_t *h);
/* test hook: overwrite the range word a launch ahead of the clock reads (the next clock commit replaces it) */
int tau3d_debug_set_fmax_in(tau3d_t *h, float v);
/* fill own halos from own interior (periodic single domain) */
int tau3d_fill_halo_periodic_async(tau3d_t *h);
/* side 0 = low-z, 1 = high-z; which = 0 current (input) state, 1 = next (output) state.
 * send: first/last 3 INTERIOR planes; recv: the halo planes.  Each is 3*ny*nx floats. */
/* (A caller that fills halo planes through tau3d_halo_recv_ptr itself — instead of the packed buffers and
 * tau3d_unpack_halos_async / tau3d_slab_begin_async — must call tau3d_state_written before the next step, so that the
 * field range the WENO weight form depends on covers what it wrote.) */
/* Direct halos (the ring's IPC transport): with tau3d_set_halo_direct(h, 1) the neighbours' boundary planes are written
 * straight into this slab's halo planes (tau3d_halo_recv_ptr of the NEXT state) by whoever drives the exchange, so
 * tau3d_slab_begin_async shrinks to the one-thread controller / clock kernel and tau3d_slab_edges_async fills no packed
 * send buffer.  tau3d_state_group: base address, size and per-field stride (floats) of the ONE allocation that holds the six
 * fields of the current (which = 0) / next (1) state — what a peer process maps with hipIpcOpenMemHandle — and which of the
 * handle's two ping-pong allocations it is (*index 0 / 1; the two swap roles with every step).  Any out pointer may be NULL. */
int tau3d_set_halo_direct(tau3d_t *h, int on);
int tau3d_state_group(tau3d_t *h, int which, void **base, size_t *bytes, size_t *field_stride, int *index);
int tau3d_halo_send_ptr(tau3d_t *h, int which, int field, int side, float **p);
int tau3d_halo_recv_ptr(tau3d_t *h, int which, int field, int side, float **p);
/* Packed exchange buffers: 6 fields x 3 planes contiguous, one send and one recv buffer per side, so
 * a step needs 2 sends + 2 recvs instead of 24.  pack: boundary planes of state `which` -> send buffers of
 * both sides; unpack: recv buffers -> halo planes of state `which`.  kind 0 = send, 1 = recv. */
int tau3d_pack_halos_async(tau3d_t *h, int which);
int tau3d_unpack_halos_async(tau3d_t *h, int which);
int tau3d_halo_buf_ptr(tau3d_t *h, int kind, int side, float **p, size_t *nfloats);
/* Two consecutive device words, both non-negative floats: [0] the max wavespeed of the step in flight (the
 * controller input of :1697-1704), [1] the largest |primitive| that step wrote — or, after init / upload, the
 * largest in the state (the step kernel picks its WENO weight form from it).  A ring all-reduces
 * (max) both after the step's launches, and once after init / upload before the first step. */
int tau3d_max_ptr(tau3d_t *h, float **p);
/* Diagnostics (synchronises): *read_max = the largest |primitive| the last launched step was told its input holds,
 * *written_max = the largest it (or init / upload since) wrote, *fast_form = 1 if that launch took the
 * common-denominator WENO weights, 0 if the reciprocal form (input range above 2.5e3).  Any pointer may be NULL. */
int tau3d_field_range(tau3d_t *h, float *read_max, float *written_max, int *fast_form);
/* Uniform-region exits of the kernel pair (round 6): a tile of k_flux_xy whose cells and x/y halo cells all hold one encoded state
 * has a divergence of exactly +0 in every cell (every face takes the same flux) and writes a flag instead of computing it; a wave
 * of k_update_z whose five stencil planes hold one state per column reads the reconstruction's result (that state, exactly)
 * instead of recomputing it.  Same bits as the full path (tests/test_gpu_tau3d.py: exits on == exits off, byte for byte); the
 * reference has no counterpart — its k_step (tau_hypersonic_3d_cuda.cu:987-1359) evaluates every face everywhere.
 * TAU3D_UNIFORM_EXITS=0 at tau3d_create switches them off.  tau3d_uniform_tiles: the tiles the LAST step flagged / all tiles. */
int tau3d_uniform_tiles(tau3d_t *h, long *uniform, long *tiles, int *enabled);
/* Predicted-uniform tiles (round 6; no reference counterpart): after a whole-domain step, the tiles whose neighbourhood held one
 * encoded state are flagged for the next step without being looked at again, and k_flux_xy is launched over the LIST of the
 * others.  TAU3D_TILE_LIST at tau3d_create: 0 off, 1 (default) on, 2 verify (predictions checked against a k_flux_xy over every
 * tile).  mode: what this handle does (0 also for ragged tiles; a slab predicts no plane within three of its edges); listed: length of the list made by the last step (-1: none
 * valid); checked / mismatches: mode 2's tally (mismatches must stay 0).  Waits for the stream.  Same bits in every mode. */
int tau3d_tile_list_stats(tau3d_t *h, int *mode, long *listed, long *tiles, long *checked, long *mismatches);
/* The pointers of tau3d_state_ptrs are for reading.  A caller that does write the state (or the solid mask) through them
 * says so here before the next step (tau3d_init / tau3d_upload_* do it themselves): the field range is measured again and
 * the static solid-free tile flags of the x/y flux kernel are rebuilt from the mask. */
int tau3d_state_written(tau3d_t *h);
int tau3d_sync(tau3d_t *h);
/* what the handle was created with: its slab [z0, z0+nzl) of the global nz, its device and the stream its work runs on */
int tau3d_slab_info(tau3d_t *h, int *z0, int *nzl, int *nz, int *device, void **stream);
/* 1 if a step of this handle is the kernel pair k_flux_xy + k_update_z, 0 if the fused k_step */
int tau3d_is_split(tau3d_t *h);
/* choose the form of the step for this handle: 1 = the kernel pair, 0 = the fused kernel (tau3d_create picks the pair from
 * 128^2 cells per plane; TAU3D_SPLIT=0/1 in the environment overrides that default).  Both give the same results to rounding. */
int tau3d_set_split(tau3d_t *h, int on);

/* ---- The Z-slab ring, in the library (csrc/ring.hip): one process per GPU, each owning one slab handle; the ring adds
 * the halo exchange with both z neighbours and the all-reduce(max) of the two words of tau3d_max_ptr, issued from C on a
 * private stream beside the handle's — replaces, for N GPUs, the loop body tau_hypersonic_3d_cuda.cu:1678-1713 (the
 * reference is single-GPU: SURVEY §8e).  Per step: slab_begin, slab_edges(E), [exchange posted], slab_interior(E)
 * overlapping it, [all-reduce], slab_end — enqueued without any host synchronisation.
 *   transport  TAU3D_RING_RCCL  ncclSend / ncclRecv / ncclAllReduce (librccl bound at run time; world 1 talks to itself)
 *              TAU3D_RING_HOST  staged through the rendezvous file by the host (ranks may share a device; tests, fallback)
 *              TAU3D_RING_LOCAL world 1: device copies, no collective
 *              TAU3D_RING_IPC   direct halos: each rank maps its neighbours' state (hipIpc*MemHandle) and copies its new boundary
 *                               planes straight into their halo planes (hipMemcpyAsync: SDMA over xGMI, no CU, no pack / unpack);
 *                               RCCL only for the 8-byte all-reduce, which also orders the copies (csrc/ring.hip)
 *              TAU3D_RING_IPC_HOSTMAX  the same copies with the all-reduce staged by the host (ranks may share a device: tests)
 *   job_key    non-zero and unique per launch when world > 1 (tells this job's rendezvous file from a stale one)
 *   rendezvous path of a file rank 0 creates and the others map (ncclUniqueId, barrier, staging); every rank of a job
 *              passes the same path and job_key, a later job a different key.  May be NULL when world == 1. */
enum { TAU3D_RING_RCCL = 0, TAU3D_RING_HOST = 1, TAU3D_RING_LOCAL = 2, TAU3D_RING_IPC = 3, TAU3D_RING_IPC_HOSTMAX = 4 };
typedef struct tau3d_ring tau3d_ring_t;
/* contiguous split of nz planes over `world` ranks (every slab needs >= 6 planes) */
int tau3d_slab_bounds(int nz, int world, int rank, int *z0, int *nzl);
int tau3d_ring_create(tau3d_ring_t **out, tau3d_t *h, int rank, int world, int transport, const char *rendezvous, uint64_t job_key);
void tau3d_ring_destroy(tau3d_ring_t *r);          /* does not destroy the slab handle */
/* exchange the halos of the current state and agree on its field range; tau3d_ring_step_async does it itself after
 * create / tau3d_ring_invalidate (call that after tau3d_init / tau3d_upload_* on the handle) */
int tau3d_ring_prime(tau3d_ring_t *r);
int tau3d_ring_invalidate(tau3d_ring_t *r);
int tau3d_ring_step_async(tau3d_ring_t *r, int nsteps);
int tau3d_ring_finish(tau3d_ring_t *r);            /* waits for both streams */
int tau3d_ring_get_clock(tau3d_ring_t *r, tau3d_clock *out);
int tau3d_ring_barrier(tau3d_ring_t *r);           /* host barrier over the ranks (through the rendezvous file) */
/* RCCL's version (ncclGetVersion), the rank count of the communicator (ncclCommCount), planes per edge launch, the
 * librccl the ring bound to; any pointer may be NULL */
/* HIP events on the ring's communication stream around the exchange and around the all-reduce of every step (default
 * schedule): enable, step, finish, read the sums over `steps` timed steps (<= 256 per enable).  The all-reduce follows the
 * exchange on that stream and completes when the slowest rank's has — it carries the skew between ranks.  (The reference has no
 * counterpart: its loop is single-GPU, tau_hypersonic_3d_cuda.cu:1680-1711; these are bench.py's per-rank ring figures.) */
int tau3d_ring_timing_enable(tau3d_ring_t *r, int on);
int tau3d_ring_timing_read(tau3d_ring_t *r, double *exchange_ms, double *allreduce_ms, int *steps);
int tau3d_ring_info(tau3d_ring_t *r, int *rccl_version, int *comm_ranks, int *edge_planes, char *lib_path, size_t lib_path_len);

/* The export path of th3cs.cu (:1193-1222): the volume of the last tau3d_vis — th3cs uses mode 0, its
 * k_schlieren_export (:641-673) is that field — mapped to 8-bit palette indices,
 * (int)(powf((v - min) / fmaxf(max - min, 1e-12f), gamma) * 255) clamped to 0..255 (th3cs: gamma = 0.65).  The
 * reference downloads the float volume and does min / max / map on the host; here one byte per voxel comes back
 * (host_idx: nx*ny*nzl bytes, may be NULL to leave them on the device).  *mn, *mx: the range used. */
int tau3d_palette_indices(tau3d_t *h, float gamma, uint8_t *host_idx, float *mn, float *mx);
/* Visualisation fields — replaces k_vis (tau_hypersonic_3d_cuda.cu:800-905, launched :1715-1716),
 * slice_to_rgba (:1416-1442, called per slice :1735-1739) and k_outflow_reflection_metric (:1389-1408,
 * launched :1724-1731).  mode = the reference's VisMode (:784-794): 0 |grad rho|, 1 log(1+rho),
 * 2 log(1+p), 3 |u|, 4 Mach, 5 |curl u|, 6 div u, 7 Q.  The field covers the handle's nzl planes
 * (nx*ny*nzl floats, reference layout, no halo).  A single-domain handle refreshes its periodic halo
 * itself; a slab handle needs current halo planes (one exchange) before the call.
 *   tau3d_vis        computes the field into the handle's buffer and, if host_out != NULL, copies it out
 *   tau3d_vis_async  same without the copy; out_dev != NULL writes to caller's device memory instead
 *   tau3d_slice_rgba pixels of local plane `zslice` (clamped) of the LAST field: grey = t, alpha =
 *                    clamp(a_gain t^2), t normalised by that slice's own min/max, 0xAABBGGRR words
 *   tau3d_outflow_reflection  max |p - p_inflow| over the last nprobe x-columns of the slab */
int tau3d_vis(tau3d_t *h, int mode, float *host_out);
int tau3d_vis_async(tau3d_t *h, int mode, float *out_dev);
int tau3d_slice_rgba(tau3d_t *h, int zslice, int log_scale, float a_gain, uint32_t *host_rgba, float *mn, float *mx);
int tau3d_outflow_reflection(tau3d_t *h, int nprobe, float *max_dp);
/* Per-launch timing of k_step with HIP events on the launch stream (for bench.py's roofline
 * figure).  enable(1) starts collecting (up to 4096 launches), read() synchronises and returns
 * the summed duration in ms, the number of launches and the cells they updated. */
int tau3d_timing_enable(tau3d_t *h, int on);
int tau3d_timing_read(tau3d_t *h, double *total_ms, int *launches, double *cells);
/* device time from the start of the first timed interval to the end of the last (kernels and the gaps between them) */
int tau3d_timing_span(tau3d_t *h, double *span_ms);
/* the same intervals split at the point between the two kernels of the split step (single-domain steps only):
 * summed duration of k_flux_xy and of k_update_z, and the number of intervals that had such a point */
int tau3d_timing_read_split(tau3d_t *h, double *xy_ms, double *z_ms, int *intervals);

/* =====================================================================
 * 2D compressible Euler, GPU scheme — replaces the six launches per step of
 * tau_hypersonic_cuda.cu:1833-1886 (and run_hypersonic_steps, tau_hypersonic_cuda_tests.cu:178-243)
 * with one fused fp32 kernel.  State: rho, mx, my, E as four row-major y*W+x fp32 arrays
 * (the reference keeps them fp64; BASELINE.json asks for fp32) + a u8 body mask.
 * ===================================================================== */
typedef struct tauh2 tauh2_t;
void tauh2_params_default(tauh2_params *p, int W, int H);                   /* default_config, :1394-1409 */
int tauh2_create(tauh2_t **out, const tauh2_params *p, int device, void *stream);
void tauh2_destroy(tauh2_t *h);
int tauh2_init(tauh2_t *h);                                                 /* k_init, :740-770 */
int tauh2_upload(tauh2_t *h, const float *const host[4], const uint8_t *mask /* may be NULL */);
int tauh2_download(tauh2_t *h, float *const host[4], uint8_t *mask /* may be NULL */);
int tauh2_state_ptrs(tauh2_t *h, float *dptr[4], uint8_t **mask);
/* n steps of the loop body :1833-1888 (inflow column, CFL/diffusion dt, predict, fluxes, update,
 * swap); dt control stays on the device.  *t_out = accumulated sim_t. */
int tauh2_step(tauh2_t *h, int nsteps, double *t_out);
int tauh2_step_async(tauh2_t *h, int nsteps);
/* one step with a caller-chosen dt (parity tests) */
int tauh2_step_explicit(tauh2_t *h, double dt);
/* Rendering — replaces k_render_vals, k_reduce_minmax, k_compute_inv_range and k_render_pixels
 * (tau_hypersonic_cuda.cu:1178-1334; frame loop :1871-1888).  view_mode as the reference's keys 1-7:
 * 0 log rho, 1 log p, 2 speed, 3 log |grad rho|, 4 asinh(vorticity), 5 Mach, 6 log(p/rho).
 * host_rgba: W*H words, bytes R,G,B,255 in memory order (uchar4), body cells grey 110; host_vals: the
 * scalar per cell (0 in the body); vmin/vmax: its range over the fluid.  Any output may be NULL. */
int tauh2_render(tauh2_t *h, int view_mode, uint32_t *host_rgba, float *host_vals, double *vmin, double *vmax);
/* test seam (the reference's is the NO_MAIN/NO_RAYLIB include boundary, tau_hypersonic_cuda.cu:16-18):
 * evaluates the kernel's device helpers on the known answers of tau_hypersonic_cuda_tests.cu:245-346;
 * out[48] layout is documented at h2d::k_unit */
int tauh2_unit_eval(tauh2_t *h, float out[48]);
/* The hand-built-field neighbour lookups of tau_hypersonic_cuda_tests.cu:348-371 / 567-640 evaluated by the engine's own
 * staging rule on the handle's CURRENT state (upload the test field first), centre cell (x, y):
 * out = left.rho, left.mx, right.rho, right.mx, up.mx | left.rho, left.mx, wall(x,y+1).mx, top_clamped(x,H+20).rho */
int tauh2_unit_neighbors(tauh2_t *h, int x, int y, float out[9]);
double tauh2_body_sdf(double x, double y, double Rb, double Rn, double theta); /* sdSphereConeCapsule, :644-686 */
int tauh2_get_time(tauh2_t *h, double *t, double *dt_last, double *maxs, int *step);
int tauh2_sync(tauh2_t *h);

/* =====================================================================
 * 2D WCSPH — replaces the per-sub-step launches of tau_sph.cu:676-701 (clear heads, build
 * cells, density/pressure, forces, integrate) and the host dt / log-time loop :665-721.
 * State in the reference layout and particle order: pos, vel, acc (float2 AoS), s = ln rho, press.
 * XSPH (:274-322) and rain (:377-392) run inside the sub-step when tausph_params.useXSPH / .rain are set.
 * ===================================================================== */
typedef struct tausph tausph_t;
void tausph_params_default(tausph_params *p, int N);                        /* :49-85 */
int tausph_create(tausph_t **out, const tausph_params *p, int device, void *stream);
void tausph_destroy(tausph_t *h);
int tausph_reset_particles(tausph_t *h);                                    /* :493-510 + H2D :567-570 */
int tausph_upload(tausph_t *h, const float *pos_xy, const float *vel_xy);
/* any pointer may be NULL; cellOf = integer cell index gy*Gx+gx of the last sub-step's build */
int tausph_download(tausph_t *h, float *pos_xy, float *vel_xy, float *acc_xy, float *s, float *press, int32_t *cellOf);
int tausph_state_ptrs(tausph_t *h, float **pos, float **vel, float **acc, float **s, float **press);
/* Once the position pointer has been handed out the handle counts the cells from the positions at EVERY sub-step (as the
 * reference's k_build_cells does), instead of trusting the count its force pass made: a caller may move particles through it at
 * any time.  tausph_state_written is still how a caller says so for a sub-step that is already enqueued. */
int tausph_state_written(tausph_t *h);
int tausph_grid(tausph_t *h, int *Gx, int *Gy, float *cell, float *hh, float *mass); /* :512-521, 573-576 */
float tausph_dt(tausph_t *h);                                               /* :666-669 */
int tausph_substep_async(tausph_t *h, float dt);                            /* one pass of :676-701 */
int tausph_step(tausph_t *h, int nsteps);                                   /* :665-721 */
int tausph_step_async(tausph_t *h, int nsteps);
int tausph_get_clock(tausph_t *h, float *t, float *tau, int64_t *step);
int tausph_sync(tausph_t *h);
/* XSPH (k_xsph_cell + k_apply_xsph, :274-322) and rain (k_rain :377-392 + host bookkeeping :706-716) run inside
 * tausph_substep_async when tausph_params.useXSPH / .rain are set at create.  Rain collisions (two drops on one
 * particle) are resolved deterministically: the highest drop index wins. */
int tausph_rasterize(tausph_t *h, int W, int H, int32_t *host_grid2);        /* k_clear_grid + k_rasterize, :357-374: W x 2H counts */
int64_t tausph_rain_spawned(tausph_t *h);                                    /* drops spawned so far */
/* diagnostic: ordered pairs (i, j), i != j, closer than the 2h support among the records of the LAST cell build — the
 * pair interactions each neighbour pass of that sub-step evaluated (needs at least one sub-step) */
int tausph_count_pairs(tausph_t *h, int64_t *ordered_pairs);

/* =====================================================================
 * Gray-Scott — replaces step_kernel launch + swap, tau_gray_scott.cu:321-329
 * ===================================================================== */
typedef struct taugs taugs_t;
void taugs_params_default(taugs_params *p, int nx, int ny);               /* :43-61 */
int taugs_create(taugs_t **out, const taugs_params *p, int device, void *stream);
void taugs_destroy(taugs_t *h);
int taugs_init_pattern(taugs_t *h, uint32_t seed);                         /* :173-204 + H2D :308-309 */
int taugs_upload(taugs_t *h, const float *u, const float *v);
int taugs_download(taugs_t *h, float *u, float *v);
int taugs_state_ptrs(taugs_t *h, float **u, float **v);
int taugs_info(taugs_t *h, int *nx, int *ny, int *device, void **stream);   /* any pointer may be NULL */
/* init_pattern (tau_gray_scott.cu:173-204) into host arrays of nx * ny floats (what taugs_init_pattern uploads): a row-slab
 * rank cuts its rows, halo rows included, out of the global pattern */
int taugs_pattern_host(int nx, int ny, uint32_t seed, float *u, float *v);
int taugs_step(taugs_t *h, int nsteps);                                    /* :321-329 */
int taugs_step_async(taugs_t *h, int nsteps);
/* time levels per launch: 0 = default (four levels fused per pass, the remainder as single steps), 1 = one launch per
 * step (the reference's structure, :321-329), 2..4.  The results are bit-identical either way. */
int taugs_set_levels(taugs_t *h, int levels);
int taugs_sync(taugs_t *h);

/* =====================================================================
 * 5-point Laplacian viscosity passes (periodic), race-free ping-pong form of
 * tau_burgers.cu:490-525 (kind 0, asinh-encoded fields) and
 * tau_shallow_water.cu:516-547 (kind 1, plain u, v).
 * ===================================================================== */
typedef struct taulap taulap_t;
int taulap_create(taulap_t **out, const taulap_params *p, int kind, int oneD, int device, void *stream);
void taulap_destroy(taulap_t *h);
int taulap_upload(taulap_t *h, const float *a, const float *b);
int taulap_download(taulap_t *h, float *a, float *b);
int taulap_state_ptrs(taulap_t *h, float **a, float **b);
int taulap_info(taulap_t *h, int *nx, int *ny, int *device, void **stream);
int taulap_set_dt(taulap_t *h, float dt);
int taulap_step(taulap_t *h, int npasses);
int taulap_step_async(taulap_t *h, int npasses);
int taulap_sync(taulap_t *h);

/* ---- row-slab ring of the two periodic 5-point-stencil handles above (csrc/ring.hip; SURVEY §8e, no reference counterpart: the
 * reference is single-GPU, the loop sharded is tau_gray_scott.cu:321-329 / the viscosity calls of tau_burgers.cu:490-525 and
 * tau_shallow_water.cu:516-547).  Rank r of `world` owns rows [y0, y0 + nyl) (taurow_bounds) and creates ITS handle with
 * ny = nyl + 2 * halo rows; the ring steps it `halo` time levels at a time and refreshes the halo rows from the ring neighbours in
 * between — owned rows are bit-identical to the single-domain run (a 5-point stencil carries the wrap-around error one row per
 * step).  transport: TAU3D_RING_RCCL (one device per rank), TAU3D_RING_HOST (ranks may share a device), TAU3D_RING_LOCAL
 * (world 1).  rendezvous / job_key as for tau3d_ring_create.  Upload the local rows INCLUDING halo rows (or upload anything and
 * call *_ring_exchange_async once: the halos are then current). */
typedef struct taurow_ring taurow_ring_t;
int taurow_bounds(int ny, int world, int rank, int *y0, int *nyl);
int taugs_ring_create(taurow_ring_t **out, taugs_t *h, int halo, int rank, int world, int transport, const char *rendezvous, uint64_t job_key);
int taugs_ring_step_async(taurow_ring_t *r, int nsteps);
int taugs_ring_exchange_async(taurow_ring_t *r);
int taugs_ring_finish(taurow_ring_t *r);
int taugs_ring_barrier(taurow_ring_t *r);
void taugs_ring_destroy(taurow_ring_t *r);
int taulap_ring_create(taurow_ring_t **out, taulap_t *h, int halo, int rank, int world, int transport, const char *rendezvous, uint64_t job_key);
int taulap_ring_step_async(taurow_ring_t *r, int npasses);
int taulap_ring_exchange_async(taurow_ring_t *r);
int taulap_ring_finish(taurow_ring_t *r);
int taulap_ring_barrier(taurow_ring_t *r);
void taulap_ring_destroy(taurow_ring_t *r);
/* the same ring under its own name (one implementation) */
int taurow_ring_step_async(taurow_ring_t *r, int nsteps);
int taurow_ring_exchange_async(taurow_ring_t *r);
int taurow_ring_finish(taurow_ring_t *r);
int taurow_ring_barrier(taurow_ring_t *r);
void taurow_ring_destroy(taurow_ring_t *r);
int taurow_ring_info(taurow_ring_t *r, int *nyl, int *halo, long *exchanges, int *rccl_version, int *comm_ranks);

/* =====================================================================
 * Full Burgers (kind 0) and shallow-water (kind 1) programs — replace do_step of
 * tau_burgers.cu:677-718 and tau_shallow_water.cu:671-705 (wavespeed reduction + host max, flux
 * kernels, update, viscosity) with one fused kernel per step; the log-time clock (t *= exp(dtau),
 * :798-799) is advanced by the library.  State: Burgers phi_u, phi_v; shallow water sigma = ln h, u, v.
 * ===================================================================== */
typedef struct tauflow tauflow_t;
void tauflow_params_default(tauflow_params *p, int kind, int nx, int ny);
int tauflow_create(tauflow_t **out, const tauflow_params *p, int kind, int device, void *stream);
void tauflow_destroy(tauflow_t *h);
int tauflow_init(tauflow_t *h);                              /* initialize_host + H2D */
int tauflow_upload(tauflow_t *h, const float *const f[3]);
int tauflow_download(tauflow_t *h, float *const f[3]);
int tauflow_state_ptrs(tauflow_t *h, float *f[3]);
int tauflow_step(tauflow_t *h, int nsteps);
int tauflow_step_async(tauflow_t *h, int nsteps);
int tauflow_step_explicit(tauflow_t *h, float dt);          /* one do_step with a caller-chosen dt_eff */
int tauflow_get_clock(tauflow_t *h, float *t, float *tau, float *dt_last, float *wavespeed, int64_t *step);
int tauflow_colehopf_relL2(tauflow_t *h, float t_now, double *rel);   /* tau_burgers.cu:720-736 */
int tauflow_sync(tauflow_t *h);
/* device time (ms) of what is enqueued between the two calls, from events on the handle's stream: the "GPU" / "GPU only" line of
 * the reference's headless summaries (cudaEvent pairs, tau_burgers.cu:790-820, tau_shallow_water.cu:751-782) */
int tauflow_timer_start(tauflow_t *h);
int tauflow_timer_stop(tauflow_t *h, double *ms);

/* =====================================================================
 * D2Q9 BGK lattice Boltzmann — replaces the launches of tau_lbm.cu:245-247 (init) and the loop body
 * :262-269 (collide_stream_kernel, pointer swap, render_kernel).  State: nine populations as nine
 * row-major j*nx+i fp32 planes back to back (fidx, :62-64) + a u8 solid mask; ping-pong inside the handle.
 * ===================================================================== */
typedef struct taulbm taulbm_t;
void taulbm_params_default(taulbm_params *p);                               /* Params, :43-55 */
int taulbm_create(taulbm_t **out, const taulbm_params *p, int device, void *stream);
void taulbm_destroy(taulbm_t *h);
int taulbm_init(taulbm_t *h);                                               /* init_kernel :72-92 + D2D copy :247 */
int taulbm_upload(taulbm_t *h, const float *f9 /* 9*nx*ny or NULL */, const uint8_t *solid /* or NULL */);
int taulbm_download(taulbm_t *h, float *f9, uint8_t *solid);                /* either may be NULL */
int taulbm_state_ptrs(taulbm_t *h, float **f9, uint8_t **solid);
int taulbm_set_drive(taulbm_t *h, float drive);                             /* keys '+' / '-', :283-284 */
int taulbm_step(taulbm_t *h, int nsteps);                                   /* :262-265 */
int taulbm_step_async(taulbm_t *h, int nsteps);
int taulbm_speed(taulbm_t *h, float *host_speed);                           /* render_kernel :134-153: |u|, -1 in solids */
int64_t taulbm_steps_done(taulbm_t *h);
int taulbm_sync(taulbm_t *h);

#ifdef __cplusplus
}
#endif
#endif /* TAUENG_H */
