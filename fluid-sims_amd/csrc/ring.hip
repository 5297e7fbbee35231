// ring.hip — the Z-slab ring of the 3D hypersonic grid inside the library: one process per GPU, halo exchange and the
// max all-reduce issued from plain C (the reference is single-GPU, its loop is tau_hypersonic_3d_cuda.cu:1678-1713).
//
// A ring owns nothing of the simulation: it drives ONE tau3d slab handle through the public pieces of include/taueng.h
// (tau3d_slab_begin/edges/interior/end_async, the packed halo buffers, the two max words) and adds the communication:
//
//   compute stream S (the handle's)              exchange stream X (the ring's)
//   wait evX(n-1)   halos + max of step n-1 landed
//   slab_begin      controller(n-1), clock(n), unpack received halos
//   slab_edges(E)   planes [0,E) + [nzl-E,nzl), new boundary planes -> packed send buffers
//   record evE  ------------------------------>  wait evE
//   slab_interior(E)   overlaps                  ncclGroupStart; Send x2, Recv x2; ncclGroupEnd     (18.9 MB per direction at 512^2)
//   record evI  ------------------------------>  wait evI
//                                                ncclAllReduce(max) on the two words of tau3d_max_ptr, in place
//                                                record evX(n)
//   slab_end        swap (host bookkeeping)
//
// No host synchronisation between the pieces: tau3d_ring_step_async(n) only enqueues.  One communicator, used on ONE stream
// (X), so RCCL sees its operations in one order on every rank.
//
// Transports:
//   TAU3D_RING_RCCL   ncclSend / ncclRecv / ncclAllReduce over xGMI.  librccl is bound at run time (dlopen) — the copy the
//                     process already holds (PyTorch's, under Python) or the one beside the HIP runtime in use — so a
//                     single-GPU user of libtaueng never loads it.  world == 1 sends to / receives from itself.
//   TAU3D_RING_HOST   host-staged through the shared rendezvous file: D2H, process barrier, H2D.  Synchronous and slow; it
//                     exists so that N ranks SHARING one device (RCCL refuses duplicate GPUs) can run the ring's ordering
//                     — the multi-process tests on a one-GPU box — and as a fallback where RCCL is absent.
//   TAU3D_RING_LOCAL  world == 1 only: two device copies (periodic self-neighbour), no collective.
//
// Rendezvous: a small file in /dev/shm (or anywhere mmap-able) that rank 0 creates and the others map.  It carries the
// ncclUniqueId, a barrier with a timeout, one status word per rank (a rank that cannot proceed says so instead of leaving
// the others inside ncclCommInitRank for ever), and the staging area of the host transport.
#include "../../include/taueng.h"
#include "tau_common.h"

#include <rccl/rccl.h>   // types and prototypes only: every call goes through the table below

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <dlfcn.h>
#include <fcntl.h>
#include <new>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace ring {

constexpr int MAX_WORLD = 64;
constexpr uint64_t MAGIC = 0x7461753364726e67ull;   // "tau3drng"

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  char path[512] = "";
};

static Rccl g_rccl;

static void *try_open(const char *name, int flags, char *where, size_t nwhere) {
  void *l = dlopen(name, flags);
  if (l && where) snprintf(where, nwhere, "%s", name);
  return l;
}

// Bind librccl: an explicit TAU_RCCL_LIB, else the copy already in the process, else the one beside the HIP runtime this
// process runs on (so RCCL and the engine share ONE libamdhip64), else the loader's search path.
static int rccl_load() {
  Rccl &R = g_rccl;
  if (R.lib) return 0;
  void *l = nullptr;
  if (const char *e = getenv("TAU_RCCL_LIB")) l = try_open(e, RTLD_NOW | RTLD_GLOBAL, R.path, sizeof R.path);
  if (!l) l = try_open("librccl.so.1", RTLD_NOW | RTLD_NOLOAD, R.path, sizeof R.path);
  if (!l) l = try_open("librccl.so", RTLD_NOW | RTLD_NOLOAD, R.path, sizeof R.path);
  if (!l) {
    Dl_info di;
    if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
      char dir[400];
      snprintf(dir, sizeof dir, "%s", di.dli_fname);
      if (char *s = strrchr(dir, '/')) {
        *s = 0;
        char cand[512];
        for (const char *n : {"librccl.so.1", "librccl.so"}) {
          snprintf(cand, sizeof cand, "%s/%s", dir, n);
          if (!l && access(cand, R_OK) == 0) l = try_open(cand, RTLD_NOW | RTLD_GLOBAL, R.path, sizeof R.path);
        }
      }
    }
  }
  if (!l) l = try_open("librccl.so.1", RTLD_NOW | RTLD_GLOBAL, R.path, sizeof R.path);
  if (!l) l = try_open("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL, R.path, sizeof R.path);
  if (!l) return tau::fail("tau3d_ring: librccl could not be loaded (%s); set TAU_RCCL_LIB", dlerror());
#define SYM(field, name)                                                                  \
  do {                                                                                    \
    *(void **)(&R.field) = dlsym(l, name);                                                \
    if (!R.field) return tau::fail("tau3d_ring: %s has no symbol %s", R.path, name);      \
  } while (0)
  SYM(GetVersion, "ncclGetVersion"); SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(GetErrorString, "ncclGetErrorString");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce");
#undef SYM
  R.lib = l;
  return 0;
}

#define TAU_NCCL(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess)                                                                               \
      return ::tau::fail("%s: %s (%s:%d)", #expr, ring::g_rccl.GetErrorString(r_), __FILE__, __LINE__);  \
  } while (0)

// ---- the rendezvous file
struct Shared {
  std::atomic<uint64_t> ready;      // MAGIC ^ job_key once rank 0 has filled the header
  int32_t world, transport;
  uint64_t slot_bytes;              // host transport: bytes of ONE packed buffer (a side); 0 otherwise
  std::atomic<int32_t> bar_count, bar_gen;
  std::atomic<int32_t> status[MAX_WORLD];   // 0 unknown, 1 ready, 2 failed
  float maxw[MAX_WORLD][2];
  ncclUniqueId id;
  // then: world x 2 x slot_bytes of staging (host transport)
};
static size_t shared_bytes(int world, size_t slot) { return ((sizeof(Shared) + 4095) & ~(size_t)4095) + (size_t)world * 2 * slot; }
static char *slot_ptr(Shared *sh, int rank, int side) {
  return (char *)sh + ((sizeof(Shared) + 4095) & ~(size_t)4095) + ((size_t)rank * 2 + side) * sh->slot_bytes;
}
static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static double timeout_s() {
  static const double v = [] { const char *e = getenv("TAU3D_RING_TIMEOUT"); double t = e ? atof(e) : 0.0; return t > 0.0 ? t : 120.0; }();
  return v;
}
static void nap(int &spins) {
  if (++spins < 200) sched_yield();
  else { timespec ts = {0, 200000}; nanosleep(&ts, nullptr); }
}
// sense-counting barrier over the mapped file; returns non-zero on timeout or when any rank reported failure
static int barrier(Shared *sh, const char *what) {
  const int gen = sh->bar_gen.load(std::memory_order_acquire);
  if (sh->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == sh->world) {
    sh->bar_count.store(0, std::memory_order_relaxed);
    sh->bar_gen.store(gen + 1, std::memory_order_release);
    return 0;
  }
  const double t0 = now_s();
  int spins = 0;
  while (sh->bar_gen.load(std::memory_order_acquire) == gen) {
    nap(spins);
    if ((spins & 1023) == 0) {
      for (int r = 0; r < sh->world; r++)
        if (sh->status[r].load(std::memory_order_acquire) == 2) return tau::fail("tau3d_ring: rank %d failed (%s)", r, what);
      if (now_s() - t0 > timeout_s()) return tau::fail("tau3d_ring: barrier timed out after %.0f s (%s)", timeout_s(), what);
    }
  }
  return 0;
}

} // namespace ring

struct tau3d_ring {
  tau3d_t *h = nullptr;
  int rank = 0, world = 1, transport = 0, lo = 0, hi = 0;
  int nzl = 0, edge = 3, device = 0;
  hipStream_t S = nullptr, X = nullptr;
  hipEvent_t evE = nullptr, evI = nullptr, evX = nullptr;
  ncclComm_t comm = nullptr;
  ring::Shared *sh = nullptr;
  size_t sh_bytes = 0;
  char path[256] = "";
  float *buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [kind: 0 send, 1 recv][side]
  size_t nfloats = 0;
  float *maxw = nullptr;
  bool primed = false;
  long steps = 0;
};

static int ring_map(tau3d_ring *r, const char *path, uint64_t key, size_t slot) {
  using namespace ring;
  const size_t bytes = shared_bytes(r->world, slot);
  snprintf(r->path, sizeof r->path, "%s", path);
  int fd = -1;
  if (r->rank == 0) {
    unlink(path);   // a stale file of an earlier job: rank 0 always starts fresh
    fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) return tau::fail("tau3d_ring: cannot create %s: %s", path, strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return tau::fail("tau3d_ring: ftruncate(%s, %zu): %s", path, bytes, strerror(errno)); }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return tau::fail("tau3d_ring: mmap(%s): %s", path, strerror(errno));
    Shared *sh = new (m) Shared();   // ftruncate zero-filled it; the atomics start at 0
    sh->world = r->world; sh->transport = r->transport; sh->slot_bytes = slot;
    r->sh = sh; r->sh_bytes = bytes;
    return 0;   // `ready` is published by the caller once the id is in
  }
  const double t0 = now_s();
  int spins = 0;
  for (;;) {   // wait for THIS job's file: an older one under the same name carries another key
    fd = open(path, O_RDWR);
    if (fd >= 0) {
      struct stat st;
      if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) {
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
          Shared *sh = (Shared *)m;
          const double t1 = now_s();
          int sp2 = 0;
          while (sh->ready.load(std::memory_order_acquire) == 0 && now_s() - t1 < 1.0) nap(sp2);   // rank 0 may still be filling it
          if (sh->ready.load(std::memory_order_acquire) == (MAGIC ^ key)) {
            close(fd);
            if (sh->world != r->world || sh->transport != r->transport || sh->slot_bytes != slot) {
              munmap(m, bytes);
              return tau::fail("tau3d_ring: %s was made for world %d / transport %d / %llu-byte slots, this rank wants %d / %d / %zu",
                               path, sh->world, sh->transport, (unsigned long long)sh->slot_bytes, r->world, r->transport, slot);
            }
            r->sh = sh; r->sh_bytes = bytes;
            return 0;
          }
          munmap(m, bytes);
        }
      }
      close(fd);
    }
    if (now_s() - t0 > timeout_s()) return tau::fail("tau3d_ring: rank %d waited %.0f s for rank 0's rendezvous file %s", r->rank, timeout_s(), path);
    nap(spins);
  }
}

extern "C" int tau3d_slab_bounds(int nz, int world, int rank, int *z0, int *nzl) {
  if (world < 1 || rank < 0 || rank >= world || !z0 || !nzl) return tau::fail("tau3d_slab_bounds: bad argument");
  const int base = nz / world, rem = nz % world;
  *z0 = rank * base + (rank < rem ? rank : rem);
  *nzl = base + (rank < rem ? 1 : 0);
  if (*nzl < 6) return tau::fail("tau3d_slab_bounds: nz=%d over %d ranks leaves a %d-plane slab; need >= 6", nz, world, *nzl);
  return 0;
}

extern "C" void tau3d_ring_destroy(tau3d_ring_t *r) {
  if (!r) return;
  hipSetDevice(r->device);
  if (r->X) hipStreamSynchronize(r->X);
  if (r->S) hipStreamSynchronize(r->S);
  if (r->comm) ring::g_rccl.CommDestroy(r->comm);
  if (r->evE) hipEventDestroy(r->evE);
  if (r->evI) hipEventDestroy(r->evI);
  if (r->evX) hipEventDestroy(r->evX);
  if (r->X) hipStreamDestroy(r->X);
  if (r->sh) {
    munmap(r->sh, r->sh_bytes);
    if (r->rank == 0 && r->path[0]) unlink(r->path);   // the mappings of the other ranks keep the pages alive
  }
  delete r;
}

extern "C" int tau3d_ring_create(tau3d_ring_t **out, tau3d_t *h, int rank, int world, int transport, const char *rendezvous,
                                 uint64_t job_key) {
  using namespace ring;
  if (!out || !h) return tau::fail("tau3d_ring_create: null argument");
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return tau::fail("tau3d_ring_create: rank %d of %d", rank, world);
  if (transport < TAU3D_RING_RCCL || transport > TAU3D_RING_LOCAL) return tau::fail("tau3d_ring_create: unknown transport %d", transport);
  if (transport == TAU3D_RING_LOCAL && world != 1) return tau::fail("tau3d_ring_create: the local transport is for world 1");
  if (world > 1 && (!rendezvous || !rendezvous[0])) return tau::fail("tau3d_ring_create: world %d needs a rendezvous path", world);
  int z0 = 0, nzl = 0, nz = 0, device = 0;
  void *stream = nullptr;
  if (tau3d_slab_info(h, &z0, &nzl, &nz, &device, &stream)) return 1;
  {
    int ez0, enzl;
    if (tau3d_slab_bounds(nz, world, rank, &ez0, &enzl)) return 1;
    if (ez0 != z0 || enzl != nzl)
      return tau::fail("tau3d_ring_create: rank %d of %d owns planes [%d,%d) of nz=%d, the handle was created for [%d,%d)", rank, world,
                       ez0, ez0 + enzl, nz, z0, z0 + nzl);
  }
  tau3d_ring *r = new (std::nothrow) tau3d_ring();
  if (!r) return tau::fail("tau3d_ring_create: out of host memory");
  tau::HandleGuard<tau3d_ring> guard{r, tau3d_ring_destroy};
  r->h = h; r->rank = rank; r->world = world; r->transport = transport; r->nzl = nzl; r->device = device;
  r->lo = (rank + world - 1) % world; r->hi = (rank + 1) % world;
  r->edge = nzl / 2 < 8 ? (nzl / 2 < 3 ? 3 : nzl / 2) : 8;   // planes per edge launch: >= the 3 halo planes, <= half a slab, 8 where they fit
  if (const char *e = getenv("TAU3D_RING_EDGE")) { const int v = atoi(e); if (v >= 3) r->edge = v; }
  r->S = (hipStream_t)stream;
  TAU_HIP(hipSetDevice(device));
  TAU_HIP(hipStreamCreateWithFlags(&r->X, hipStreamNonBlocking));
  TAU_HIP(hipEventCreateWithFlags(&r->evE, hipEventDisableTiming));
  TAU_HIP(hipEventCreateWithFlags(&r->evI, hipEventDisableTiming));
  TAU_HIP(hipEventCreateWithFlags(&r->evX, hipEventDisableTiming));
  for (int k = 0; k < 2; k++)
    for (int s = 0; s < 2; s++)
      if (tau3d_halo_buf_ptr(h, k, s, &r->buf[k][s], &r->nfloats)) return 1;
  if (tau3d_max_ptr(h, &r->maxw)) return 1;

  const bool need_file = world > 1 || (transport == TAU3D_RING_HOST && rendezvous && rendezvous[0]);
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  int my_status = 1;
  if (transport == TAU3D_RING_RCCL) {
    if (rccl_load()) my_status = 2;
    else if (rank == 0 && g_rccl.GetUniqueId(&id) != ncclSuccess) { tau::fail("tau3d_ring_create: ncclGetUniqueId failed"); my_status = 2; }
  }
  if (need_file) {
    const size_t slot = transport == TAU3D_RING_HOST ? r->nfloats * sizeof(float) : 0;
    if (ring_map(r, rendezvous, job_key, slot)) return 1;
    if (rank == 0) {
      r->sh->id = id;
      r->sh->ready.store(MAGIC ^ job_key, std::memory_order_release);
    }
    // every rank says whether it can go on BEFORE anyone enters ncclCommInitRank (which waits for all ranks for ever)
    r->sh->status[rank].store(my_status, std::memory_order_release);
    if (my_status == 2) return 1;
    if (barrier(r->sh, "create")) return 1;
    for (int k = 0; k < world; k++)
      if (r->sh->status[k].load(std::memory_order_acquire) != 1) return tau::fail("tau3d_ring_create: rank %d could not start", k);
    id = r->sh->id;
  } else if (my_status == 2) return 1;

  if (transport == TAU3D_RING_RCCL) {
    // RCCL refuses two ranks on one device with an error deep inside the init; say it here, in the caller's terms
    int ndev = 0;
    TAU_HIP(hipGetDeviceCount(&ndev));
    if (world > ndev && !getenv("TAU3D_RING_NO_DEVICE_CHECK"))
      return tau::fail("tau3d_ring_create: the RCCL transport needs %d devices (one per rank), this node shows %d", world, ndev);
    TAU_NCCL(g_rccl.CommInitRank(&r->comm, world, id, rank));
  }
  *out = guard.release();
  return 0;
}

extern "C" int tau3d_ring_info(tau3d_ring_t *r, int *rccl_version, int *comm_ranks, int *edge_planes, char *lib_path, size_t lib_path_len) {
  if (!r) return tau::fail("tau3d_ring_info: null ring");
  if (rccl_version) *rccl_version = 0;
  if (comm_ranks) *comm_ranks = r->transport == TAU3D_RING_RCCL ? 0 : r->world;
  if (edge_planes) *edge_planes = r->edge;
  if (lib_path && lib_path_len) lib_path[0] = 0;
  if (r->transport == TAU3D_RING_RCCL) {
    if (rccl_version) TAU_NCCL(ring::g_rccl.GetVersion(rccl_version));
    if (comm_ranks) TAU_NCCL(ring::g_rccl.CommCount(r->comm, comm_ranks));
    if (lib_path && lib_path_len) snprintf(lib_path, lib_path_len, "%s", ring::g_rccl.path);
  }
  return 0;
}

// halos of step n: my low boundary planes are the low neighbour's HIGH halo, my high planes the high neighbour's LOW halo.
// Between one pair of ranks RCCL matches sends and receives in issue order; with world == 2 both neighbours are the same
// peer, so sends go (side 0, side 1) and receives (side 1, side 0): the peer's first send (its low planes) is my high halo.
static int exchange_rccl(tau3d_ring *r) {
  using namespace ring;
  const size_t n = r->nfloats;
  TAU_NCCL(g_rccl.GroupStart());
  TAU_NCCL(g_rccl.Send(r->buf[0][0], n, ncclFloat, r->lo, r->comm, r->X));
  TAU_NCCL(g_rccl.Send(r->buf[0][1], n, ncclFloat, r->hi, r->comm, r->X));
  TAU_NCCL(g_rccl.Recv(r->buf[1][1], n, ncclFloat, r->hi, r->comm, r->X));
  TAU_NCCL(g_rccl.Recv(r->buf[1][0], n, ncclFloat, r->lo, r->comm, r->X));
  TAU_NCCL(g_rccl.GroupEnd());
  return 0;
}
static int exchange_local(tau3d_ring *r) {
  const size_t b = r->nfloats * sizeof(float);
  TAU_HIP(hipMemcpyAsync(r->buf[1][1], r->buf[0][0], b, hipMemcpyDeviceToDevice, r->X));
  TAU_HIP(hipMemcpyAsync(r->buf[1][0], r->buf[0][1], b, hipMemcpyDeviceToDevice, r->X));
  return 0;
}
// host transport: everything on X, synchronously.  Without a rendezvous file (world 1) it degenerates to the local copy.
static int exchange_host(tau3d_ring *r) {
  using namespace ring;
  if (!r->sh) { if (exchange_local(r)) return 1; TAU_HIP(hipStreamSynchronize(r->X)); return 0; }
  const size_t b = r->nfloats * sizeof(float);
  TAU_HIP(hipMemcpyAsync(slot_ptr(r->sh, r->rank, 0), r->buf[0][0], b, hipMemcpyDeviceToHost, r->X));
  TAU_HIP(hipMemcpyAsync(slot_ptr(r->sh, r->rank, 1), r->buf[0][1], b, hipMemcpyDeviceToHost, r->X));
  TAU_HIP(hipStreamSynchronize(r->X));
  if (barrier(r->sh, "halo slots written")) return 1;
  TAU_HIP(hipMemcpyAsync(r->buf[1][1], slot_ptr(r->sh, r->hi, 0), b, hipMemcpyHostToDevice, r->X));
  TAU_HIP(hipMemcpyAsync(r->buf[1][0], slot_ptr(r->sh, r->lo, 1), b, hipMemcpyHostToDevice, r->X));
  TAU_HIP(hipStreamSynchronize(r->X));
  return barrier(r->sh, "halo slots read");   // nobody overwrites a slot a neighbour is still reading
}
static int allreduce_host(tau3d_ring *r) {
  using namespace ring;
  if (!r->sh) return 0;
  float w[2];
  TAU_HIP(hipMemcpyAsync(w, r->maxw, sizeof w, hipMemcpyDeviceToHost, r->X));
  TAU_HIP(hipStreamSynchronize(r->X));
  r->sh->maxw[r->rank][0] = w[0]; r->sh->maxw[r->rank][1] = w[1];
  if (barrier(r->sh, "max words written")) return 1;
  for (int k = 0; k < r->world; k++) {
    const float a = r->sh->maxw[k][0], c = r->sh->maxw[k][1];
    if (a > w[0]) w[0] = a;
    if (c > w[1]) w[1] = c;
  }
  TAU_HIP(hipMemcpyAsync(r->maxw, w, sizeof w, hipMemcpyHostToDevice, r->X));
  TAU_HIP(hipStreamSynchronize(r->X));
  return barrier(r->sh, "max words read");
}

// the communication of one step (or of the priming exchange): X waits for `after_send` (send buffers written), exchanges,
// waits for `after_max` (all launches that raise the max words), reduces them, and records evX
static int communicate(tau3d_ring *r, bool with_max) {
  using namespace ring;
  TAU_HIP(hipStreamWaitEvent(r->X, r->evE, 0));
  switch (r->transport) {
    case TAU3D_RING_RCCL: if (exchange_rccl(r)) return 1; break;
    case TAU3D_RING_HOST: if (exchange_host(r)) return 1; break;
    default: if (exchange_local(r)) return 1; break;
  }
  if (with_max && r->transport != TAU3D_RING_LOCAL) {
    TAU_HIP(hipStreamWaitEvent(r->X, r->evI, 0));
    if (r->transport == TAU3D_RING_RCCL) TAU_NCCL(g_rccl.AllReduce(r->maxw, r->maxw, 2, ncclFloat, ncclMax, r->comm, r->X));
    else if (allreduce_host(r)) return 1;
  }
  TAU_HIP(hipEventRecord(r->evX, r->X));
  return 0;
}

/* exchange the halos of the CURRENT state and agree on its field range (after init / upload): without it the first step
 * would read undefined halo planes and every slab could pick its own WENO weight form */
extern "C" int tau3d_ring_prime(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_prime: null ring");
  TAU_HIP(hipSetDevice(r->device));
  if (tau3d_pack_halos_async(r->h, 0)) return 1;
  TAU_HIP(hipEventRecord(r->evE, r->S));
  TAU_HIP(hipEventRecord(r->evI, r->S));       // init / upload measured the field range on S: it is in the max words by now
  if (communicate(r, true)) return 1;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));
  if (tau3d_unpack_halos_async(r->h, 0)) return 1;   // (the first slab_begin unpacks the same buffers once more: idempotent)
  r->primed = true;
  return 0;
}
extern "C" int tau3d_ring_invalidate(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_invalidate: null ring");
  r->primed = false;
  return 0;
}

extern "C" int tau3d_ring_step_async(tau3d_ring_t *r, int nsteps) {
  if (!r) return tau::fail("tau3d_ring_step: null ring");
  TAU_HIP(hipSetDevice(r->device));
  if (!r->primed && tau3d_ring_prime(r)) return 1;
  const int E = r->edge;
  for (int s = 0; s < nsteps; s++) {
    TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));     // halos and max words of the step before have landed
    if (tau3d_slab_begin_async(r->h)) return 1;
    // Edge launches are E planes deep, not just the 3 that are sent: a marching launch pays a warm-up per chunk, so E = 8
    // keeps the edge launch at the duty of the interior one; the interior that hides the exchange is still ~0.5 ms at 512^2 x 48.
    if (tau3d_slab_edges_async(r->h, E)) return 1;
    TAU_HIP(hipEventRecord(r->evE, r->S));
    if (tau3d_slab_interior_async(r->h, E)) return 1;
    TAU_HIP(hipEventRecord(r->evI, r->S));
    if (communicate(r, true)) return 1;
    if (tau3d_slab_end_async(r->h)) return 1;
    r->steps++;
  }
  return 0;
}

extern "C" int tau3d_ring_finish(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_finish: null ring");
  TAU_HIP(hipSetDevice(r->device));
  TAU_HIP(hipStreamSynchronize(r->X));
  TAU_HIP(hipStreamSynchronize(r->S));
  return 0;
}

/* the clock after the last step: the controller of that step needs its all-reduced max, so wait for X first */
extern "C" int tau3d_ring_get_clock(tau3d_ring_t *r, tau3d_clock *out) {
  if (!r) return tau::fail("tau3d_ring_get_clock: null ring");
  TAU_HIP(hipSetDevice(r->device));
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));
  return tau3d_get_clock(r->h, out);
}

/* a barrier over the ring's ranks through the rendezvous file (host side; world 1: nothing) — the thin C driver's
 * start / stop line for timing */
extern "C" int tau3d_ring_barrier(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_barrier: null ring");
  if (!r->sh) return 0;
  return ring::barrier(r->sh, "tau3d_ring_barrier");
}
