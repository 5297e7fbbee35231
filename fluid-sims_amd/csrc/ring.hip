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
// That table is the round-2 schedule, kept behind TAU3D_RING_PIPELINE=0 (the edge depth TAU3D_RING_EDGE applies to it alone;
// the direct transports then put the twelve halo copies on X and the all-reduce on S behind the interior launch).  What runs
// by default, on EVERY transport, is ring_step_spec below (round 5): no edge / interior split, the x/y fluxes of all planes
// issued ahead of the all-reduce of the step before, exchange and ONE all-reduce per step beside them on X.  TAU3D_RING_SPEC=0
// selects the round-4 pipelined schedules (ring_step_pipelined / _packed: all-reduce first, then the exchange beside the x/y launch).
//
// No host synchronisation between the pieces: tau3d_ring_step_async(n) only enqueues.  One communicator, used on ONE stream
// (X) in the default and the round-4 pipelined schedules, tau3d_ring_prime included, so RCCL sees its operations in one order on
// every rank.  (TAU3D_RING_PIPELINE=0 and TAU3D_RING_AR_ON_S=1 put the direct transport's all-reduce on S, behind the interior
// launch; RCCL then orders the operations of two user streams.)
//
// Transports:
//   TAU3D_RING_RCCL   ncclSend / ncclRecv / ncclAllReduce over xGMI.  librccl is bound at run time (dlopen) — the copy the
//                     process already holds (PyTorch's, under Python) or the one beside the HIP runtime in use — so a
//                     single-GPU user of libtaueng never loads it.  world == 1 sends to / receives from itself.
//   TAU3D_RING_HOST   host-staged through the shared rendezvous file: D2H, process barrier, H2D.  Synchronous and slow; it
//                     exists so that N ranks SHARING one device (RCCL refuses duplicate GPUs) can run the ring's ordering
//                     — the multi-process tests on a one-GPU box — and as a fallback where RCCL is absent.
//   TAU3D_RING_LOCAL  world == 1 only: two device copies (periodic self-neighbour), no collective.
//   TAU3D_RING_IPC    direct halos, no compute unit involved: every rank maps its two neighbours' state allocations
//                     (hipIpcGetMemHandle / hipIpcOpenMemHandle, handles passed through the rendezvous file) and, once its
//                     edge planes are done, copies its new boundary planes STRAIGHT INTO THE NEIGHBOURS' HALO PLANES with
//                     hipMemcpyAsync on X (12 copies of 3 planes: the SDMA engines over xGMI) — no packed buffers, no pack in
//                     the edge launch, no unpack kernel (slab_begin shrinks to the one-thread clock kernel), and no RCCL
//                     send/recv kernel taking CUs from the VALU-bound interior launch that runs beside the exchange.  RCCL is
//                     kept for the 8-byte all-reduce only, and that all-reduce is also what orders the copies: a rank leaves
//                     all-reduce(n) only after every rank has entered it, i.e. after every neighbour's copies of step n
//                     (earlier on its X) have landed and every neighbour has finished reading the halo planes step n+1's
//                     copies will overwrite (its interior launch precedes its all-reduce).
//   TAU3D_RING_IPC_HOSTMAX  the same copies, all-reduce through the rendezvous file by the host: ranks may share a device
//                     (RCCL refuses that) — how the direct transport runs with 2-8 processes on a one-GPU test box.
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
  uint64_t created_ns;              // CLOCK_REALTIME when rank 0 made the file: tells a failed rank 0 of THIS launch from a leftover
  std::atomic<int32_t> bar_count, bar_gen;
  std::atomic<int32_t> status[MAX_WORLD];   // 0 unknown, 1 ready, 2 failed
  float maxw[MAX_WORLD][2];
  ncclUniqueId id;
  // per rank: device identity (one device per rank is checked by IDENTITY, not by counting visible devices: the usual
  // launch shows each rank exactly one device through HIP_VISIBLE_DEVICES) and what the IPC transport maps
  char busid[MAX_WORLD][32];
  hipIpcMemHandle_t grp[MAX_WORLD][2];   // the two ping-pong state allocations, by allocation index
  uint64_t field_stride[MAX_WORLD];      // floats between the six fields of an allocation
  int32_t nzl[MAX_WORLD];
  std::atomic<int32_t> cur[MAX_WORLD];   // allocation index of the rank's CURRENT state (published by tau3d_ring_prime)
  // then: world x 2 x slot_bytes of staging (host transport)
};
static size_t shared_bytes(int world, size_t slot) { return ((sizeof(Shared) + 4095) & ~(size_t)4095) + (size_t)world * 2 * slot; }
static char *slot_ptr(Shared *sh, int rank, int side) {
  return (char *)sh + ((sizeof(Shared) + 4095) & ~(size_t)4095) + ((size_t)rank * 2 + side) * sh->slot_bytes;
}
static uint64_t realtime_ns() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
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

// what a ring knows about its rendezvous file (shared by the 3D Z-slab ring and the row-slab ring of the 2D stencil handles)
struct tau_rendezvous {
  int rank = 0, world = 1, transport = 0;
  ring::Shared *sh = nullptr;
  size_t sh_bytes = 0;
  char path[256] = "";
  uint64_t ino = 0, dev = 0;   // identity of the mapped rendezvous file (ranks != 0)
  uint64_t key = 0;
  bool published = false, failed = false;
};
struct tau3d_ring : tau_rendezvous {
  tau3d_t *h = nullptr;
  int lo = 0, hi = 0;
  int nzl = 0, edge = 3, device = 0;
  hipStream_t S = nullptr, X = nullptr;
  hipStream_t X2 = nullptr;      // direct transports: the copies towards the HIGH neighbour (another xGMI link than the low one's)
  hipEvent_t evE = nullptr, evI = nullptr, evX = nullptr, evH = nullptr;
  hipEvent_t evJ = nullptr, evC2 = nullptr;   // fork of X2 from X / its join back (exchange_ipc)
  float *syncw = nullptr;        // one device word: the all-reduce that says "my halo copies have landed" (pipelined direct step)
  bool pipelined = false;        // x/y fluxes of step n+1 overlap the exchange of step n (ring_step_pipelined*; every transport)
  bool spec = false;             // ... and start AHEAD of all-reduce(n) (ring_step_spec): one all-reduce per step, off the critical path
  int inject_us = 0;             // TAU3D_RING_INJECT_AR_US: a spin kernel of that many us behind every all-reduce (latency pricing on one GPU)
  ncclComm_t comm = nullptr;
  float *buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [kind: 0 send, 1 recv][side]
  size_t nfloats = 0;
  float *maxw = nullptr;
  bool primed = false;
  long steps = 0;
  // IPC transport: the neighbours' state allocations in this process's address space, [side 0 lo / 1 hi][allocation index]
  float *peer_base[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  void *opened[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};    // what hipIpcCloseMemHandle gets back
  size_t peer_stride[2] = {0, 0};
  int peer_nzl[2] = {0, 0}, peer_cur[2] = {0, 0};
  // tau3d_ring_timing_*: HIP events on X around the exchange and around the all-reduce of every step (ring_step_spec), read
  // back by tau3d_ring_timing_read once the streams are idle — what one rank's communication stream spent where
  static constexpr int TIMED_STEPS = 256;
  bool timing = false;
  int timed = 0;
  hipEvent_t tev[TIMED_STEPS][3] = {};
  bool uses_rccl() const { return transport == TAU3D_RING_RCCL || transport == TAU3D_RING_IPC; }
  bool direct() const { return transport == TAU3D_RING_IPC || transport == TAU3D_RING_IPC_HOSTMAX; }
};

// a rank that cannot go on says so in the rendezvous file: its peers leave their barriers with an error instead of waiting
// out the timeout (or sitting inside a collective for ever)
static int ring_publish(tau_rendezvous *r);
static void mark_failed(tau_rendezvous *r) {
  if (!r || !r->sh || r->rank < 0 || r->rank >= ring::MAX_WORLD) return;
  r->failed = true;
  r->sh->status[r->rank].store(2, std::memory_order_release);
  // rank 0 failing before its file is public: make it public all the same (saving the message ring_publish might overwrite),
  // so that the waiting peers find the status word instead of sitting out the timeout; tau3d_ring_destroy then leaves the
  // file for them (a few KB in /dev/shm under a per-job name; the launcher removes it)
  if (r->rank == 0 && !r->published) {
    char keep[512];
    snprintf(keep, sizeof keep, "%s", tau_last_error());
    ring_publish(r);
    tau::fail("%s", keep);
  }
}

static int ring_map(tau_rendezvous *r, const char *path, uint64_t key, size_t slot) {
  using namespace ring;
  const size_t bytes = shared_bytes(r->world, slot);
  snprintf(r->path, sizeof r->path, "%s", path);
  int fd = -1;
  if (r->rank == 0) {
    // Rank 0 builds the file under a private name and rename()s it into place once the header is complete: whoever opens
    // `path` sees either nothing, a file of an EARLIER job (another key: ignored below), or this job's finished header.
    // First thing, before streams / RCCL / IPC handles are made: whatever an earlier job left under this name goes (a failed
    // job leaves its file for its own waiting peers, tau3d_ring_destroy) — the window in which a peer of THIS launch can pick
    // up a stale same-key file is then the start-up skew of the ranks, and what it may still find there is refused below.
    unlink(path);
    char tmp[300];
    snprintf(tmp, sizeof tmp, "%s.%ld.tmp", path, (long)getpid());
    unlink(tmp);
    fd = open(tmp, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) return tau::fail("tau3d_ring: cannot create %s: %s", tmp, strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); unlink(tmp); return tau::fail("tau3d_ring: ftruncate(%s, %zu): %s", tmp, bytes, strerror(errno)); }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { unlink(tmp); return tau::fail("tau3d_ring: mmap(%s): %s", tmp, strerror(errno)); }
    Shared *sh = new (m) Shared();   // ftruncate zero-filled it; the atomics start at 0
    sh->world = r->world; sh->transport = r->transport; sh->slot_bytes = slot;
    sh->created_ns = realtime_ns();
    r->sh = sh; r->sh_bytes = bytes;
    return 0;   // the caller puts the id in, publishes `ready` and calls ring_publish
  }
  const double t0 = now_s();
  const uint64_t entered_ns = realtime_ns();
  bool saw_failed = false;
  int spins = 0;
  for (;;) {   // wait for THIS job's file: an older one under the same name carries another key
    fd = open(path, O_RDWR);
    if (fd >= 0) {
      struct stat st;
      if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) {
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
          Shared *sh = (Shared *)m;
          // A file of THIS launch cannot have passed a barrier (this rank has not arrived at one), cannot hold a failed rank 0
          // that this rank should follow into a fresh start, and cannot have this rank's slot taken: any of the three says
          // "left behind by an earlier job with the same key" — keep waiting for rank 0 to replace it.
          const bool mine = sh->ready.load(std::memory_order_acquire) == (MAGIC ^ key);
          const bool rank0_failed = sh->status[0].load(std::memory_order_acquire) == 2;
          if (mine && rank0_failed && sh->created_ns >= entered_ns) {   // made after this rank started waiting: this launch's rank 0
            munmap(m, bytes);
            close(fd);
            return tau::fail("tau3d_ring: rank 0 failed before the ring was set up (rendezvous file %s)", path);
          }
          if (mine && rank0_failed) saw_failed = true;
          const bool stale = sh->bar_gen.load(std::memory_order_acquire) != 0 || rank0_failed ||
                             sh->status[r->rank].load(std::memory_order_acquire) != 0;
          if (!stale && mine) {
            close(fd);
            if (sh->world != r->world || sh->transport != r->transport || sh->slot_bytes != slot) {
              const int w = sh->world, t = sh->transport;
              const unsigned long long sb = sh->slot_bytes;
              munmap(m, bytes);
              return tau::fail("tau3d_ring: %s was made for world %d / transport %d / %llu-byte slots, this rank wants %d / %d / %zu",
                               path, w, t, sb, r->world, r->transport, slot);
            }
            r->sh = sh; r->sh_bytes = bytes;
            r->ino = (uint64_t)st.st_ino; r->dev = (uint64_t)st.st_dev;
            return 0;
          }
          munmap(m, bytes);
        }
      }
      close(fd);
    }
    if (now_s() - t0 > timeout_s())
      return tau::fail("tau3d_ring: rank %d waited %.0f s for rank 0's rendezvous file %s%s", r->rank, timeout_s(), path,
                       saw_failed ? " — a file with this job key whose rank 0 had failed was there all along: left by an earlier launch, or "
                                    "rank 0 of this launch failed before this rank started" : "");
    nap(spins);
  }
}
// rank 0: the header is complete -> the file appears under its public name
static int ring_publish(tau_rendezvous *r) {
  char tmp[300];
  snprintf(tmp, sizeof tmp, "%s.%ld.tmp", r->path, (long)getpid());
  r->sh->ready.store(ring::MAGIC ^ r->key, std::memory_order_release);
  if (rename(tmp, r->path) != 0) return tau::fail("tau3d_ring: rename(%s, %s): %s", tmp, r->path, strerror(errno));
  r->published = true;
  return 0;
}
// after the create barrier: the file this rank mapped must still be the one under `path` (a file left behind by a crashed job
// with the same path AND key could have been picked up before rank 0 replaced it — then the barrier was someone else's)
static int ring_check_same_file(tau_rendezvous *r) {
  if (r->rank == 0) return 0;
  struct stat st;
  if (stat(r->path, &st) != 0 || (uint64_t)st.st_ino != r->ino || (uint64_t)st.st_dev != r->dev)
    return tau::fail("tau3d_ring: rank %d mapped a stale rendezvous file at %s (left by an earlier job with the same job key); "
                     "use a job key that is unique per launch", r->rank, r->path);
  return 0;
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
  if (r->direct() && r->h) tau3d_set_halo_direct(r->h, 0);
  for (int s = 0; s < 2; s++)
    for (int g = 0; g < 2; g++)
      if (r->opened[s][g]) hipIpcCloseMemHandle(r->opened[s][g]);
  if (r->comm) ring::g_rccl.CommDestroy(r->comm);
  if (r->evE) hipEventDestroy(r->evE);
  if (r->evI) hipEventDestroy(r->evI);
  if (r->evX) hipEventDestroy(r->evX);
  if (r->evH) hipEventDestroy(r->evH);
  for (auto &st : r->tev)
    for (hipEvent_t e : st)
      if (e) hipEventDestroy(e);
  if (r->syncw) hipFree(r->syncw);
  if (r->evJ) hipEventDestroy(r->evJ);
  if (r->evC2) hipEventDestroy(r->evC2);
  if (r->X2) { hipStreamSynchronize(r->X2); hipStreamDestroy(r->X2); }
  if (r->X) hipStreamDestroy(r->X);
  if (r->sh) {
    munmap(r->sh, r->sh_bytes);
    if (r->rank == 0 && r->path[0] && !(r->failed && r->world > 1)) {   // the mappings of the other ranks keep the pages alive
      unlink(r->path);
      char tmp[300];
      snprintf(tmp, sizeof tmp, "%s.%ld.tmp", r->path, (long)getpid());
      unlink(tmp);
    }
  }
  delete r;
}

static int ring_create_impl(tau3d_ring *r, tau3d_t *h, int rank, int world, int transport, const char *rendezvous, uint64_t job_key) {
  using namespace ring;
  int z0 = 0, nzl = 0, nz = 0, device = 0;
  void *stream = nullptr;
  if (tau3d_slab_info(h, &z0, &nzl, &nz, &device, &stream)) return 1;
  r->h = h; r->rank = rank; r->world = world; r->transport = transport; r->nzl = nzl; r->device = device; r->key = job_key;
  r->lo = (rank + world - 1) % world; r->hi = (rank + 1) % world;
  r->S = (hipStream_t)stream;
  for (int k = 0; k < 2; k++)
    for (int s = 0; s < 2; s++)
      if (tau3d_halo_buf_ptr(h, k, s, &r->buf[k][s], &r->nfloats)) return 1;

  // the rendezvous file comes FIRST: whatever fails from here on is reported to the peers through it
  const bool need_file = world > 1 || (transport == TAU3D_RING_HOST && rendezvous && rendezvous[0]);
  if (need_file) {
    const size_t slot = transport == TAU3D_RING_HOST ? r->nfloats * sizeof(float) : 0;
    if (ring_map(r, rendezvous, job_key, slot)) return 1;
  }
  {
    int ez0, enzl;
    if (tau3d_slab_bounds(nz, world, rank, &ez0, &enzl)) return 1;
    if (ez0 != z0 || enzl != nzl)
      return tau::fail("tau3d_ring_create: rank %d of %d owns planes [%d,%d) of nz=%d, the handle was created for [%d,%d)", rank, world,
                       ez0, ez0 + enzl, nz, z0, z0 + nzl);
  }
  r->edge = nzl / 2 < 8 ? (nzl / 2 < 3 ? 3 : nzl / 2) : 8;   // planes per edge launch: >= the 3 halo planes, <= half a slab, 8 where they fit
  if (const char *e = getenv("TAU3D_RING_EDGE")) { const int v = atoi(e); if (v >= 3) r->edge = v; }
  TAU_HIP(hipSetDevice(device));
  TAU_HIP(hipStreamCreateWithFlags(&r->X, hipStreamNonBlocking));
  TAU_HIP(hipEventCreateWithFlags(&r->evE, hipEventDisableTiming));
  TAU_HIP(hipEventCreateWithFlags(&r->evI, hipEventDisableTiming));
  TAU_HIP(hipEventCreateWithFlags(&r->evX, hipEventDisableTiming));
  TAU_HIP(hipEventCreateWithFlags(&r->evH, hipEventDisableTiming));
  { const char *e = getenv("TAU3D_RING_PIPELINE"); r->pipelined = !(e && atoi(e) == 0); }
  { const char *e = getenv("TAU3D_RING_SPEC"); r->spec = r->pipelined && !(e && atoi(e) == 0); }
  if (const char *e = getenv("TAU3D_RING_INJECT_AR_US")) { const int v = atoi(e); r->inject_us = v > 0 ? v : 0; }
  if (r->direct()) {
    TAU_HIP(hipMalloc(&r->syncw, sizeof(float)));
    TAU_HIP(hipMemset(r->syncw, 0, sizeof(float)));
    const char *e = getenv("TAU3D_RING_ONE_COPY_STREAM");
    if (!(e && atoi(e) != 0)) {
      TAU_HIP(hipStreamCreateWithFlags(&r->X2, hipStreamNonBlocking));
      TAU_HIP(hipEventCreateWithFlags(&r->evJ, hipEventDisableTiming));
      TAU_HIP(hipEventCreateWithFlags(&r->evC2, hipEventDisableTiming));
    }
  }
  if (tau3d_max_ptr(h, &r->maxw)) return 1;

  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (r->uses_rccl()) {
    if (rccl_load()) return 1;
    if (rank == 0 && g_rccl.GetUniqueId(&id) != ncclSuccess) return tau::fail("tau3d_ring_create: ncclGetUniqueId failed");
  }
  // the two state allocations of this slab (by allocation index) for the IPC transport
  void *my_base[2] = {nullptr, nullptr};
  size_t my_stride = 0;
  if (r->direct()) {
    for (int w = 0; w < 2; w++) {
      void *b = nullptr;
      int idx = 0;
      if (tau3d_state_group(h, w, &b, nullptr, &my_stride, &idx)) return 1;
      my_base[idx] = b;
    }
  }
  if (r->sh) {
    Shared *sh = r->sh;
    char bus[32] = "";
    TAU_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof bus, device));
    snprintf(sh->busid[rank], sizeof sh->busid[rank], "%s", bus);
    sh->nzl[rank] = nzl;
    sh->field_stride[rank] = my_stride;
    if (r->direct() && world > 1)
      for (int g = 0; g < 2; g++) TAU_HIP(hipIpcGetMemHandle(&sh->grp[rank][g], my_base[g]));
    if (rank == 0) {
      sh->id = id;
      if (ring_publish(r)) return 1;
    }
    // every rank says whether it can go on BEFORE anyone enters ncclCommInitRank (which waits for all ranks for ever)
    sh->status[rank].store(1, std::memory_order_release);
    if (barrier(sh, "create")) return 1;
    if (ring_check_same_file(r)) return 1;
    for (int k = 0; k < world; k++)
      if (sh->status[k].load(std::memory_order_acquire) != 1) return tau::fail("tau3d_ring_create: rank %d could not start", k);
    id = sh->id;
    if (r->uses_rccl() && !getenv("TAU3D_RING_NO_DEVICE_CHECK"))   // RCCL refuses two ranks on one device deep inside its init: say it here
      for (int a = 0; a < world; a++)
        for (int b = a + 1; b < world; b++)
          if (strncmp(sh->busid[a], sh->busid[b], sizeof sh->busid[a]) == 0)
            return tau::fail("tau3d_ring_create: this transport needs one device per rank over RCCL, ranks %d and %d both run on %s "
                             "(the host / ipc-host transports let ranks share a device)", a, b, sh->busid[a]);
  }
  if (r->direct()) {
    for (int s = 0; s < 2; s++) {
      const int peer = s == 0 ? r->lo : r->hi;
      if (peer == rank) {                          // world 1: my own planes are my neighbour's
        for (int g = 0; g < 2; g++) r->peer_base[s][g] = (float *)my_base[g];
        r->peer_stride[s] = my_stride; r->peer_nzl[s] = nzl;
      } else if (s == 1 && r->hi == r->lo) {       // world 2: both neighbours are the same peer, mapped once
        for (int g = 0; g < 2; g++) r->peer_base[1][g] = r->peer_base[0][g];
        r->peer_stride[1] = r->peer_stride[0]; r->peer_nzl[1] = r->peer_nzl[0];
      } else {
        for (int g = 0; g < 2; g++) {
          void *p = nullptr;
          hipError_t e = hipIpcOpenMemHandle(&p, r->sh->grp[peer][g], hipIpcMemLazyEnablePeerAccess);
          if (e != hipSuccess)
            return tau::fail("tau3d_ring_create: hipIpcOpenMemHandle of rank %d's state (allocation %d): %s", peer, g, hipGetErrorString(e));
          r->opened[s][g] = p;
          r->peer_base[s][g] = (float *)p;
        }
        r->peer_stride[s] = (size_t)r->sh->field_stride[peer]; r->peer_nzl[s] = r->sh->nzl[peer];
      }
    }
    if (tau3d_set_halo_direct(h, 1)) return 1;
    if (r->sh && barrier(r->sh, "peer state mapped")) return 1;
  }
  if (r->uses_rccl()) TAU_NCCL(g_rccl.CommInitRank(&r->comm, world, id, rank));
  return 0;
}

extern "C" int tau3d_ring_create(tau3d_ring_t **out, tau3d_t *h, int rank, int world, int transport, const char *rendezvous,
                                 uint64_t job_key) {
  using namespace ring;
  if (!out || !h) return tau::fail("tau3d_ring_create: null argument");
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return tau::fail("tau3d_ring_create: rank %d of %d", rank, world);
  if (transport < TAU3D_RING_RCCL || transport > TAU3D_RING_IPC_HOSTMAX) return tau::fail("tau3d_ring_create: unknown transport %d", transport);
  if (transport == TAU3D_RING_LOCAL && world != 1) return tau::fail("tau3d_ring_create: the local transport is for world 1");
  if (world > 1 && (!rendezvous || !rendezvous[0])) return tau::fail("tau3d_ring_create: world %d needs a rendezvous path", world);
  if (world > 1 && job_key == 0)
    return tau::fail("tau3d_ring_create: world %d needs a non-zero job key, unique per launch (it tells this job's rendezvous file "
                     "from one an earlier job left under the same path)", world);
  tau3d_ring *r = new (std::nothrow) tau3d_ring();
  if (!r) return tau::fail("tau3d_ring_create: out of host memory");
  if (ring_create_impl(r, h, rank, world, transport, rendezvous, job_key)) {
    mark_failed(r);
    tau3d_ring_destroy(r);   // (tau::fail's message survives: destroy reports nothing)
    return 1;
  }
  *out = r;
  return 0;
}

extern "C" int tau3d_ring_info(tau3d_ring_t *r, int *rccl_version, int *comm_ranks, int *edge_planes, char *lib_path, size_t lib_path_len) {
  if (!r) return tau::fail("tau3d_ring_info: null ring");
  if (rccl_version) *rccl_version = 0;
  if (comm_ranks) *comm_ranks = r->uses_rccl() ? 0 : r->world;
  if (edge_planes) *edge_planes = r->edge;
  if (lib_path && lib_path_len) lib_path[0] = 0;
  if (r->uses_rccl()) {
    if (rccl_version) TAU_NCCL(ring::g_rccl.GetVersion(rccl_version));
    if (comm_ranks) TAU_NCCL(ring::g_rccl.CommCount(r->comm, comm_ranks));
    if (lib_path && lib_path_len) snprintf(lib_path, lib_path_len, "%s", ring::g_rccl.path);
  }
  return 0;
}

// Latency pricing on a one-GPU box (scripts/ring_rank_emulation.py --inject-allreduce-us): a world of one prices an all-reduce at
// the ~10 us of its launch; eight ranks over xGMI are plausibly 20-80 us.  One wave spinning on the realtime counter (100 MHz),
// enqueued right behind the collective on its stream, makes the step see that latency where it would sit.
__global__ void k_ring_spin(unsigned ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static int ring_allreduce(tau3d_ring *r, hipStream_t st) {
  using namespace ring;
  TAU_NCCL(g_rccl.AllReduce(r->maxw, r->maxw, 2, ncclFloat, ncclMax, r->comm, st));
  if (r->inject_us > 0) {
    hipLaunchKernelGGL(k_ring_spin, dim3(1), dim3(64), 0, st, (unsigned)r->inject_us * 100u);
    TAU_HIP(hipGetLastError());
  }
  return 0;
}
static int ring_inject(tau3d_ring *r, hipStream_t st) {   // (transports whose all-reduce is not RCCL's: local, host)
  if (r->inject_us > 0) {
    hipLaunchKernelGGL(k_ring_spin, dim3(1), dim3(64), 0, st, (unsigned)r->inject_us * 100u);
    TAU_HIP(hipGetLastError());
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
    if (a > w[0] || a != a) w[0] = a;   // NaN propagates, as ncclMax does (a blown-up field must not look bounded)
    if (c > w[1] || c != c) w[1] = c;
  }
  TAU_HIP(hipMemcpyAsync(r->maxw, w, sizeof w, hipMemcpyHostToDevice, r->X));
  TAU_HIP(hipStreamSynchronize(r->X));
  return barrier(r->sh, "max words read");
}

// direct halos: my first / last three interior planes of the state `which` (0 current, 1 next) go straight into the
// neighbours' halo planes of THEIR state `which` — the low neighbour's high halo, the high neighbour's low halo — field by
// field (a field's three planes are contiguous on both sides): 12 copies of 3 planes, nothing packed, nothing unpacked.
// peer_cur[] is the allocation index of each neighbour's current state (tau3d_ring_prime publishes it, every step flips it).
static int exchange_ipc(tau3d_ring *r, int which) {
  const size_t plane_n = r->nfloats / (6 * 3);
  const size_t bytes = 3 * plane_n * sizeof(float);
  void *base = nullptr;
  size_t stride = 0;
  if (tau3d_state_group(r->h, which, &base, nullptr, &stride, nullptr)) return 1;
  const float *me = (const float *)base;
  float *lo = r->peer_base[0][r->peer_cur[0] ^ which], *hi = r->peer_base[1][r->peer_cur[1] ^ which];
  // The two directions go to two different peers over two different xGMI links: on ONE stream the twelve copies run one after the
  // other — 37.7 MB at one link's rate, more than the x/y launch they are meant to hide behind —, so the copies towards the high
  // neighbour take a second stream that forks from X here (whatever X waits for, X2 waits for) and joins it again below: what
  // follows on X (the all-reduce, evX) still follows every copy.  (TAU3D_RING_ONE_COPY_STREAM=1: all on X, the round-4 form.)
  hipStream_t xh = r->X2 ? r->X2 : r->X;
  if (r->X2) {
    TAU_HIP(hipEventRecord(r->evJ, r->X));
    TAU_HIP(hipStreamWaitEvent(r->X2, r->evJ, 0));
  }
  for (int f = 0; f < 6; f++) {
    // (hipMemcpyDefault: the destination is another device's memory behind an IPC mapping — the runtime resolves both ends)
    TAU_HIP(hipMemcpyAsync(lo + f * r->peer_stride[0] + (size_t)(r->peer_nzl[0] + 3) * plane_n, me + f * stride + 3 * plane_n, bytes,
                           hipMemcpyDefault, r->X));
    TAU_HIP(hipMemcpyAsync(hi + f * r->peer_stride[1], me + f * stride + (size_t)r->nzl * plane_n, bytes, hipMemcpyDefault, xh));
  }
  if (r->X2) {
    TAU_HIP(hipEventRecord(r->evC2, r->X2));
    TAU_HIP(hipStreamWaitEvent(r->X, r->evC2, 0));
  }
  return 0;
}

// the communication of one step (or of the priming exchange, which = 0): X waits for evE (boundary planes / send buffers
// written), exchanges, waits for evI (all launches that raise the max words), reduces them, and records evX
static int communicate(tau3d_ring *r, bool with_max, int which = 1) {
  using namespace ring;
  TAU_HIP(hipStreamWaitEvent(r->X, r->evE, 0));
  switch (r->transport) {
    case TAU3D_RING_RCCL: if (exchange_rccl(r)) return 1; break;
    case TAU3D_RING_HOST: if (exchange_host(r)) return 1; break;
    case TAU3D_RING_IPC: case TAU3D_RING_IPC_HOSTMAX: if (exchange_ipc(r, which)) return 1; break;
    default: if (exchange_local(r)) return 1; break;
  }
  if (r->transport == TAU3D_RING_IPC) {
    // The copies are the only thing on X; the all-reduce runs on the COMPUTE stream, behind the interior launch and behind the
    // copies (which are long done by then): no stream hop on either side of it — the all-reduce sits on the critical path of
    // every step (the next step's clock needs its result), the copies do not.  A rank thus enters all-reduce(n) only after ITS
    // copies of step n have landed, which is the ordering the direct transport rests on (top of this file).
    TAU_HIP(hipEventRecord(r->evX, r->X));
    TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));
    if (with_max && ring_allreduce(r, r->S)) return 1;
    return 0;
  }
  if (with_max && r->transport != TAU3D_RING_LOCAL) {
    TAU_HIP(hipStreamWaitEvent(r->X, r->evI, 0));
    if (r->uses_rccl()) { if (ring_allreduce(r, r->X)) return 1; }
    else if (allreduce_host(r)) return 1;
  }
  TAU_HIP(hipEventRecord(r->evX, r->X));
  return 0;
}

/* exchange the halos of the CURRENT state and agree on its field range (after init / upload): without it the first step
 * would read undefined halo planes and every slab could pick its own WENO weight form */
static int ring_prime_impl(tau3d_ring *r) {
  TAU_HIP(hipSetDevice(r->device));
  if (r->direct()) {
    // Host-synchronous (it runs once after create / init / upload): every rank's state is final and nobody reads halo planes
    // any more -> tell the neighbours which allocation holds my current state -> copy -> reduce the field range -> all landed.
    TAU_HIP(hipStreamSynchronize(r->S));
    TAU_HIP(hipStreamSynchronize(r->X));
    int idx = 0;
    if (tau3d_state_group(r->h, 0, nullptr, nullptr, nullptr, &idx)) return 1;
    r->peer_cur[0] = r->peer_cur[1] = idx;
    if (r->sh) {
      r->sh->cur[r->rank].store(idx, std::memory_order_release);
      if (ring::barrier(r->sh, "prime: states final")) return 1;
      r->peer_cur[0] = r->sh->cur[r->lo].load(std::memory_order_acquire);
      r->peer_cur[1] = r->sh->cur[r->hi].load(std::memory_order_acquire);
    }
    TAU_HIP(hipEventRecord(r->evE, r->S));
    TAU_HIP(hipEventRecord(r->evI, r->S));
    if (r->pipelined && r->transport == TAU3D_RING_IPC) {
      // as the default step does it: copies, then the all-reduce, both on X — the communicator never sees another stream
      TAU_HIP(hipStreamWaitEvent(r->X, r->evE, 0));
      if (exchange_ipc(r, 0) || ring_allreduce(r, r->X)) return 1;
      TAU_HIP(hipEventRecord(r->evX, r->X));
    } else if (communicate(r, true, 0)) return 1;
    TAU_HIP(hipStreamSynchronize(r->X));
    TAU_HIP(hipStreamSynchronize(r->S));
    if (r->sh && ring::barrier(r->sh, "prime: halos landed")) return 1;
    r->primed = true;
    return 0;
  }
  if (tau3d_pack_halos_async(r->h, 0)) return 1;
  TAU_HIP(hipEventRecord(r->evE, r->S));
  TAU_HIP(hipEventRecord(r->evI, r->S));       // init / upload measured the field range on S: it is in the max words by now
  if (communicate(r, true)) return 1;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));
  if (tau3d_unpack_halos_async(r->h, 0)) return 1;   // (the first slab_begin unpacks the same buffers once more: idempotent)
  r->primed = true;
  return 0;
}
extern "C" int tau3d_ring_prime(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_prime: null ring");
  const int rc = ring_prime_impl(r);
  if (rc) mark_failed(r);
  return rc;
}
extern "C" int tau3d_ring_invalidate(tau3d_ring_t *r) {
  if (!r) return tau::fail("tau3d_ring_invalidate: null ring");
  r->primed = false;
  return 0;
}

static int ring_step_impl(tau3d_ring *r, int nsteps);
extern "C" int tau3d_ring_step_async(tau3d_ring_t *r, int nsteps) {
  if (!r) return tau::fail("tau3d_ring_step: null ring");
  const int rc = ring_step_impl(r, nsteps);
  if (rc) mark_failed(r);   // peers leave their next barrier with an error instead of waiting for this rank
  return rc;
}
// The direct transports' step, software-pipelined over the step boundary.  k_flux_xy reads no halo plane, so the x/y fluxes of
// step n+1 (57 % of a step) need only the all-reduced max of step n — not its halos: the copies of step n run on X beside them,
// and nothing is split into edge and interior launches (a 64-plane slab then costs what a plain 64-plane domain costs plus the
// all-reduce: rocprof timeline 912 -> ~850 us per step against 833 plain).
//   S: [all-reduce(n-1) is on S]  begin (clock)   xy fluxes, ALL planes   wait evH(n-1)   z + update, ALL planes   record evI
//      all-reduce(max)(n)
//   X: wait evI   12 halo copies (n)   all-reduce(one word) = "my copies have landed"   record evH(n)
// The second, one-word all-reduce is what lets a rank's z kernel of step n+1 read its halo planes: it completes only after
// every rank's copies of step n completed on its X.  Both all-reduces use the one communicator: RCCL runs a communicator's
// operations in issue order, and the issue order (max(n) on S, landed(n) on X, max(n+1) on S ...) is the same on every rank.
// Writes into a neighbour's halo planes cannot overtake its reads: copies(n+2) — the next ones into the same allocation —
// follow this rank's z(n+2), hence all-reduce(n+1), hence the neighbour's z(n+1), the last reader.
// IPC_HOSTMAX (ranks sharing a device, tests): copies, then the host all-reduce whose barriers also say "copies landed".
static int ring_step_pipelined(tau3d_ring *r) {
  using namespace ring;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));      // (HOSTMAX: the host all-reduce + copies of the step before, on X)
  if (tau3d_slab_begin_async(r->h)) return 1;
  if (tau3d_slab_xy_async(r->h)) return 1;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evH, 0));      // every rank's halo copies of the step before have landed
  if (tau3d_slab_z_async(r->h)) return 1;
  TAU_HIP(hipEventRecord(r->evI, r->S));
  TAU_HIP(hipStreamWaitEvent(r->X, r->evI, 0));
  if (r->transport == TAU3D_RING_IPC) {
    // Both all-reduces on X by default: ONE communicator used on ONE stream, the plainest contract RCCL offers (the max on S
    // itself saves the S -> X -> S hop, ~1 % of a 64-plane step, but has RCCL order two user streams: TAU3D_RING_AR_ON_S=1).
    static const bool ar_on_s = [] { const char *e = getenv("TAU3D_RING_AR_ON_S"); return e && atoi(e) != 0; }();
    if (ar_on_s) { if (ring_allreduce(r, r->S)) return 1; }
    else {
      if (ring_allreduce(r, r->X)) return 1;
      TAU_HIP(hipEventRecord(r->evX, r->X));   // (the next step's clock waits for it: top of this function)
    }
    if (exchange_ipc(r, 1)) return 1;
    TAU_NCCL(g_rccl.AllReduce(r->syncw, r->syncw, 1, ncclFloat, ncclMax, r->comm, r->X));
    if (ring_inject(r, r->X)) return 1;
    TAU_HIP(hipEventRecord(r->evH, r->X));
  } else {
    // copies first: the host all-reduce behind them synchronises X (so this rank's copies have landed) before its first
    // barrier — past that barrier every rank's copies have landed, and the reduced max follows two barriers later
    if (exchange_ipc(r, 1)) return 1;
    if (allreduce_host(r)) return 1;
    TAU_HIP(hipEventRecord(r->evX, r->X));
  }
  if (tau3d_slab_end_async(r->h)) return 1;
  r->peer_cur[0] ^= 1; r->peer_cur[1] ^= 1;
  r->steps++;
  return 0;
}
// The same pipelining for the packed transports (rccl, host, local): the all-reduce goes FIRST on X (the next step's clock waits
// for it alone), the send / recv group behind it runs beside the next step's x/y flux launch, and the received planes are
// unpacked just before the z kernel.
//   S: wait evX(n-1) [all-reduce]   clock   xy fluxes, ALL planes   wait evH(n-1) [exchange]   unpack   z + update + pack   record evI
//   X: wait evI   all-reduce(max)(n)   record evX(n)   send / recv (n)   record evH(n)
// Receive buffers: Recv(n) follows z(n), which follows unpack of what Recv(n-1) brought.  Send buffers: z(n+1) packs them after
// evH(n), i.e. after this rank's sends of step n completed.
static int ring_step_pipelined_packed(tau3d_ring *r) {
  using namespace ring;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));
  if (tau3d_slab_clock_async(r->h)) return 1;
  if (tau3d_slab_xy_async(r->h)) return 1;
  TAU_HIP(hipStreamWaitEvent(r->S, r->evH, 0));
  if (tau3d_unpack_halos_async(r->h, 0)) return 1;
  if (tau3d_slab_z_async(r->h)) return 1;
  TAU_HIP(hipEventRecord(r->evI, r->S));
  TAU_HIP(hipStreamWaitEvent(r->X, r->evI, 0));
  if (r->transport == TAU3D_RING_RCCL) { if (ring_allreduce(r, r->X)) return 1; }
  else if (r->transport == TAU3D_RING_HOST) { if (allreduce_host(r)) return 1; }
  else if (ring_inject(r, r->X)) return 1;
  TAU_HIP(hipEventRecord(r->evX, r->X));
  switch (r->transport) {
    case TAU3D_RING_RCCL: if (exchange_rccl(r)) return 1; break;
    case TAU3D_RING_HOST: if (exchange_host(r)) return 1; break;
    default: if (exchange_local(r)) return 1; break;
  }
  TAU_HIP(hipEventRecord(r->evH, r->X));
  if (tau3d_slab_end_async(r->h)) return 1;
  r->steps++;
  return 0;
}
// Round 5 — the default: the x/y fluxes run AHEAD of the all-reduce.  What k_flux_xy(n+1) takes from all-reduce(n) is one bit:
// on which side of the WENO weight-form limit the global field range lies (the inflow gain and dt enter in the z kernel only).
// That bit flips once in a run, if ever, so the launch goes ahead with the range of the step before; the clock kernel — which
// commits the all-reduced range — records whether the bit moved, and a near-empty second launch repeats the fluxes if it did
// (tau3d_slab_xy_fix_async): results do not depend on the speculation, only the timing does.  With nothing waiting for the
// all-reduce until the z kernel, ONE all-reduce per step is enough for every transport — issued BEHIND the exchange on X, so its
// completion also says "every rank's halos of this step have landed" (what the second, one-word all-reduce of the round-4
// schedule was for) — and exchange + all-reduce run beside the x/y launch, 57 % of a step:
//   S: xy fluxes (n+1), ALL planes   wait evX(n)   clock (controller n, clock n+1, range commit)   xy repeat (empty)
//      [packed: unpack]   z + update (n+1) [packed: + pack]   record evI
//   X: wait evI   halo copies | send / recv (n+1)   all-reduce(max)(n+1)   record evX
// Hazards (direct copies): copies(n+1) overwrite halo planes the neighbour's z(n) read; they follow this rank's z(n+1), hence
// all-reduce(n) complete, hence the neighbour's all-reduce(n) issued — behind its z(n).  The speculative launch reads interior
// planes and the divergence buffer only: neither is written by a neighbour, and stream order keeps it behind this rank's z(n).
// (TAU3D_RING_SPEC=0: the round-4 schedules below, all-reduce first.)
static int ring_step_spec(tau3d_ring *r) {
  using namespace ring;
  if (tau3d_slab_xy_async(r->h)) return 1;               // ahead of the clock: reads the range word of the step before
  TAU_HIP(hipStreamWaitEvent(r->S, r->evX, 0));          // exchange + all-reduce of the step before
  if (tau3d_slab_clock_async(r->h)) return 1;
  if (tau3d_slab_xy_fix_async(r->h)) return 1;
  if (!r->direct() && tau3d_unpack_halos_async(r->h, 0)) return 1;
  if (tau3d_slab_z_async(r->h)) return 1;
  TAU_HIP(hipEventRecord(r->evI, r->S));
  TAU_HIP(hipStreamWaitEvent(r->X, r->evI, 0));
  const bool timed = r->timing && r->timed < tau3d_ring::TIMED_STEPS;
  hipEvent_t *te = timed ? r->tev[r->timed] : nullptr;
  if (timed) {
    for (int k = 0; k < 3; k++)
      if (!te[k]) TAU_HIP(hipEventCreate(&te[k]));
    TAU_HIP(hipEventRecord(te[0], r->X));
  }
  auto mid = [&]() -> int { return timed ? (int)(hipEventRecord(te[1], r->X) != hipSuccess) : 0; };
  switch (r->transport) {
    case TAU3D_RING_RCCL: if (exchange_rccl(r) || mid() || ring_allreduce(r, r->X)) return 1; break;
    case TAU3D_RING_IPC: if (exchange_ipc(r, 1) || mid() || ring_allreduce(r, r->X)) return 1; break;
    case TAU3D_RING_HOST: if (exchange_host(r) || mid() || allreduce_host(r)) return 1; break;
    case TAU3D_RING_IPC_HOSTMAX: if (exchange_ipc(r, 1) || mid() || allreduce_host(r)) return 1; break;   // (its first barrier follows a sync of X: copies landed)
    default: if (exchange_local(r) || mid() || ring_inject(r, r->X)) return 1; break;
  }
  if (timed) { TAU_HIP(hipEventRecord(te[2], r->X)); r->timed++; }
  TAU_HIP(hipEventRecord(r->evX, r->X));
  if (tau3d_slab_end_async(r->h)) return 1;
  r->peer_cur[0] ^= 1; r->peer_cur[1] ^= 1;
  r->steps++;
  return 0;
}
static int ring_step_impl(tau3d_ring *r, int nsteps) {
  TAU_HIP(hipSetDevice(r->device));
  if (!r->primed && ring_prime_impl(r)) return 1;
  if (r->spec) {
    for (int s = 0; s < nsteps; s++)
      if (ring_step_spec(r)) return 1;
    return 0;
  }
  if (r->pipelined) {
    for (int s = 0; s < nsteps; s++)
      if (r->direct() ? ring_step_pipelined(r) : ring_step_pipelined_packed(r)) return 1;
    return 0;
  }
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
    r->peer_cur[0] ^= 1; r->peer_cur[1] ^= 1;   // every rank swaps its two allocations with every step
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

/* Per-step event timing of the communication stream (default schedule, ring_step_spec): enable, step, tau3d_ring_finish, read.
 * exchange_ms: halo copies / send-recv of this rank; allreduce_ms: the max all-reduce behind them — which also waits for the
 * slowest rank's exchange, so it carries the skew between ranks.  Sums over `steps` timed steps (at most 256 per enable). */
extern "C" int tau3d_ring_timing_enable(tau3d_ring_t *r, int on) {
  if (!r) return tau::fail("tau3d_ring_timing_enable: null ring");
  r->timing = on != 0;
  r->timed = 0;
  return 0;
}
extern "C" int tau3d_ring_timing_read(tau3d_ring_t *r, double *exchange_ms, double *allreduce_ms, int *steps) {
  if (!r || !exchange_ms || !allreduce_ms || !steps) return tau::fail("tau3d_ring_timing_read: null argument");
  TAU_HIP(hipSetDevice(r->device));
  TAU_HIP(hipStreamSynchronize(r->X));
  double ex = 0.0, ar = 0.0;
  for (int s = 0; s < r->timed; s++) {
    float a = 0.f, b = 0.f;
    TAU_HIP(hipEventElapsedTime(&a, r->tev[s][0], r->tev[s][1]));
    TAU_HIP(hipEventElapsedTime(&b, r->tev[s][1], r->tev[s][2]));
    ex += a; ar += b;
  }
  *exchange_ms = ex; *allreduce_ms = ar; *steps = r->timed;
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

// =====================================================================================================================
// The row-slab ring of the periodic two-field 5-point-stencil handles — Gray-Scott (taugs_*) and the Burgers / shallow-water
// viscosity passes (taulap_*) — in the library (round 5; SURVEY §8e: "also slab-shardable, 1-row halo, periodic"; the loop it
// shards is tau_gray_scott.cu:321-329).  Same scheme as fluid-sims_amd/slab2d.py, without Python or torch.distributed:
//
// Rank r owns rows [y0, y0 + nyl) of the ny x nx grid and keeps H halo rows on each side; its handle is created with
// ny_local = nyl + 2H rows and steps that array as the periodic domain it believes it has.  What wraps around the ends of the
// local array is wrong, but a 5-point stencil carries that one row per step: after k <= H steps only the outer k rows of each
// halo are contaminated and every owned row is what the single-domain run computes — same kernel, same operands, bit for bit.
// Every H steps the halos are refreshed from the ring neighbours (first H owned rows -> the low neighbour's high halo, last H
// owned rows -> the high neighbour's low halo): 2 fields x H x nx x 4 B a side — 262 KB at nx = 8192, H = 4, one exchange per
// four-level fused pass.  A deeper halo (H = 8, 16 ...) trades 2H / nyl of redundant rows for fewer, larger exchanges.
// No global reduction (fixed dt), no pack kernel (H rows of a field are contiguous: RCCL sends and receives them in place).
// Transports: TAU3D_RING_RCCL (ncclSend / ncclRecv on the handle's stream), TAU3D_RING_HOST (staged through the rendezvous
// file: ranks may share a device — the multi-process tests of a one-GPU box), TAU3D_RING_LOCAL (world 1, device copies).
struct taurow_ring : tau_rendezvous {
  void *h = nullptr;
  int (*ptrs)(void *, float **, float **) = nullptr;
  int (*step)(void *, int) = nullptr;
  int nx = 0, nyl = 0, H = 0, device = 0, lo = 0, hi = 0;
  hipStream_t S = nullptr;
  ncclComm_t comm = nullptr;
  long steps = 0, exchanges = 0;
};

extern "C" int taurow_bounds(int ny, int world, int rank, int *y0, int *nyl) {
  if (world < 1 || rank < 0 || rank >= world || !y0 || !nyl) return tau::fail("taurow_bounds: bad argument");
  const int base = ny / world, rem = ny % world;
  *y0 = rank * base + (rank < rem ? rank : rem);
  *nyl = base + (rank < rem ? 1 : 0);
  if (*nyl < 1) return tau::fail("taurow_bounds: ny=%d over %d ranks leaves an empty slab", ny, world);
  return 0;
}

static int rowring_exchange(taurow_ring *r) {
  using namespace ring;
  float *f[2] = {nullptr, nullptr};
  if (r->ptrs(r->h, &f[0], &f[1])) return 1;
  const size_t n = (size_t)r->H * r->nx, b = n * sizeof(float);
  const size_t first = (size_t)r->H * r->nx, last = (size_t)r->nyl * r->nx, hihalo = (size_t)(r->nyl + r->H) * r->nx;
  switch (r->transport) {
    case TAU3D_RING_RCCL:
      // one group; between one pair of ranks RCCL matches sends and receives in issue order (world 2: both neighbours are one
      // peer): per field, send (first rows -> lo, last rows -> hi), receive (high halo <- hi, low halo <- lo)
      TAU_NCCL(g_rccl.GroupStart());
      for (int k = 0; k < 2; k++) {
        TAU_NCCL(g_rccl.Send(f[k] + first, n, ncclFloat, r->lo, r->comm, r->S));
        TAU_NCCL(g_rccl.Send(f[k] + last, n, ncclFloat, r->hi, r->comm, r->S));
        TAU_NCCL(g_rccl.Recv(f[k] + hihalo, n, ncclFloat, r->hi, r->comm, r->S));
        TAU_NCCL(g_rccl.Recv(f[k], n, ncclFloat, r->lo, r->comm, r->S));
      }
      TAU_NCCL(g_rccl.GroupEnd());
      break;
    case TAU3D_RING_HOST:
      if (r->sh) {
        for (int k = 0; k < 2; k++) {
          TAU_HIP(hipMemcpyAsync(slot_ptr(r->sh, r->rank, 0) + k * b, f[k] + first, b, hipMemcpyDeviceToHost, r->S));
          TAU_HIP(hipMemcpyAsync(slot_ptr(r->sh, r->rank, 1) + k * b, f[k] + last, b, hipMemcpyDeviceToHost, r->S));
        }
        TAU_HIP(hipStreamSynchronize(r->S));
        if (barrier(r->sh, "row halo slots written")) return 1;
        for (int k = 0; k < 2; k++) {
          TAU_HIP(hipMemcpyAsync(f[k] + hihalo, slot_ptr(r->sh, r->hi, 0) + k * b, b, hipMemcpyHostToDevice, r->S));
          TAU_HIP(hipMemcpyAsync(f[k], slot_ptr(r->sh, r->lo, 1) + k * b, b, hipMemcpyHostToDevice, r->S));
        }
        TAU_HIP(hipStreamSynchronize(r->S));
        if (barrier(r->sh, "row halo slots read")) return 1;
        break;
      }
      // (no rendezvous file: a world of one)
      [[fallthrough]];
    default:
      for (int k = 0; k < 2; k++) {
        TAU_HIP(hipMemcpyAsync(f[k] + hihalo, f[k] + first, b, hipMemcpyDeviceToDevice, r->S));
        TAU_HIP(hipMemcpyAsync(f[k], f[k] + last, b, hipMemcpyDeviceToDevice, r->S));
      }
      break;
  }
  r->exchanges++;
  return 0;
}

static int rowring_create_impl(taurow_ring *r, int nx, int ny_local, int device, void *stream, int halo, int rank, int world, int transport,
                               const char *rendezvous, uint64_t job_key) {
  using namespace ring;
  r->rank = rank; r->world = world; r->transport = transport; r->key = job_key;
  r->nx = nx; r->H = halo; r->nyl = ny_local - 2 * halo; r->device = device; r->S = (hipStream_t)stream;
  r->lo = (rank + world - 1) % world; r->hi = (rank + 1) % world;
  // arguments first: nothing is mapped, and no peer can be attached to anything, when a bad halo depth is refused
  if (halo < 1 || r->nyl < halo)
    return tau::fail("taurow_ring_create: a handle of %d rows with a %d-row halo either side owns %d rows; it needs at least the halo depth",
                     ny_local, halo, r->nyl);
  const size_t slot = transport == TAU3D_RING_HOST ? 2 * (size_t)halo * nx * sizeof(float) : 0;
  if (world > 1 && ring_map(r, rendezvous, job_key, slot)) return 1;
  TAU_HIP(hipSetDevice(device));
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (transport == TAU3D_RING_RCCL) {
    if (rccl_load()) return 1;
    if (rank == 0 && g_rccl.GetUniqueId(&id) != ncclSuccess) return tau::fail("taurow_ring_create: ncclGetUniqueId failed");
  }
  if (r->sh) {
    Shared *sh = r->sh;
    char bus[32] = "";
    TAU_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof bus, device));
    snprintf(sh->busid[rank], sizeof sh->busid[rank], "%s", bus);
    sh->nzl[rank] = r->nyl;
    sh->field_stride[rank] = (uint64_t)nx | ((uint64_t)halo << 32);   // what every rank must agree on
    if (rank == 0) {
      sh->id = id;
      if (ring_publish(r)) return 1;
    }
    sh->status[rank].store(1, std::memory_order_release);
    if (barrier(sh, "row ring create")) return 1;
    if (ring_check_same_file(r)) return 1;
    for (int k = 0; k < world; k++) {
      if (sh->status[k].load(std::memory_order_acquire) != 1) return tau::fail("taurow_ring_create: rank %d could not start", k);
      if (sh->field_stride[k] != sh->field_stride[rank])
        return tau::fail("taurow_ring_create: rank %d runs nx=%u halo=%u, rank %d nx=%d halo=%d", k, (unsigned)sh->field_stride[k],
                         (unsigned)(sh->field_stride[k] >> 32), rank, nx, halo);
      if (sh->nzl[k] < halo) return tau::fail("taurow_ring_create: rank %d owns %d rows, fewer than the %d-row halo", k, sh->nzl[k], halo);
    }
    id = sh->id;
    if (transport == TAU3D_RING_RCCL && !getenv("TAU3D_RING_NO_DEVICE_CHECK"))
      for (int a = 0; a < world; a++)
        for (int b = a + 1; b < world; b++)
          if (strncmp(sh->busid[a], sh->busid[b], sizeof sh->busid[a]) == 0)
            return tau::fail("taurow_ring_create: the RCCL transport needs one device per rank, ranks %d and %d both run on %s "
                             "(the host transport lets ranks share a device)", a, b, sh->busid[a]);
  }
  if (transport == TAU3D_RING_RCCL) TAU_NCCL(g_rccl.CommInitRank(&r->comm, world, id, rank));
  return 0;
}

static int rowring_create(taurow_ring_t **out, void *h, int (*ptrs)(void *, float **, float **), int (*step)(void *, int), int nx, int ny_local,
                          int device, void *stream, int halo, int rank, int world, int transport, const char *rendezvous, uint64_t job_key) {
  using namespace ring;
  if (!out || !h) return tau::fail("taurow_ring_create: null argument");
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return tau::fail("taurow_ring_create: rank %d of %d", rank, world);
  if (transport != TAU3D_RING_RCCL && transport != TAU3D_RING_HOST && transport != TAU3D_RING_LOCAL)
    return tau::fail("taurow_ring_create: transport %d (the row ring runs over rccl, host or local)", transport);
  if (transport == TAU3D_RING_LOCAL && world != 1) return tau::fail("taurow_ring_create: the local transport is for world 1");
  if (world > 1 && (!rendezvous || !rendezvous[0])) return tau::fail("taurow_ring_create: world %d needs a rendezvous path", world);
  if (world > 1 && job_key == 0) return tau::fail("taurow_ring_create: world %d needs a non-zero job key, unique per launch", world);
  taurow_ring *r = new (std::nothrow) taurow_ring();
  if (!r) return tau::fail("taurow_ring_create: out of host memory");
  r->h = h; r->ptrs = ptrs; r->step = step;
  if (rowring_create_impl(r, nx, ny_local, device, stream, halo, rank, world, transport, rendezvous, job_key)) {
    char keep[512];
    snprintf(keep, sizeof keep, "%s", tau_last_error());
    mark_failed(r);
    taurow_ring_destroy(r);
    return tau::fail("%s", keep);
  }
  *out = r;
  return 0;
}
extern "C" int taugs_ring_create(taurow_ring_t **out, taugs_t *h, int halo, int rank, int world, int transport, const char *rendezvous,
                                 uint64_t job_key) {
  int nx = 0, ny = 0, device = 0;
  void *stream = nullptr;
  if (!h || taugs_info(h, &nx, &ny, &device, &stream)) return tau::fail("taugs_ring_create: null handle");
  return rowring_create(out, h, [](void *p, float **a, float **b) { return taugs_state_ptrs((taugs_t *)p, a, b); },
                        [](void *p, int n) { return taugs_step_async((taugs_t *)p, n); }, nx, ny, device, stream, halo, rank, world, transport,
                        rendezvous, job_key);
}
extern "C" int taulap_ring_create(taurow_ring_t **out, taulap_t *h, int halo, int rank, int world, int transport, const char *rendezvous,
                                  uint64_t job_key) {
  int nx = 0, ny = 0, device = 0;
  void *stream = nullptr;
  if (!h || taulap_info(h, &nx, &ny, &device, &stream)) return tau::fail("taulap_ring_create: null handle");
  return rowring_create(out, h, [](void *p, float **a, float **b) { return taulap_state_ptrs((taulap_t *)p, a, b); },
                        [](void *p, int n) { return taulap_step_async((taulap_t *)p, n); }, nx, ny, device, stream, halo, rank, world, transport,
                        rendezvous, job_key);
}
extern "C" void taurow_ring_destroy(taurow_ring_t *r) {
  if (!r) return;
  hipSetDevice(r->device);
  if (r->S) hipStreamSynchronize(r->S);
  if (r->comm) ring::g_rccl.CommDestroy(r->comm);
  if (r->sh) {
    munmap(r->sh, r->sh_bytes);
    if (r->rank == 0 && r->path[0] && !(r->failed && r->world > 1)) {
      unlink(r->path);
      char tmp[300];
      snprintf(tmp, sizeof tmp, "%s.%ld.tmp", r->path, (long)getpid());
      unlink(tmp);
    }
  }
  delete r;
}
/* refresh both halos of the current state from the ring neighbours (after init / upload, and every `halo` steps) */
extern "C" int taurow_ring_exchange_async(taurow_ring_t *r) {
  if (!r) return tau::fail("taurow_ring_exchange: null ring");
  TAU_HIP(hipSetDevice(r->device));
  const int rc = rowring_exchange(r);
  if (rc) mark_failed(r);
  return rc;
}
/* nsteps time steps (any count: the last stretch may be shorter than the halo); the halos are current when it returns */
extern "C" int taurow_ring_step_async(taurow_ring_t *r, int nsteps) {
  if (!r) return tau::fail("taurow_ring_step: null ring");
  TAU_HIP(hipSetDevice(r->device));
  for (int done = 0; done < nsteps;) {
    const int k = nsteps - done < r->H ? nsteps - done : r->H;
    if (r->step(r->h, k) || rowring_exchange(r)) { mark_failed(r); return 1; }
    done += k;
    r->steps += k;
  }
  return 0;
}
extern "C" int taurow_ring_finish(taurow_ring_t *r) {
  if (!r) return tau::fail("taurow_ring_finish: null ring");
  TAU_HIP(hipSetDevice(r->device));
  TAU_HIP(hipStreamSynchronize(r->S));
  return 0;
}
extern "C" int taurow_ring_barrier(taurow_ring_t *r) {
  if (!r) return tau::fail("taurow_ring_barrier: null ring");
  if (!r->sh) return 0;
  return ring::barrier(r->sh, "taurow_ring_barrier");
}
extern "C" int taurow_ring_info(taurow_ring_t *r, int *nyl, int *halo, long *exchanges, int *rccl_version, int *comm_ranks) {
  if (!r) return tau::fail("taurow_ring_info: null ring");
  if (nyl) *nyl = r->nyl;
  if (halo) *halo = r->H;
  if (exchanges) *exchanges = r->exchanges;
  if (rccl_version) *rccl_version = 0;
  if (comm_ranks) *comm_ranks = r->comm ? 0 : r->world;
  if (r->comm) {
    if (rccl_version) TAU_NCCL(ring::g_rccl.GetVersion(rccl_version));
    if (comm_ranks) TAU_NCCL(ring::g_rccl.CommCount(r->comm, comm_ranks));
  }
  return 0;
}
// the names the two simulators' drivers use (one implementation: the ring does not care which stencil steps its rows)
extern "C" int taugs_ring_step_async(taurow_ring_t *r, int nsteps) { return taurow_ring_step_async(r, nsteps); }
extern "C" int taugs_ring_exchange_async(taurow_ring_t *r) { return taurow_ring_exchange_async(r); }
extern "C" int taugs_ring_finish(taurow_ring_t *r) { return taurow_ring_finish(r); }
extern "C" int taugs_ring_barrier(taurow_ring_t *r) { return taurow_ring_barrier(r); }
extern "C" void taugs_ring_destroy(taurow_ring_t *r) { taurow_ring_destroy(r); }
extern "C" int taulap_ring_step_async(taurow_ring_t *r, int npasses) { return taurow_ring_step_async(r, npasses); }
extern "C" int taulap_ring_exchange_async(taurow_ring_t *r) { return taurow_ring_exchange_async(r); }
extern "C" int taulap_ring_finish(taurow_ring_t *r) { return taurow_ring_finish(r); }
extern "C" int taulap_ring_barrier(taurow_ring_t *r) { return taurow_ring_barrier(r); }
extern "C" void taulap_ring_destroy(taurow_ring_t *r) { taurow_ring_destroy(r); }
