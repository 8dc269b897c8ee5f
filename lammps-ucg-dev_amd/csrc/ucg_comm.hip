// ucg_comm.hip -- decomposed runs behind the C ABI: the communicator and the rank-level Verlet step loop.
//
// What it replaces: upstream CommBrick::exchange / borders / forward_comm driven by the field lists of
// UCG/atom_vec_ucg.cpp:66-82, Neighbor::decide's MPI_Allreduce of the "a bead moved too far" flag, the mid-compute
// forward_comm of table_ucg_bethe_density (UCG/pair_table_ucg_bethe_density.cpp:280, a no-op as shipped) and the
// MPI_Allreduce steps of fix cluster_switch (UCG/fix_cluster_switch.cpp:114-120, 157-158, 587, 664, 750).
//
// One process per GPU.  The device work (count / pack / unpack, rebuild, forces, fixes) is the single-rank
// library's; this file only sequences it and moves the packed buffers through a communicator:
//   * built in: RCCL, called directly (ncclGroupStart; one ncclSend + ncclRecv per peer with bytes to move;
//     ncclGroupEnd on the context's stream) -- with 2 bricks per periodic dimension the 26 neighbour directions
//     fold onto the 7 other ranks, i.e. exactly the point-to-point xGMI links.  librccl is loaded at run time
//     (dlopen), so the library does not depend on it for single-GPU use;
//   * or caller-provided callbacks (ucg_comm_ops): MPI in a LAMMPS build, gloo in the tests that put two ranks
//     on one GPU (RCCL refuses two ranks per device).
// Small host-side reductions (per-peer counts at a re-neighbouring, the decision flag every `every` steps, the
// per-molecule arrays of fix cluster_switch) go through the same communicator.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ucg_hip.h"
#include "ucg_ctx.h"

namespace ucg {

// the few RCCL entry points used, resolved with dlsym (types as in rccl.h; ncclComm_t is an opaque pointer)
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *id) = nullptr;
  int (*CommInitRank)(void **comm, int nranks, ucg_rccl_id id, int rank) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
  int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
  int (*AllReduce)(const void *s, void *r, size_t count, int dtype, int op, void *comm, hipStream_t st) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(void *comm, int *count) = nullptr;
  int (*CommCuDevice)(void *comm, int *device) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool loaded = false;  // every symbol resolved (a half-loaded library must not be called through)
  bool load()
  {
    if (loaded) return true;
    if (lib) {
      dlclose(lib);
      lib = nullptr;
    }
    // UCG_RCCL_LIBRARY: a site's own build of RCCL (or, in the tests, a name that does not exist)
    if (const char *own = getenv("UCG_RCCL_LIBRARY")) {
      lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
      if (!lib) {
        err = std::string("UCG_RCCL_LIBRARY=") + own + " could not be loaded";
        return false;
      }
    } else {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
      }
    }
    if (!lib) {
      err = "librccl.so.1 could not be loaded";
      return false;
    }
#define UCG_SYM(field, sym)                                         \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));      \
  if (!field) {                                                     \
    err = std::string("librccl lacks ") + sym;                      \
    dlclose(lib);                                                   \
    lib = nullptr;                                                  \
    return false;                                                   \
  }
    UCG_SYM(GetUniqueId, "ncclGetUniqueId");
    UCG_SYM(CommInitRank, "ncclCommInitRank");
    UCG_SYM(CommDestroy, "ncclCommDestroy");
    UCG_SYM(Send, "ncclSend");
    UCG_SYM(Recv, "ncclRecv");
    UCG_SYM(AllReduce, "ncclAllReduce");
    UCG_SYM(GroupStart, "ncclGroupStart");
    UCG_SYM(GroupEnd, "ncclGroupEnd");
    UCG_SYM(CommCount, "ncclCommCount");
    UCG_SYM(CommCuDevice, "ncclCommCuDevice");
    UCG_SYM(GetErrorString, "ncclGetErrorString");
#undef UCG_SYM
    loaded = true;
    return true;
  }
};
static Rccl g_rccl;
// rccl.h: ncclInt8 = 0 (ncclChar), ncclInt64 = 4, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2, ncclMin = 3
enum { NCCL_CHAR = 0, NCCL_INT64 = 4, NCCL_F64 = 8, NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

struct CommState {
  bool attached = false, rccl = false;
  int rank = 0, world = 1;
  ucg_comm_ops ops{};
  void *nccl = nullptr;
  DevBuf<char> send, recv, auxsend, auxrecv;
  DevBuf<long long> dsmall;  // device staging of small host messages on the RCCL transport
  std::vector<long long> halo_send, halo_recv;  // per-peer ghost counts of the last border exchange
  long long nsend = 0, nrecv = 0;
  long long nrebuild = 0;
  bool cluster_synced = false;
  // UCG_RCCL_SELF_SEND=1 (read once, at ucg_comm_attach_rccl): keep the block a rank sends to itself on RCCL instead of a
  // device copy -- the one-rank tests run the grouped send / receive path that way
  bool self_copy = true;
  // ucg_comm_attach_host: the callbacks' alltoallv works on HOST memory; messages are staged through these pinned buffers
  bool host_staged = false;
  char *hsend = nullptr, *hrecv = nullptr;
  size_t hsend_cap = 0, hrecv_cap = 0;
  ~CommState()
  {
    if (hsend) (void) hipHostFree(hsend);
    if (hrecv) (void) hipHostFree(hrecv);
  }
};

void comm_destroy(ucg_ctx *ctx)
{
  if (!ctx->comm) return;
  if (ctx->comm->nccl && g_rccl.loaded) (void) g_rccl.CommDestroy(ctx->comm->nccl);
  delete ctx->comm;
  ctx->comm = nullptr;
}

namespace {

struct CommFailure {
  std::string msg;
};

void nccl_check(int rc, const char *what)
{
  if (rc != 0) throw CommFailure{std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?")};
}

void cb_check(int rc, const char *what)
{
  if (rc != 0) throw CommFailure{std::string("communicator callback failed: ") + what};
}

// device buffers, `sendbytes[r]` bytes to rank r taken from consecutive blocks of `send` (rank order), received
// blocks land in `recv` in rank order; ordered on the context's stream
void alltoallv(ucg_ctx *ctx, const void *send, const long long *sendbytes, void *recv, const long long *recvbytes)
{
  CommState &C = *ctx->comm;
  if (!C.rccl && C.host_staged) {
    size_t ns = 0, nr = 0;
    for (int r = 0; r < C.world; r++) {
      ns += (size_t) sendbytes[r];
      nr += (size_t) recvbytes[r];
    }
    auto grow = [](char *&p, size_t &cap, size_t want) {
      if (want <= cap) return;
      if (p) (void) hipHostFree(p);
      p = nullptr;
      cap = want + want / 4 + 4096;
      UCG_HIP(hipHostMalloc((void **) &p, cap, hipHostMallocDefault));
    };
    grow(C.hsend, C.hsend_cap, ns);
    grow(C.hrecv, C.hrecv_cap, nr);
    if (ns) UCG_HIP(hipMemcpyAsync(C.hsend, send, ns, hipMemcpyDeviceToHost, ctx->stream));
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    cb_check(C.ops.alltoallv(C.ops.user, C.hsend, sendbytes, C.hrecv, recvbytes, nullptr), "alltoallv");
    if (nr) UCG_HIP(hipMemcpyAsync(recv, C.hrecv, nr, hipMemcpyHostToDevice, ctx->stream));
    return;
  }
  if (!C.rccl) {
    cb_check(C.ops.alltoallv(C.ops.user, send, sendbytes, recv, recvbytes, (void *) ctx->stream), "alltoallv");
    return;
  }
  // the block a rank sends to itself (its own periodic images: every grid with a dimension of one rank) never leaves the
  // device: a plain copy on the stream instead of an ncclSend / ncclRecv pair to self
  // (CommState::self_copy = false keeps the self block on RCCL: the one-rank tests run the grouped send / receive path that way)
  const bool self_copy = C.self_copy;
  long long so = 0, ro = 0;
  bool peers = false;
  for (int r = 0; r < C.world; r++) {
    if (r == C.rank && self_copy) {
      if (sendbytes[r] != recvbytes[r]) throw CommFailure{"alltoallv: a rank's block to itself has two sizes"};
      if (sendbytes[r] > 0)
        UCG_HIP(hipMemcpyAsync((char *) recv + ro, (const char *) send + so, (size_t) sendbytes[r], hipMemcpyDeviceToDevice, ctx->stream));
    } else if (sendbytes[r] > 0 || recvbytes[r] > 0) {
      peers = true;
    }
    so += sendbytes[r];
    ro += recvbytes[r];
  }
  if (!peers) return;
  nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  so = ro = 0;
  for (int r = 0; r < C.world; r++) {
    if (r != C.rank || !self_copy) {
      if (sendbytes[r] > 0)
        nccl_check(g_rccl.Send((const char *) send + so, (size_t) sendbytes[r], NCCL_CHAR, r, C.nccl, ctx->stream), "ncclSend");
      if (recvbytes[r] > 0)
        nccl_check(g_rccl.Recv((char *) recv + ro, (size_t) recvbytes[r], NCCL_CHAR, r, C.nccl, ctx->stream), "ncclRecv");
    }
    so += sendbytes[r];
    ro += recvbytes[r];
  }
  nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

// the RCCL all-to-all with the block of rank `gap_rank` (this rank's own) left out of the transfer but kept as a hole of
// `gap_bytes` in both buffers, so that every other block stays where the ghost permutation expects it
void alltoallv_gap(ucg_ctx *ctx, const void *send, const long long *sendbytes, void *recv, const long long *recvbytes, int gap_rank,
                   long long gap_bytes)
{
  CommState &C = *ctx->comm;
  long long so = 0, ro = 0;
  bool peers = false;
  for (int r = 0; r < C.world; r++) peers = peers || (r != gap_rank && (sendbytes[r] > 0 || recvbytes[r] > 0));
  if (!peers) return;
  nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  for (int r = 0; r < C.world; r++) {
    if (r == gap_rank) {
      so += gap_bytes;
      ro += gap_bytes;
      continue;
    }
    if (sendbytes[r] > 0)
      nccl_check(g_rccl.Send((const char *) send + so, (size_t) sendbytes[r], NCCL_CHAR, r, C.nccl, ctx->stream), "ncclSend");
    if (recvbytes[r] > 0)
      nccl_check(g_rccl.Recv((char *) recv + ro, (size_t) recvbytes[r], NCCL_CHAR, r, C.nccl, ctx->stream), "ncclRecv");
    so += sendbytes[r];
    ro += recvbytes[r];
  }
  nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
}

// RCCL transport, host arrays, in the wire format of alltoall_counts_dev (4 bytes per peer each way): what a rank uses whose
// counting kernel did not run (it holds a failure and sends zeros) while its peers exchange device counters
void alltoall_counts32(ucg_ctx *ctx, const long long *send, long long *recv)
{
  CommState &C = *ctx->comm;
  const size_t w = (size_t) C.world;
  C.dsmall.reserve(w + 8);
  int *d = reinterpret_cast<int *>(C.dsmall.get());
  std::vector<int> h(2 * w, 0);
  for (size_t r = 0; r < w; r++) h[r] = (int) send[r];
  UCG_HIP(hipMemcpyAsync(d, h.data(), w * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  for (int r = 0; r < C.world; r++) {
    nccl_check(g_rccl.Send(d + r, 4, NCCL_CHAR, r, C.nccl, ctx->stream), "ncclSend");
    nccl_check(g_rccl.Recv(d + w + r, 4, NCCL_CHAR, r, C.nccl, ctx->stream), "ncclRecv");
  }
  nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
  UCG_HIP(hipMemcpyAsync(h.data() + w, d + w, w * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t r = 0; r < w; r++) recv[r] = h[w + r];
}

// one long long per peer, host arrays, blocking
void alltoall_counts(ucg_ctx *ctx, const long long *send, long long *recv)
{
  CommState &C = *ctx->comm;
  if (!C.rccl) {
    cb_check(C.ops.alltoall_ll(C.ops.user, send, recv), "alltoall_ll");
    return;
  }
  const size_t w = (size_t) C.world;
  C.dsmall.reserve(2 * w + 8);
  UCG_HIP(hipMemcpyAsync(C.dsmall.get(), send, w * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  for (int r = 0; r < C.world; r++) {
    nccl_check(g_rccl.Send(C.dsmall.get() + r, 1, NCCL_INT64, r, C.nccl, ctx->stream), "ncclSend");
    nccl_check(g_rccl.Recv(C.dsmall.get() + w + r, 1, NCCL_INT64, r, C.nccl, ctx->stream), "ncclRecv");
  }
  nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
  UCG_HIP(hipMemcpyAsync(recv, C.dsmall.get() + w, w * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
}

// RCCL transport: the per-destination counters (world ints on the device, fresh from the counting kernel) go straight into
// a grouped 4-byte send / receive per peer; the host gets its send AND receive counts with one synchronisation (the
// host-array form above costs two: the counters' download, then the exchanged counts')
void alltoall_counts_dev(ucg_ctx *ctx, const int *dev_send, long long *send, long long *recv)
{
  CommState &C = *ctx->comm;
  const size_t w = (size_t) C.world;
  C.dsmall.reserve(w + 8);
  int *drecv = reinterpret_cast<int *>(C.dsmall.get());
  nccl_check(g_rccl.GroupStart(), "ncclGroupStart");
  for (int r = 0; r < C.world; r++) {
    nccl_check(g_rccl.Send(dev_send + r, 4, NCCL_CHAR, r, C.nccl, ctx->stream), "ncclSend");
    nccl_check(g_rccl.Recv(drecv + r, 4, NCCL_CHAR, r, C.nccl, ctx->stream), "ncclRecv");
  }
  nccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
  std::vector<int> h(2 * w);
  UCG_HIP(hipMemcpyAsync(h.data(), dev_send, w * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipMemcpyAsync(h.data() + w, drecv, w * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t r = 0; r < w; r++) {
    send[r] = h[r];
    recv[r] = h[w + r];
  }
}

// op: 0 sum, 1 max, 2 min
void allreduce_ll(ucg_ctx *ctx, long long *buf, int n, int op)
{
  CommState &C = *ctx->comm;
  if (n <= 0) return;
  if (!C.rccl) {
    cb_check(C.ops.allreduce_ll(C.ops.user, buf, n, op), "allreduce_ll");
    return;
  }
  C.dsmall.reserve((size_t) n + 8);
  UCG_HIP(hipMemcpyAsync(C.dsmall.get(), buf, (size_t) n * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  nccl_check(g_rccl.AllReduce(C.dsmall.get(), C.dsmall.get(), (size_t) n, NCCL_INT64, op == 0 ? NCCL_SUM : op == 1 ? NCCL_MAX : NCCL_MIN,
                              C.nccl, ctx->stream),
             "ncclAllReduce");
  UCG_HIP(hipMemcpyAsync(buf, C.dsmall.get(), (size_t) n * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
}

void allreduce_f64(ucg_ctx *ctx, double *buf, int n, int op)
{
  CommState &C = *ctx->comm;
  if (n <= 0) return;
  if (!C.rccl) {
    cb_check(C.ops.allreduce_f64(C.ops.user, buf, n, op), "allreduce_f64");
    return;
  }
  C.dsmall.reserve((size_t) n + 8);
  UCG_HIP(hipMemcpyAsync(C.dsmall.get(), buf, (size_t) n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  nccl_check(g_rccl.AllReduce(C.dsmall.get(), C.dsmall.get(), (size_t) n, NCCL_F64, op == 0 ? NCCL_SUM : op == 1 ? NCCL_MAX : NCCL_MIN,
                              C.nccl, ctx->stream),
             "ncclAllReduce");
  UCG_HIP(hipMemcpyAsync(buf, C.dsmall.get(), (size_t) n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
}

void allreduce_int_array(ucg_ctx *ctx, std::vector<int> &a, int op)
{
  std::vector<long long> t(a.begin(), a.end());
  allreduce_ll(ctx, t.data(), (int) t.size(), op);
  for (size_t i = 0; i < a.size(); i++) a[i] = (int) t[i];
}

#define UCG_RC(call)            \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != UCG_OK) return rc__; \
  } while (0)

long long sum(const std::vector<long long> &v)
{
  long long s = 0;
  for (long long x : v) s += x;
  return s;
}

// A rank-local failure between two collectives must not leave the peers blocked in the next ncclRecv: the failing rank
// keeps taking part -- with empty messages -- until the next point where the ranks agree on a status (an all-reduce of
// the worst local code), and then every rank returns an error together.  `bad` holds the first local failure.
struct LocalStatus {
  int bad = UCG_OK;
  std::string msg;
  template <typename F>
  void run(ucg_ctx *ctx, F &&fn)
  {
    if (bad != UCG_OK) return;
    int rc;
    try {
      rc = fn();
    } catch (const InputError &e) {
      ctx->err = e.msg;
      rc = UCG_ERR_INPUT;
    } catch (const HipFailure &e) {
      ctx->err = std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what;
      rc = UCG_ERR_HIP;
    }  // (a CommFailure is not rank-local: it propagates)
    if (rc != UCG_OK) {
      bad = rc;
      msg = ctx->err;
    }
  }
  void fail(int code, const std::string &why)
  {
    if (bad != UCG_OK) return;
    bad = code;
    msg = why;
  }
  // all ranks: the worst code; a rank that was fine itself reports UCG_ERR_COMM-style text
  int agree(ucg_ctx *ctx, const char *where)
  {
    long long worst = bad;
    allreduce_ll(ctx, &worst, 1, 1);
    if (worst == UCG_OK) return UCG_OK;
    if (bad != UCG_OK) ctx->err = msg;
    else ctx->err = std::string("another rank failed ") + where + " (every rank stops; the job must be aborted)";
    return (int) worst;
  }
};

// CommBrick::exchange + borders: every bead to the rank that owns its wrapped position, then the images every
// rank's extended brick needs; afterwards bins and rows are rebuilt (ucg_border_unpack).  Ends with a status agreement.
// `agree` = false (the step loop): no all-reduce of its own at the end -- a local failure is handed back in *deferred and
// the ranks agree on it at the next re-neighbour decision, whose all-reduce carries every rank's status anyway (at most
// `every` steps later; until then the failing rank keeps taking part in the halos with stale data and runs nothing local)
int multi_rebuild(ucg_ctx *ctx, const LocalStatus *held = nullptr, bool agree = true, LocalStatus *deferred = nullptr)
{
  CommState &C = *ctx->comm;
  const size_t w = (size_t) C.world;
  int arec = 0, hrec = 0;
  ucg_record_bytes(&arec, &hrec);
  std::vector<long long> sc(w, 0), rc(w, 0), sb(w, 0), rb(w, 0);
  LocalStatus st;
  if (held && held->bad != UCG_OK) st = *held;  // a rank that has already failed sends nothing and reports at the agreement
  auto zero = [&]() {
    if (st.bad != UCG_OK) std::fill(sc.begin(), sc.end(), 0LL);
  };
  // counts: on the RCCL transport exchanged on the device (one synchronisation); a rank that has already failed takes
  // the host form with zeros, which its peers' device form matches message by message (4 bytes each way per peer)
  const int *dcounts = nullptr;
  auto exchange_counts = [&](int which) {
    if (C.rccl && st.bad == UCG_OK) {
      st.run(ctx, [&] { return which == 0 ? exchange_count_launch(ctx, &dcounts) : border_count_launch(ctx, &dcounts); });
      if (st.bad == UCG_OK) {
        alltoall_counts_dev(ctx, dcounts, sc.data(), rc.data());
        st.run(ctx, [&] { return counts_adopt(ctx, which, sc.data()); });
        return;
      }
    } else {
      st.run(ctx, [&] { return which == 0 ? ucg_exchange_count(ctx, sc.data()) : ucg_border_count(ctx, sc.data()); });
    }
    zero();
    if (C.rccl) alltoall_counts32(ctx, sc.data(), rc.data());
    else alltoall_counts(ctx, sc.data(), rc.data());
  };
  exchange_counts(0);
  for (size_t r = 0; r < w; r++) {
    sb[r] = sc[r] * arec;
    rb[r] = rc[r] * arec;
  }
  C.send.reserve((size_t) sum(sb) + 16);
  C.recv.reserve((size_t) sum(rb) + 16);
  st.run(ctx, [&] { return ucg_exchange_pack(ctx, C.send.get()); });
  alltoallv(ctx, C.send.get(), sb.data(), C.recv.get(), rb.data());
  st.run(ctx, [&] { return ucg_exchange_unpack(ctx, C.recv.get(), sum(rc)); });
  exchange_counts(1);
  for (size_t r = 0; r < w; r++) {
    sb[r] = sc[r] * hrec;
    rb[r] = rc[r] * hrec;
  }
  C.send.reserve((size_t) sum(sb) + 16);
  C.recv.reserve((size_t) sum(rb) + 16);
  st.run(ctx, [&] { return ucg_border_pack(ctx, C.send.get()); });
  alltoallv(ctx, C.send.get(), sb.data(), C.recv.get(), rb.data());
  st.run(ctx, [&] { return ucg_border_unpack(ctx, C.recv.get(), sum(rc)); });
  C.halo_send = sc;
  C.halo_recv = rc;
  C.nsend = sum(sc);
  C.nrecv = sum(rc);
  if (ctx->cs) {  // ghosts' group bits and molecule ids, for fix cluster_switch
    for (size_t r = 0; r < w; r++) {
      sb[r] = sc[r] * 8;
      rb[r] = rc[r] * 8;
    }
    C.auxsend.reserve((size_t) C.nsend * 8 + 16);
    C.auxrecv.reserve((size_t) C.nrecv * 8 + 16);
    st.run(ctx, [&] { return ucg_halo_molmask_pack(ctx, C.auxsend.get()); });
    alltoallv(ctx, C.auxsend.get(), sb.data(), C.auxrecv.get(), rb.data());
    st.run(ctx, [&] { return ucg_halo_molmask_unpack(ctx, C.auxrecv.get()); });
  }
  C.nrebuild++;
  if (!agree) {
    if (deferred && st.bad != UCG_OK) deferred->fail(st.bad, st.msg);
    return UCG_OK;
  }
  return st.agree(ctx, "during the re-neighbouring");
}

// forward_comm: x (+ shift), lambda, ucgp, state of every ghost from its owner, with the send lists of the last
// border exchange; the unpack of a step is ordered before the next step's transfer on the stream, so one pair of
// buffers serves
int multi_halo_forward(ucg_ctx *ctx, LocalStatus *ls = nullptr)
{
  CommState &C = *ctx->comm;
  const size_t w = (size_t) C.world;
  int arec = 0, hrec = 0;
  ucg_record_bytes(&arec, &hrec);
  std::vector<long long> sb(w), rb(w);
  for (size_t r = 0; r < w; r++) {
    sb[r] = C.halo_send[r] * hrec;
    rb[r] = C.halo_recv[r] * hrec;
  }
  // a rank with a pending local failure (ls->bad) still moves its (stale) buffers: the peers' receives complete
  LocalStatus own;
  LocalStatus &st = ls ? *ls : own;
  if (C.rccl && C.self_copy && C.halo_send[(size_t) C.rank] > 0 && C.halo_send[(size_t) C.rank] == C.halo_recv[(size_t) C.rank]) {
    // the rank's own periodic images (a grid dimension of one rank): made straight from their owner beads by the unpack
    // kernel; only the peers' records are packed and moved -- with one rank the whole halo is one launch
    long long so_self = 0, ro_self = 0;
    for (int r = 0; r < C.rank; r++) {
      so_self += C.halo_send[(size_t) r];
      ro_self += C.halo_recv[(size_t) r];
    }
    const long long nself = C.halo_send[(size_t) C.rank];
    st.run(ctx, [&] { return halo_pack_peers(ctx, C.send.get(), so_self, nself); });
    sb[(size_t) C.rank] = rb[(size_t) C.rank] = 0;
    // (the offsets of the peers' blocks inside the buffers must stay what the permutation expects: the self block's
    // bytes are skipped, not closed up)
    alltoallv_gap(ctx, C.send.get(), sb.data(), C.recv.get(), rb.data(), C.rank, nself * hrec);
    st.run(ctx, [&] { return halo_unpack_self(ctx, C.recv.get(), ro_self, so_self, nself); });
  } else {
    st.run(ctx, [&] { return ucg_halo_pack(ctx, C.send.get()); });
    alltoallv(ctx, C.send.get(), sb.data(), C.recv.get(), rb.data());
    st.run(ctx, [&] { return ucg_halo_unpack(ctx, C.recv.get()); });
  }
  if (!ls && own.bad != UCG_OK) {
    ctx->err = own.msg;
    return own.bad;
  }
  return UCG_OK;
}

// one double2 per ghost from its owner: the density style's priors (which = 0) and CV forces (which = 1)
int multi_aux_halo(ucg_ctx *ctx, ucg_pair *p, int which, LocalStatus &st)
{
  CommState &C = *ctx->comm;
  const size_t w = (size_t) C.world;
  std::vector<long long> sb(w), rb(w);
  for (size_t r = 0; r < w; r++) {
    sb[r] = C.halo_send[r] * 16;
    rb[r] = C.halo_recv[r] * 16;
  }
  C.auxsend.reserve((size_t) C.nsend * 16 + 16);
  C.auxrecv.reserve((size_t) C.nrecv * 16 + 16);
  void *field = ucg_pair_density_buffer(p, which);
  st.run(ctx, [&] { return field ? ucg_halo_aux_pack(ctx, field, C.auxsend.get()) : (int) UCG_ERR_INVALID; });
  alltoallv(ctx, C.auxsend.get(), sb.data(), C.auxrecv.get(), rb.data());
  st.run(ctx, [&] { return ucg_halo_aux_unpack(ctx, field, C.auxrecv.get()); });
  return UCG_OK;
}

int multi_pair_compute(ucg_ctx *ctx, int ev, LocalStatus *ls = nullptr)
{
  ucg_pair *p = ctx->md_pair;
  double e = 0.0, vir[6] = {0, 0, 0, 0, 0, 0};
  LocalStatus own;
  LocalStatus &st = ls ? *ls : own;
  if (p->model.style != STYLE_BETHE_DENSITY) {
    st.run(ctx, [&] { return ucg_pair_compute(p, ev, ev, ev ? &e : nullptr, ev ? vir : nullptr); });
  } else {
    // table_ucg_bethe_density: its two mid-compute halos cross ranks
    st.run(ctx, [&] { return ucg_pair_density_phase(p, 1, ev, ev, nullptr, nullptr); });
    multi_aux_halo(ctx, p, 0, st);
    st.run(ctx, [&] { return ucg_pair_density_phase(p, 2, ev, ev, nullptr, nullptr); });
    multi_aux_halo(ctx, p, 1, st);
    st.run(ctx, [&] { return ucg_pair_density_phase(p, 3, ev, ev, ev ? &e : nullptr, ev ? vir : nullptr); });
  }
  if (ev) {  // thermo output: one all-reduce of {E_pair, virial[6]}; each rank keeps the totals
    double t[7] = {e, vir[0], vir[1], vir[2], vir[3], vir[4], vir[5]};
    allreduce_f64(ctx, t, 7, 0);
    for (int c = 0; c < 7; c++) ctx->thermo[c] = t[c];
  }
  if (!ls && own.bad != UCG_OK) {
    ctx->err = own.msg;
    return own.bad;
  }
  return UCG_OK;
}

// FixClusterSwitch across ranks: the reductions the reference does with MPI_Allreduce.  Local calls run through `st`
// (a rank-local failure is held and the rank keeps taking part in every collective with whatever it has), the
// collectives are unconditional: their number and sizes are the same on every rank.
void multi_cluster_sync_after_create(ucg_ctx *ctx, LocalStatus &st)
{
  long long s[3] = {0, 0, 0};
  st.run(ctx, [&] { return ucg_fix_cluster_switch_scalars(ctx, s); });
  allreduce_ll(ctx, &s[0], 1, 1);
  allreduce_ll(ctx, &s[1], 2, 0);
  st.run(ctx, [&] { return ucg_fix_cluster_switch_set_scalars(ctx, s[0], s[1], s[2]); });
  // maxmol is the all-reduced value on every rank that is still fine; a failed rank takes the peers' size from s[0]
  const int n = (int) s[0] + 1;
  std::vector<int> a((size_t) (n > 0 ? n : 1), 0);
  for (int which : {1, 2, 4}) {  // mol_state, mol_restrict, presence
    st.run(ctx, [&] { return ucg_fix_cluster_switch_array(ctx, which, a.data()); });
    allreduce_int_array(ctx, a, 1);
    st.run(ctx, [&] { return ucg_fix_cluster_switch_set_array(ctx, which, a.data()); });
  }
  ctx->comm->cluster_synced = true;
}

// check_cluster + attempt_switch on fresh lists (UCG/fix_cluster_switch.cpp:452-469)
void multi_cluster_step(ucg_ctx *ctx, LocalStatus &st)
{
  const int n = ucg_fix_cluster_switch_maxmol(ctx) + 1;  // all-reduced at creation: the same on every rank
  std::vector<int> a((size_t) (n > 0 ? n : 1), 0);
  int changed = 0;
  st.run(ctx, [&] { return ucg_fix_cluster_switch_sweep(ctx, 1, &changed); });
  for (;;) {
    st.run(ctx, [&] { return ucg_fix_cluster_switch_array(ctx, 5, a.data()); });
    allreduce_int_array(ctx, a, 2);
    st.run(ctx, [&] { return ucg_fix_cluster_switch_set_array(ctx, 5, a.data()); });
    long long any = st.bad == UCG_OK ? changed : 0;  // a failed rank sweeps no more: the others' labels still converge
    allreduce_ll(ctx, &any, 1, 1);
    if (!any) break;
    changed = 0;
    st.run(ctx, [&] { return ucg_fix_cluster_switch_sweep(ctx, 0, &changed); });
  }
  st.run(ctx, [&] { return ucg_fix_cluster_switch_finalize(ctx); });
  st.run(ctx, [&] { return ucg_fix_cluster_switch_attempt_local(ctx); });
  st.run(ctx, [&] { return ucg_fix_cluster_switch_array(ctx, 3, a.data()); });
  allreduce_int_array(ctx, a, 1);
  st.run(ctx, [&] { return ucg_fix_cluster_switch_set_array(ctx, 3, a.data()); });
  st.run(ctx, [&] { return ucg_fix_cluster_switch_attempt_apply(ctx); });
  st.run(ctx, [&] { return ucg_fix_cluster_switch_advance(ctx); });
  multi_halo_forward(ctx, &st);  // comm->forward_comm(this): the ghosts' new atom types
}

template <typename F>
int guarded_comm(ucg_ctx *ctx, F &&fn)
{
  try {
    return fn();
  } catch (const CommFailure &e) {
    ctx->err = e.msg;
    return UCG_ERR_COMM;
  } catch (const InputError &e) {
    ctx->err = e.msg;
    return UCG_ERR_INPUT;
  } catch (const HipFailure &e) {
    ctx->err = std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what;
    return UCG_ERR_HIP;
  } catch (const std::exception &e) {
    ctx->err = e.what();
    return UCG_ERR_INVALID;
  }
}

int poll_pair_errors(ucg_ctx *ctx, LocalStatus *ls = nullptr)
{
  // the sticky table-range flag of the pair kernels and any pending rank-local failure; every rank takes the same decision
  int rc = (ls && ls->bad != UCG_OK) ? ls->bad : ucg_pair_check_errors(ctx->md_pair);
  if (ls && ls->bad != UCG_OK) ctx->err = ls->msg;
  long long worst = rc;
  allreduce_ll(ctx, &worst, 1, 1);
  if (rc == UCG_OK && worst != 0) {
    ctx->err = "another rank reported an error (a pair-table range error or a failed call; every rank stops, the job must be aborted)";
    return (int) worst;
  }
  return rc;
}

}  // namespace

// ---- the step loop of one rank (upstream Verlet::setup / run), called by ucg_md_setup / ucg_md_run
int md_setup_multi(ucg_ctx *ctx)
{
  if (ctx->dom_world != ctx->comm->world) {
    ctx->err = "the processor grid of ucg_decomp_set and the attached communicator have different rank counts";
    return UCG_ERR_INVALID;
  }
  return guarded_comm(ctx, [&]() -> int {
    // Every local call runs through `ls`: a rank that fails anywhere in the setup keeps taking part in the collectives
    // that follow (the halos of the density style, the thermo all-reduce) and all ranks return together from the final
    // agreement (include/ucg_hip.h, "communicator of a decomposed run").
    LocalStatus ls;
    if (ctx->fault_setup) ls.fail(UCG_ERR_INVALID, "injected rank-local failure in the setup (option fault_inject_setup)");
    if (ctx->cs && !ctx->comm->cluster_synced) {
      multi_cluster_sync_after_create(ctx, ls);
      UCG_RC(ls.agree(ctx, "in the survey of fix cluster_switch"));
    }
    {
      // table_ucg_bethe: beads arrive from other ranks in the exchange below; one rank's "a first-call marker may be around"
      // is every rank's (uploads differ per rank; the fixes that lower the flag run in lockstep)
      long long fp = ctx->ucgp_first_possible ? 1 : 0;
      allreduce_ll(ctx, &fp, 1, 1);
      ctx->ucgp_first_possible = fp != 0;
    }
    UCG_RC(multi_rebuild(ctx, &ls));  // (ends with an agreement of its own)
    if (ctx->md_lang && !ctx->lang.inited) {
      // Fix_UCGLD_Langevin::init(): reads atom->ucgml[1..ntypes] of the LOCAL bead order (App. B #5)
      ls.run(ctx, [&]() -> int {
        std::vector<double> ml((size_t) ctx->ntypes + 1, 0.0);
        const int cnt = ctx->ntypes + 1 <= ctx->nlocal ? ctx->ntypes + 1 : ctx->nlocal;
        if (cnt > 0) {
          UCG_HIP(hipMemcpyAsync(ml.data(), ctx->ucgml.get(), (size_t) cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
          UCG_HIP(hipStreamSynchronize(ctx->stream));
        }
        for (int i = cnt; i <= ctx->ntypes; i++) ml[(size_t) i] = cnt > 0 ? ml[0] : 1.0;
        return ucg_fix_langevin_init_from_ucgml(ctx, ctx->ntypes, ml.data());
      });
    }
    multi_pair_compute(ctx, 1, &ls);
    if (ctx->md_lang)
      ls.run(ctx, [&] { return ucg_fix_langevin_post_force(ctx, ctx->groupbit, ctx->ntimestep, ctx->beginstep, ctx->endstep); });
    if (ctx->md_ucgst) ls.run(ctx, [&] { return ucg_fix_ucgstate_post_force(ctx); });
    return poll_pair_errors(ctx, &ls);
  });
}

int md_run_multi(ucg_ctx *ctx, long long nsteps, int thermo_every, int ev_on_last)
{
  return guarded_comm(ctx, [&]() -> int {
    ucg_pair *p = ctx->md_pair;
    const bool density = p->model.style == STYLE_BETHE_DENSITY;
    bool initial_done = false;
    // a rank-local failure inside a step is held (`ls`) and this rank goes on taking part in the collectives until the
    // next status agreement -- the re-neighbour decision's all-reduce, an error poll, the re-neighbouring itself -- where
    // every rank returns together; nothing local runs on this rank after the failure
    LocalStatus ls;
    for (long long s = 0; s < nsteps; s++) {
      ctx->ntimestep++;
      if (ctx->fault_step == ctx->ntimestep) ls.fail(UCG_ERR_INVALID, "injected rank-local failure (option fault_inject_step)");
      const int ev = ((thermo_every > 0 && (ctx->ntimestep % thermo_every == 0)) || (ev_on_last && s + 1 == nsteps)) ? 1 : 0;
      if (ctx->md_nve && !initial_done)
        ls.run(ctx, [&] { return ctx->md_nve == 2 ? ucg_fix_nve_wall_hard_initial(ctx, ctx->groupbit) : ucg_fix_nve_initial(ctx, ctx->groupbit); });
      int due = 0, flag = 0, pair_flag = -1;
      const bool fuse_next = ctx->md_nve && !ev && (s + 1 < nsteps) && !ctx->md_no_fuse;
      bool rebuilt = false;
      if (ctx->comm->rccl) {
        // RCCL transport: {a bead moved too far, this rank's status, the pair kernels' error flag} are all-reduced where the
        // decision's kernels leave them -- on the device -- and come back with ONE download (the host-array form below
        // costs two synchronisations: the flags' download, then the all-reduce's)
        CommState &C = *ctx->comm;
        C.dsmall.reserve(16);
        int checked = 0;
        UCG_RC(decide_launch(ctx, &due, &checked, C.dsmall.get()));  // (the schedule must stay in step on every rank: a failure here is fatal)
        if (due) {
          long long h[3] = {1, ls.bad, 1};  // no distance check: the step re-neighbours; the flag is unknown: poll
          if (checked) UCG_HIP(hipMemcpyAsync(C.dsmall.get() + 2, &h[1], sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
          else UCG_HIP(hipMemcpyAsync(C.dsmall.get(), h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
          if (checked) {
            // the kernel wrote {moved, pair flag}; the wire order is {moved, status, pair flag}: status goes to slot 2 and the
            // two are swapped on the host after the download
          }
          nccl_check(g_rccl.AllReduce(C.dsmall.get(), C.dsmall.get(), 3, NCCL_INT64, NCCL_MAX, C.nccl, ctx->stream), "ncclAllReduce");
          long long f[3];
          UCG_HIP(hipMemcpyAsync(f, C.dsmall.get(), sizeof f, hipMemcpyDeviceToHost, ctx->stream));
          UCG_HIP(hipStreamSynchronize(ctx->stream));
          const long long moved = f[0], status = checked ? f[2] : f[1], pairf = checked ? f[1] : f[2];
          if (status != UCG_OK) {
            ctx->err = ls.bad != UCG_OK ? ls.msg : std::string("another rank failed in the step loop (every rank stops; the job must be aborted)");
            return (int) status;
          }
          if (pairf != 0) UCG_RC(poll_pair_errors(ctx, &ls));
          rebuilt = moved != 0;
        }
        due = 0;  // handled
      } else {
        UCG_RC(decide_local_impl(ctx, &due, &flag, &pair_flag));
      }
      if (due) {
        // Neighbor::decide(): MPI_Allreduce of the flag -- and, in the same message, of the ranks' status since the last
        // agreement and of the pair kernels' sticky table-range flag (read by the decision's own download: -1 = the
        // distance check did not run, so the poll below has to)
        long long f[3] = {flag, ls.bad, pair_flag != 0};
        allreduce_ll(ctx, f, 3, 1);
        if (f[1] != UCG_OK) {
          ctx->err = ls.bad != UCG_OK ? ls.msg : std::string("another rank failed in the step loop (every rank stops; the job must be aborted)");
          return (int) f[1];
        }
        if (f[2] != 0) UCG_RC(poll_pair_errors(ctx, &ls));  // some rank's flag is up (or unknown): every rank takes part in the poll
        rebuilt = f[0] != 0;
      }
      if (rebuilt) {
        UCG_RC(multi_rebuild(ctx, nullptr, false, &ls));
        if (ctx->cs) {
          int forced = 0, switching = 0;
          UCG_RC(ucg_fix_cluster_switch_due(ctx, &forced, &switching));  // (a function of the timestep alone: in step on every rank)
          if (switching) {
            multi_cluster_step(ctx, ls);
            UCG_RC(poll_pair_errors(ctx, &ls));  // agreement: a rank that failed inside the switch stops everyone here
          }
        }
      } else {
        multi_halo_forward(ctx, &ls);
      }
      // pair force, then langevin -> ucgstate -> final_integrate (-> next initial_integrate): one launch (the gather
      // kernel's epilogue) where that applies, else two
      int rc = UCG_ERR_UNSUPPORTED;
      if (fuse_next && !density && ls.bad == UCG_OK) {
        rc = ucg_md_pair_post(ctx, p, ctx->md_lang, ctx->md_ucgst, ctx->md_nve, ctx->groupbit, ctx->ntimestep, ctx->beginstep,
                              ctx->endstep);
        if (rc != UCG_OK && rc != UCG_ERR_UNSUPPORTED) {
          ls.bad = rc;
          ls.msg = ctx->err;
        }
      }
      if (rc == UCG_ERR_UNSUPPORTED) {
        multi_pair_compute(ctx, ev, &ls);
        ls.run(ctx, [&] { return ucg_md_post_fused(ctx, ctx->md_lang, ctx->md_ucgst, ctx->md_nve, fuse_next, ctx->groupbit, ctx->ntimestep,
                                                   ctx->beginstep, ctx->endstep); });
      }
      initial_done = fuse_next;
      if (ev && ctx->md_lang) ls.run(ctx, [&] { return ucg_fix_langevin_end_of_step(ctx, ctx->groupbit, nullptr); });
      if (ev) UCG_RC(poll_pair_errors(ctx, &ls));
    }
    return poll_pair_errors(ctx, &ls);
  });
}

}  // namespace ucg

using namespace ucg;

extern "C" {

int ucg_comm_attach(ucg_ctx *ctx, const ucg_comm_ops *ops)
{
  if (!ctx || !ops || !ops->alltoallv || !ops->alltoall_ll || !ops->allreduce_ll || !ops->allreduce_f64 || ops->world < 1 ||
      ops->rank < 0 || ops->rank >= ops->world)
    return UCG_ERR_INVALID;
  comm_destroy(ctx);
  ctx->comm = new CommState();
  ctx->comm->attached = true;
  ctx->comm->ops = *ops;
  ctx->comm->rank = ops->rank;
  ctx->comm->world = ops->world;
  return UCG_OK;
}

int ucg_comm_attach_host(ucg_ctx *ctx, const ucg_comm_ops *ops)
{
  const int rc = ucg_comm_attach(ctx, ops);
  if (rc == UCG_OK) ctx->comm->host_staged = true;
  return rc;
}

int ucg_comm_transport(const ucg_ctx *ctx, int *out4)
{
  if (!ctx || !ctx->comm || !out4) return UCG_ERR_INVALID;
  const CommState &C = *ctx->comm;
  out4[0] = C.rccl ? 1 : 0;
  out4[1] = 0;
  out4[2] = -1;
  out4[3] = C.host_staged ? 1 : 0;
  if (C.rccl && C.nccl && g_rccl.loaded) {
    // asked of RCCL itself, not of what the caller passed to ucg_comm_attach_rccl
    if (g_rccl.CommCount(C.nccl, &out4[1]) != 0 || g_rccl.CommCuDevice(C.nccl, &out4[2]) != 0) return UCG_ERR_COMM;
  }
  return UCG_OK;
}

int ucg_comm_rccl_unique_id(ucg_rccl_id *out)
{
  if (!out) return UCG_ERR_INVALID;
  if (!g_rccl.load()) return UCG_ERR_COMM;
  return g_rccl.GetUniqueId(out) == 0 ? UCG_OK : UCG_ERR_COMM;
}

int ucg_comm_attach_rccl(ucg_ctx *ctx, const ucg_rccl_id *id, int rank, int world)
{
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return UCG_ERR_INVALID;
  if (!g_rccl.load()) {
    ctx->err = g_rccl.err;
    return UCG_ERR_COMM;
  }
  comm_destroy(ctx);
  void *comm = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) {
    ctx->err = "hipSetDevice failed";
    return UCG_ERR_HIP;
  }
  const int rc = g_rccl.CommInitRank(&comm, world, *id, rank);
  if (rc != 0) {
    ctx->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc);
    return UCG_ERR_COMM;
  }
  ctx->comm = new CommState();
  ctx->comm->attached = true;
  ctx->comm->rccl = true;
  const char *selfenv = getenv("UCG_RCCL_SELF_SEND");
  ctx->comm->self_copy = !(selfenv && selfenv[0] == '1');
  ctx->comm->nccl = comm;
  ctx->comm->rank = rank;
  ctx->comm->world = world;
  return UCG_OK;
}

int ucg_comm_detach(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  comm_destroy(ctx);
  return UCG_OK;
}

int ucg_comm_info(const ucg_ctx *ctx, int *rank, int *world, int *is_rccl, long long *nrebuild)
{
  if (!ctx || !ctx->comm) return UCG_ERR_INVALID;
  if (rank) *rank = ctx->comm->rank;
  if (world) *world = ctx->comm->world;
  if (is_rccl) *is_rccl = ctx->comm->rccl ? 1 : 0;
  if (nrebuild) *nrebuild = ctx->comm->nrebuild;
  return UCG_OK;
}

/* sum / max / min of small host arrays over the ranks of the attached communicator (thermo output of a caller) */
int ucg_comm_allreduce_f64(ucg_ctx *ctx, double *buf, int n, int op)
{
  if (!ctx || !ctx->comm || !buf || n < 0 || op < 0 || op > 2) return UCG_ERR_INVALID;
  return guarded_comm(ctx, [&]() -> int {
    allreduce_f64(ctx, buf, n, op);
    return UCG_OK;
  });
}

}  // extern "C"
