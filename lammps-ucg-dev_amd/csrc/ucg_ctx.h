// ucg_ctx.h -- internal state behind the opaque ucg_ctx / ucg_pair handles.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "ucg_dev.h"
#include "ucg_launch.h"
#include "ucg_model.h"

namespace ucg {

struct HipFailure {
  hipError_t code;
  const char *what;
};

#define UCG_HIP(expr)                                       \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) throw ucg::HipFailure{e__, #expr}; \
  } while (0)

template <typename T>
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release()
  {
    if (p_) (void) hipFree(p_);
    p_ = nullptr;
    cap_ = 0;
  }
  // grow (never shrink); contents are NOT preserved unless keep is set
  void reserve(size_t n, bool keep = false, hipStream_t st = nullptr)
  {
    if (n <= cap_) return;
    size_t want = n + n / 8 + 64;
    T *q = nullptr;
    UCG_HIP(hipMalloc((void **) &q, want * sizeof(T)));
    if (keep && p_ && cap_) {
      UCG_HIP(hipMemcpyAsync(q, p_, cap_ * sizeof(T), hipMemcpyDeviceToDevice, st));
      UCG_HIP(hipStreamSynchronize(st));
    }
    if (p_) (void) hipFree(p_);
    p_ = q;
    cap_ = want;
  }
  T *get() const { return p_; }
  // exactly n elements (no growth slack), contents dropped: for buffers that trade places with another one
  void reserve_exact(size_t n)
  {
    if (n == cap_) return;
    T *q = nullptr;
    UCG_HIP(hipMalloc((void **) &q, (n ? n : 1) * sizeof(T)));
    if (p_) (void) hipFree(p_);
    p_ = q;
    cap_ = n;
  }
  void swap(DevBuf &o)
  {
    T *p = p_;
    p_ = o.p_;
    o.p_ = p;
    const size_t c = cap_;
    cap_ = o.cap_;
    o.cap_ = c;
  }
  size_t capacity() const { return cap_; }

 private:
  T *p_ = nullptr;
  size_t cap_ = 0;
};

// a window of pre-generated draws of one RanMars stream: K steps' worth per launch (option "rng_batch")
struct RngBatch {
  DevBuf<unsigned int> hist_save;  // the stream's 97 lag values at the start of the window
  long long count0 = 0;            // ... and its call count there
  long long total = 0, used = 0;   // draws in the window / handed out so far
  int unit = 0;                    // draws per step the window was sized for
};

struct FixLangevin {
  bool active = false;
  double t_start = 0, t_stop = 0, t_period = 0, t_target = 0, tsqrt = 0;
  int seed = 0, ntypes = 0;
  bool inited = false;
  bool bias = false;  // fix_modify temp with a bias-removing compute: post_force_templated<1> (ucg_fix_langevin_set_bias)
  DevBuf<double> gf1, gf2;
  RanMarsDev rng{};
  DevBuf<unsigned int> hist0, hist1, draws;
  RngBatch batch;
  double lambda_temp = 0;
};

struct FixUcgState {
  bool active = false;
  int ld_flag = 0, mc_flag = 0, mc_seed = 0;
  double mc_rate = 0.01;
  RanMarsDev rng{};
  DevBuf<unsigned int> hist0, hist1, draws;
  RngBatch batch;
};

// host mirrors of a drop-in caller (ucg_host.hip): LAMMPS' arrays of the OWNED atoms and which side is ahead, per field
struct HostMirror {
  bool bound = false;
  double *x = nullptr, *v = nullptr, *f = nullptr, *ucgl = nullptr, *ucgvl = nullptr, *ucgp = nullptr, *ucgforce = nullptr,
         *scores = nullptr;
  int *state = nullptr, *nstates = nullptr;
  unsigned dev_newer = 0, host_newer = 0;  // UCG_F_* masks
  long long uploads = 0, downloads = 0;    // synchronisations that moved data
  DevBuf<double> stage;
  DevBuf<int> istage;
};
void mirror_need(ucg_ctx *ctx, unsigned reads);
void mirror_wrote(ucg_ctx *ctx, unsigned writes);

}  // namespace ucg

struct ucg_pair;
namespace ucg {
struct Domain;
struct ClusterSwitch;
struct CommState;
}

struct ucg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  double boltz = 1, ftm2v = 1, mvv2e = 1, dt = 0.005;
  double special_lj[4] = {1, 1, 1, 1};
  bool density_proximity_as_shipped = false;  // option of the same name (App. B #12)
  bool stage_own = true;  // option "stage_own": LDS staging of the workgroup's own beads in k_pair_gather
  int gather_slots = 1;  // option "gather_slots": lanes per bead of the ucgld / bethe gather kernels
  bool force_generic_kernels = false;  // option "generic_kernels": never pick the FAST variants
  bool fma_contract = false;           // option "fma_contract": gather kernels compiled with FMA contraction (not bit-exact)
  int rng_batch = 10;                  // option "rng_batch": steps of per-bead draws generated per k_ranmars launch (1 = one launch per step)
  bool rows_untiled = false;           // option "rows_untiled": build rows with the one-lane-per-bead kernels
  bool rows_sort_r2 = false;           // option "rows_sort_r2": EXPERIMENT, rows ordered by build-time r^2 (not the specification's bits)
  bool pair_vrow = false;              // option "pair_vrow" (default off: measured slower than the full-row kernels, DESIGN.md
                                       // 4.1): the gather styles run on virtual rows where they can (ucg_pair_vrow.hip:
                                       // own-block pairs once, fixed sums)

  // atoms
  int nlocal = 0, nghost = 0, ntypes = 0;
  ucg::DevBuf<double4> pos4, vel4, frc4;
  ucg::DevBuf<double4> pos4_alt;  // second position / meta buffers of the gather kernels' epilogue (PostDev)
  ucg::DevBuf<int> meta_alt;
  ucg::DevBuf<double> ucgp_alt;
  bool post_in_pair = true;       // option "post_in_pair"
  bool hot_block = true;          // option "hot_block": PairDev::hot_type
  ucg::DevBuf<double2> scores;
  ucg::DevBuf<int> meta, tag, mask, num_ucgstates;
  ucg::DevBuf<int> mol;  // atom->molecule of the owned beads (optional: ucg_atoms_upload_molecule)
  bool has_mol = false;
  ucg::DevBuf<double> ucgp, ucgml, mass;
  // "some resident bead may still carry ucgp < -0.999" (the first-call marker of table_ucg_bethe, UCG/pair_table_ucg_bethe.cpp:
  // 179-205): raised by every upload whose ucgp holds such a value (or whose ucgp the library cannot see), lowered when a fix
  // ucgstate pass has written ucgp of ALL owned beads; in decomposed runs agreed over the ranks with the re-neighbour
  // decision.  While it is down the Bethe gather kernel skips the first-call rules (and reports a marker it meets anyway).
  bool ucgp_first_possible = true;
  // neighbour list
  ucg::DevBuf<int> neigh, numneigh;
  ucg::DevBuf<int> ghost_src;  // owned bead each ghost images (single-rank periodic images)
  bool ghost_src_valid = false;
  int list_pitch = 0, list_maxrow = 0, list_inum = 0;
  long long list_entries = 0;  // entries of the full list
  double skin = 0.0;           // Neighbor::skin as given to ucg_domain_set (0 when the caller builds the lists)
  long long list_gen = 0;      // changes whenever the rows change (device build or upload): what derived lists key on
  bool list_from_builder = false;  // rows made by the device builder (which sets no special-bond bits)
  bool kind_blocks = true;         // option "kind_blocks": KindsDev
  bool density_tcache = true;      // option "density_tcache": PairDev::tcache
  int stream_rows = -1;            // option "stream_rows": ListDev::stream_rows (-1: by the list's size)
  // shared RanMars jump table
  ucg::DevBuf<unsigned int> rm_jump;
  int rm_chunks = 0;
  // reductions
  ucg::DevBuf<double> redpart, redout;
  // fixes
  ucg::FixLangevin lang;
  ucg::FixUcgState ucgst;
  // domain / rebuild (ucg_neigh.hip)
  ucg::Domain *dom = nullptr;
  int dom_world = 1;
  // fix cluster_switch (ucg_cluster.hip)
  ucg::ClusterSwitch *cs = nullptr;
  // communicator of a decomposed run (ucg_comm.hip)
  ucg::CommState *comm = nullptr;
  // resident driver
  ucg_pair *md_pair = nullptr;
  int md_nve = 0;  // 0 none, 1 fix nve/ucgld, 2 fix nve/ucgld/wall/hard
  bool md_lang = false, md_ucgst = false;
  bool wall_bias = false;       // fix nve/ucgld/wall/hard ... bias_potential [barrier]
  double wall_barrier = 0.1;
  bool md_no_fuse = false;  // option "md_no_fuse": keep initial_integrate a separate launch
  long long ntimestep = 0, beginstep = 0, endstep = 0;
  int groupbit = 1;
  long long nrebuild = 0, pair_error_steps = 0;
  // test aid, options "fault_inject_step" / "fault_inject_setup" (tests/test_multi_rank.py): the rank whose context has them
  // set fails locally at that timestep of ucg_md_run / inside ucg_md_setup; off (-1 / false) unless a test sets them
  long long fault_step = -1;
  bool fault_setup = false;
  double thermo[9] = {0};
  ucg::HostMirror mirror;
  // profiling
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;  // pairs (start, stop) not yet read
  long long prof_launches = 0;
  double prof_ms = 0;

  ucg::AtomsDev atoms_dev() const;
  ucg::ListDev list_dev() const;
  void ensure_rm_jump(int n);
};

struct ucg_pair {
  ucg_ctx *ctx = nullptr;
  ucg::PairModel model;
  ucg::PairDev dev{};
  bool uploaded = false;
  ucg::DevBuf<double4> d_tab, d_tabpar, d_tab_fast;
  ucg::DevBuf<int> d_pairtab;
  ucg::DevBuf<double> d_cutsq, d_mu, d_prior_type;
  ucg::DevBuf<int> d_err;
  ucg::DevBuf<int> d_blockflag;  // per workgroup of the gather kernel: 1 = some bead has a ghost neighbour
  long long blockflag_build = -1;
  int blockflag_slots = 0;
  ucg::DevBuf<double> d_evpart, d_evout;
  // table_ucg_bethe_density
  ucg::DevBuf<double2> d_prior, d_cv;
  ucg::DevBuf<double> d_partial, d_denspar;
  ucg::DevBuf<double> d_tcache;  // PairDev::tcache: one double per list entry
  ucg::DevBuf<int> d_densflags;
  std::vector<int> tabmap;  // host table id -> device table id (or -1)
  std::string err;
  size_t tab_lds_bytes = 0;
  // option pair_vrow applies (decided at ucg_pair_init: it fixes the summation mode, ucg_pair_sum_fixed): the virtual
  // rows made from the resident full rows (ucg_pair_vrow.hip)
  bool vrow = false;
  long long vr_gen = -1;  // ctx->list_gen they were made from
  int vr_pitch = 0, vr_cap = 0;
  ucg::DevBuf<int2> d_vr_lanemeta;
  ucg::DevBuf<int> d_vr_ent;  // the block lists, one after the other (vr_pitch * vr_cap ints each)
  // tables read through L2 (several actual types): host copy of the device tables and the table ids of every
  // (type, type) pair, from which the LDS block of the most populous type is made (PairDev::hot_type)
  std::vector<double4> host_tab;
  std::vector<int> host_pairtab;
  ucg::DevBuf<double4> d_tab_hot;
  ucg::DevBuf<int> d_typehist;
  long long hot_checked = -1;  // ctx->nrebuild at the last choice
  // kind blocks (KindsDev): made once at ucg_pair_init
  bool kinds = false;
  ucg::DevBuf<double4> d_kind_tab;
  ucg::DevBuf<int2> d_kind_dir;
  double host_boltz = 1.0;  // used by host-only pairs (no context)
  explicit ucg_pair(int style) : model(style) {}
};

namespace ucg {
// ucg_neigh.hip
void domain_destroy(ucg_ctx *ctx);
void cluster_destroy(ucg_ctx *ctx);
void comm_destroy(ucg_ctx *ctx);
int md_setup_multi(ucg_ctx *ctx);
int md_run_multi(ucg_ctx *ctx, long long nsteps, int thermo_every, int ev_on_last);
int halo_pack_peers(ucg_ctx *ctx, void *sendbuf, long long so_self, long long nself);
int halo_unpack_self(ucg_ctx *ctx, const void *recvbuf, long long ro_self, long long so_self, long long nself);
int decide_local_impl(ucg_ctx *ctx, int *due, int *flag, int *pair_flag);
int decide_launch(ucg_ctx *ctx, int *due, int *checked, long long *dev_out3);
int exchange_count_launch(ucg_ctx *ctx, const int **dev_counts);
int border_count_launch(ucg_ctx *ctx, const int **dev_counts);
int counts_adopt(ucg_ctx *ctx, int which, const long long *sendcounts);
bool cluster_forces_rebuild(const ucg_ctx *ctx);
void cluster_pre_exchange(ucg_ctx *ctx);
}
