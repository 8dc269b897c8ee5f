/* ucg_hip.h -- C ABI of libucg_hip.so, the MI355X (gfx950) implementation of the
 * UCG hot path of LAMMPS-UCG (KJAdams2000/LAMMPS-UCG-dev).
 *
 * Plain C, plain pointers and sizes; no C++ types, no exceptions, no torch.
 * Every entry point returns 0 on success and a nonzero UCG_ERR_* code on
 * failure; ucg_last_error() then gives the message the LAMMPS glue hands to
 * error->one()/error->all().  One context per rank/GPU, called from one host
 * thread (the reference is single-threaded per rank).
 *
 * Each group below names the reference interface it replaces (paths relative
 * to the reference root).  INTEGRATION.md shows the LAMMPS-side classes that
 * bind these calls under the reference's style names.
 *
 * Host arrays use the layouts LAMMPS hands to a pair style / fix:
 *   x, v, f            double[n][3]  (AoS)
 *   ucgsoftmaxscores   double[n][2]
 *   type, tag, mask, ucgstate, num_ucgstates   int[n]
 *   ucgl, ucgvl, ucgml, ucgp, ucgforce         double[n]
 * with n = nlocal (+ nghost where stated); ghosts follow owned atoms.
 */
#ifndef UCG_HIP_H
#define UCG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define UCG_ABI_VERSION 1

enum {
  UCG_OK = 0,
  UCG_ERR_INVALID = 1,      /* bad argument / call order                         */
  UCG_ERR_INPUT = 2,        /* input-deck error (the reference's error->all)      */
  UCG_ERR_HIP = 3,          /* HIP runtime failure                                */
  UCG_ERR_TABLE_INNER = 4,  /* "Pair distance < table inner cutoff"  (ucgld.cpp:437-439) */
  UCG_ERR_TABLE_OUTER = 5,  /* "Pair distance > table outer cutoff"  (ucgld.cpp:442-444) */
  UCG_ERR_UNSUPPORTED = 6,  /* feature the GPU path does not cover (e.g. prior noise) */
  UCG_ERR_NEIGH_OVERFLOW = 7,
  UCG_ERR_COMM = 8          /* communicator failure of a decomposed run (RCCL / caller's callbacks) */
};

enum { UCG_STYLE_UCGLD = 0, UCG_STYLE_BETHE = 1, UCG_STYLE_BETHE_DENSITY = 2 };
enum { UCG_LOOKUP = 0, UCG_LINEAR = 1, UCG_SPLINE = 2, UCG_BITMAP = 3 };

/* neighbour-entry bits (device lists and ucg_neigh_upload_full): [28:0] index,
 * [29] orientation (1: the row owner is the reference's "i" of the pair, i.e.
 * tag_row <= tag_neighbour), [31:30] LAMMPS special-bond code (sbmask). */
#define UCG_NEIGHMASK 0x1FFFFFFF
#define UCG_ORIENT_BIT 29
#define UCG_SBBITS 30

typedef struct ucg_ctx ucg_ctx;
typedef struct ucg_pair ucg_pair;

/* ------------------------------------------------------------------ context */

int ucg_abi_version(void);
/* visible HIP devices (0 when there is none or the runtime fails); touches no device */
int ucg_device_count(void);
/* device < 0: use the current HIP device */
int ucg_ctx_create(int device, ucg_ctx **out);
void ucg_ctx_destroy(ucg_ctx *ctx);
const char *ucg_last_error(const ucg_ctx *ctx);
/* run on a caller-owned hipStream_t (passed as void*), e.g. torch's current stream so that
 * the kernels are ordered with RCCL collectives; NULL = the legacy default stream.  A new
 * context starts with a non-blocking stream of its own. */
int ucg_ctx_set_stream(ucg_ctx *ctx, void *hip_stream);
int ucg_ctx_synchronize(ucg_ctx *ctx);
/* force->boltz, force->ftm2v, force->mvv2e, update->dt, force->special_lj[0..3] */
int ucg_ctx_set_units(ucg_ctx *ctx, double boltz, double ftm2v, double mvv2e, double dt,
                      const double *special_lj);

/* options: "generic_kernels" = 1 keeps the pair kernels on their general code path (per-table
 * grid lookup, IEEE division) even when the faster equivalent variants apply; takes effect
 * at the next ucg_pair_init.  Used by the tests to check that both give the same bits.
 * "density_proximity_as_shipped" = 1 makes table_ucg_bethe_density use the proximity function
 * itself in the CV back-force, as shipped (:719), instead of its derivative.
 * Tuning / diagnostics (none changes a result bit except where stated):
 *   "gather_slots"  lanes per bead of the gather kernels: 0 = chosen from the bead count, or 1 (default), 2, 4, 8, 16;
 *                   part of the canonical summation order (ucg_pair_gather_slots reports the value in use)
 *   "rng_batch"     steps of per-bead RanMars draws generated per launch, 1..64 (default 10)
 *   "stage_own"     own-block staging of the gather kernels in LDS (default 1)
 *   "hot_block"     tables too large for the LDS: with several actual types the three tables of the pairs of the most
 *                   populous type with itself, with one type the far end of the r^2 grid, are staged in LDS all the
 *                   same and read there by the lanes they serve (default 1; set before ucg_pair_init)
 *   "kind_blocks"   several actual types, tables read through L1 / L2 (all three styles): the lanes that do so read a
 *                   compact block per (row type, neighbour type) kind instead of the full layout (default 1; set before
 *                   ucg_pair_init)
 *   "density_tcache" table_ucg_bethe_density: pass 1 leaves tanh of the proximity argument of every in-cutoff entry in a buffer
 *                   of one double per list entry and pass 3 reads it back instead of evaluating it again (default 1; the
 *                   same bits either way)
 *   "stream_rows"   -1 (default): neighbour rows of more than 192 MB -- more than the last-level cache keeps from step to step --
 *                   are read with non-temporal loads, so that they do not displace the beads the gathers re-read; 0 / 1:
 *                   never / always (the same bits either way)
 *   "post_in_pair"  per-bead hooks in the gather kernel's epilogue (default 1), "md_no_fuse" = 1 runs every hook as
 *                   its own kernel
 *   "rows_untiled"  = 1 builds neighbour rows with the one-lane-per-bead kernels (the fallback of the tiled builder)
 *   "pair_vrow"     = 1 (set before ucg_pair_init) table_ucgld / table_ucg_bethe whose tables fit the LDS (one shared
 *                   r^2 grid, no BITMAP tables) run on balanced virtual rows: the pairs of two beads of one 512-bead
 *                   workgroup block are evaluated once, and a bead's terms are summed as fixed sums (ucg_pair_sum_fixed).
 *                   Off by default: measured 6 % (table_ucgld) / 2 % (table_ucg_bethe) slower than the full-row gather
 *                   kernels at 1 M beads, whose ordered double sums stay the default for every style
 *   "fma_contract"  = 1 runs the gather kernels compiled with FMA contraction: NOT the bit-exact path (results within
 *                   1e-12), never the default */
int ucg_ctx_set_option(ucg_ctx *ctx, const char *name, int value);
/* PMC calibration aid: stream a fresh buffer of nbytes with 4-byte (wide=0) or 16-byte (wide=1)
 * loads per lane, `repeats` launches of k_stream; lets FETCH_SIZE be calibrated on a known
 * byte count in this library's own access widths (MI355X_MICROARCH.md, HBM section). */
int ucg_selftest_stream(ucg_ctx *ctx, long long nbytes, int wide, int repeats);
/* device self-test: n random operands, counts a / b != div_by_const(a, b) (must be 0) */
int ucg_selftest_div(ucg_ctx *ctx, double b, long long seed, int n, long long *mismatches);
/* the same for the bare Newton-Raphson core the math kernels use where the operand scaling of the hardware division is the
 * identity (csrc/ucg_math.h: ucg_div_core): n random numerators (exponents -112 ... 2, and zeros) over denominators in
 * [1.25, 2.75] and [5.25, 6.75]; counts the quotients that differ from a / b (must be 0) */
int ucg_selftest_div_core(ucg_ctx *ctx, long long seed, int n, long long *mismatches);
/* ... and for the bare square-root iteration of the Bethe closure (csrc/ucg_math.h: ucg_sqrt_core), used for arguments in
 * [2^-700, 2^700): n random arguments of that range, one in eight a perfect square or its neighbour in the last place;
 * counts the roots that differ from sqrt(x) (must be 0) */
int ucg_selftest_sqrt_core(ucg_ctx *ctx, long long seed, int n, long long *mismatches);

/* --------------------------------------------------------------- pair styles
 * replaces PairTable_UCGLD / PairTable_UCG_Bethe / PairTable_UCG_Bethe_Density
 *   settings()   UCG/pair_table_ucgld.cpp:654-716, UCG/pair_table_ucg_bethe.cpp:746-886,
 *                UCG/pair_table_ucg_bethe_density.cpp:896-958  (+ read_state_settings)
 *   coeff()      UCG/pair_table_ucgld.cpp:719-865 (read_table/spline_table/compute_table)
 *   init_style() + init_one()  UCG/pair_table_ucgld.cpp:867-895
 *   compute()    UCG/pair_table_ucgld.cpp:111-541, UCG/pair_table_ucg_bethe.cpp:88-630,
 *                UCG/pair_table_ucg_bethe_density.cpp:133-758
 * argv arrays are the words after the style name, exactly as LAMMPS passes them.
 * NOT offered on the device: `prior chemical_potential noise L seed` of table_ucg_bethe (UCG/pair_table_ucg_bethe.cpp:187-188,
 * 235-236, 844-856).  It is parsed (ucg_pair_settings accepts the words) and ucg_pair_init then returns UCG_ERR_UNSUPPORTED
 * with a message: the reference draws one RanMars number per bead and one per HALF-list entry inside the neighbour sweep of
 * the first force evaluation, before the cutoff test, so the result is a function of upstream's half-list order (undefined
 * by the reference) and, across ranks, of which rank's sweep holds a pair -- there is no order-free statement of it to be
 * bit-exact against.  `prior ucgl` and `prior chemical_potential` (without noise) are offered. */

int ucg_pair_create(ucg_ctx *ctx, int style, ucg_pair **out);
/* host-only pair: the setup half of the style (settings, coeff, init, single, table
 * inspection) without a device; compute() is refused.  This is input parsing and
 * table construction, which the reference also does on the host -- not a fallback. */
int ucg_pair_create_host(int style, double boltz, ucg_pair **out);
const char *ucg_pair_last_error(const ucg_pair *p);
void ucg_pair_destroy(ucg_pair *p);
int ucg_pair_settings(ucg_pair *p, int narg, const char *const *arg);
int ucg_pair_coeff(ucg_pair *p, int ntypes, int narg, const char *const *arg);
/* Pair::init(): T = thermostat t_target found through Fix::extract("t_target")
 * (UCG/pair_table_ucgld.cpp:873-881; must be given);
 * uploads the packed tables and type maps to the device */
int ucg_pair_init(ucg_pair *p, int ntypes, double T);
/* what init_one(i,j) returns for formal types (i,j): the table cutoff */
double ucg_pair_cut(const ucg_pair *p, int i, int j);
/* largest table cutoff (Pair::cutforce) */
double ucg_pair_cutforce(const ucg_pair *p);
/* Pair::single(): UCG/pair_table_ucgld.cpp:1474-1520, evaluated on the host */
int ucg_pair_single(const ucg_pair *p, int itype, int jtype, double rsq, double factor_lj,
                    double *fforce, double *energy);
/* lanes per bead of the table_ucgld / table_ucg_bethe kernels (option "gather_slots": 1, 4, 8, 16;
 * default 1).  It is part of the canonical summation order: entry e of a row is added into partial
 * sum e % slots, and the partial sums are combined by the tree s[l] += s[l + slots/2], ..., s[l] += s[l+1]. */
int ucg_pair_gather_slots(const ucg_pair *p);
/* How this pair sums a bead's terms (decided at ucg_pair_init; anyone comparing bits needs it):
 *   0 = ordered double sums: the canonical order above (row order, gather_slots);
 *   1 = fixed sums (option "pair_vrow" applies): every term of a bead's force components, ucgforce and scores is rounded
 *       to nearest-even at 2^-38 of a per-field power-of-two unit and the images are added as 64-bit integers, so the
 *       sum depends on no order -- not on the rows, the lanes, which lane evaluates a pair, or the decomposition.
 *       Units: 2^-(4 - ilogb(R)) with R = max |f(k)| sqrt(rsq_k) (forces), max |e(k)| (ucgforce), max |e(k)| / kT
 *       (scores) over the reachable tables' knots in the upper three quarters of the r^2 grid.  bead total = prologue
 *       value + (double) integer sum * unit * 2^-38.  A term of 2^24 units or more sets an error
 *       (ucg_pair_check_errors: UCG_ERR_UNSUPPORTED).  The oracle states the same (oracle/orc.h, sum_fixed). */
int ucg_pair_sum_fixed(const ucg_pair *p);
/* host copies of the built tables, for inspection: which in
 * {"rsq","e","f","de","df","e2","f2"}; returns the length or <0 */
int ucg_pair_table_count(const ucg_pair *p);
int ucg_pair_table_params(const ucg_pair *p, int m, double *out5 /* innersq,delta,invdelta,deltasq6,cut */);
int ucg_pair_table_array(const ucg_pair *p, int m, const char *which, double *out, int cap);
/* tabindex[(n_formal+1)^2] after init */
int ucg_pair_tabindex(const ucg_pair *p, int *out, int cap);

/* One force evaluation on the atoms/list resident in the context.  Writes
 * f, ucgforce, ucgsoftmaxscores, num_ucgstates (and ucgp for the density
 * style) of the owned atoms.  eng_vdwl / virial[6] (xx,yy,zz,xy,xz,yz) may be
 * NULL when eflag / vflag are 0. */
int ucg_pair_compute(ucg_pair *p, int eflag, int vflag, double *eng_vdwl, double *virial);
/* ucg_pair_check_errors: device-side error flags of the computes since the last call (0 = none): UCG_ERR_TABLE_INNER /
 * UCG_ERR_TABLE_OUTER = a pair distance outside the tables' range -- the glue turns those into error->one() like
 * UCG/pair_table_ucgld.cpp:437-444; with option "pair_vrow": UCG_ERR_UNSUPPORTED when a block's virtual rows do not fit
 * (a row of more than 128 entries, a list of more than 32 768 entries) or a term left the range of the fixed sums. */
/* the same compute() in two launches for decomposed runs (table_ucgld / table_ucg_bethe, no
 * energy/virial): part 1 = the workgroups none of whose beads has a ghost neighbour -- they can run
 * while the halo is in flight --, part 2 = the rest, after ucg_halo_unpack.  Together they write
 * exactly what ucg_pair_compute writes. */
int ucg_pair_compute_part(ucg_pair *p, int part);
int ucg_pair_check_errors(ucg_pair *p);
/* table_ucg_bethe_density on a decomposed run: the three passes of compute() one at a time
 * (phase 1 :219-274 local densities and priors, 2 :284-664 pair forces and CV force accumulators,
 * 3 :669-734 posterior + CV back-force); between them the caller forwards the ghosts' entries of
 * buffer 0 (priors, after phase 1) and buffer 1 (CV forces, after phase 2) from their owner ranks
 * with ucg_halo_aux_pack / ucg_halo_aux_unpack -- the halo the reference's no-op forward_comm
 *  was meant to be.  ucg_pair_density_buffer returns a DEVICE pointer
 * (double2 per owned + ghost bead). */
int ucg_pair_density_phase(ucg_pair *p, int phase, int eflag, int vflag, double *eng_vdwl, double *virial);
void *ucg_pair_density_buffer(ucg_pair *p, int which);
/* host access to `count` entries (two doubles each) of buffer `which` starting at bead `first`, for a caller that
 * moves them itself between the phases (LAMMPS' comm->forward_comm(Pair *): the glue downloads the owned entries,
 * forwards them and uploads the ghosts') */
int ucg_pair_density_aux_download(ucg_pair *p, int which, double *host, int first, int count);
int ucg_pair_density_aux_upload(ucg_pair *p, int which, const double *host, int first, int count);

/* ------------------------------------------------------------- atoms (AtomVecUCG)
 * replaces the per-atom fields of atom style "ucg": UCG/atom_vec_ucg.cpp:48-90,
 * atom.h:180-192.  Device mirrors only; LAMMPS keeps ownership of host arrays. */

int ucg_atoms_upload(ucg_ctx *ctx, int nlocal, int nghost, int ntypes, const double *x,
                     const double *v, const int *type, const int *tag, const int *mask,
                     const int *ucgstate, const double *ucgl, const double *ucgvl,
                     const double *ucgml, const double *ucgp, const double *mass);
/* fields_comm of UCG/atom_vec_ucg.cpp:71: refresh x, ucgstate, ucgl, ucgp of all nall atoms */
int ucg_atoms_upload_comm(ucg_ctx *ctx, const double *x, const int *ucgstate, const double *ucgl,
                          const double *ucgp);
/* drop-in fix hooks: refresh owned-atom fields a hook reads from LAMMPS' host arrays (any pointer may
 * be NULL = keep the device value); x/v/f are [nlocal][3], ucgsoftmaxscores [nlocal][2] */
int ucg_atoms_upload_owned(ucg_ctx *ctx, const double *x, const double *v, const double *f,
                           const int *ucgstate, const int *num_ucgstates, const double *ucgl,
                           const double *ucgvl, const double *ucgp, const double *ucgforce,
                           const double *ucgsoftmaxscores);
/* any pointer may be NULL; arrays are nlocal long (x: nlocal+nghost if with_ghosts) */
int ucg_atoms_download(ucg_ctx *ctx, int with_ghosts, double *x, double *v, double *f, int *type,
                       int *tag, int *ucgstate, int *num_ucgstates, double *ucgl, double *ucgvl,
                       double *ucgml, double *ucgp, double *ucgforce, double *ucgsoftmaxscores);
int ucg_atoms_counts(const ucg_ctx *ctx, int *nlocal, int *nghost);
/* atom->mask of the owned atoms in the device's current order (beads migrate and are re-sorted in the resident loops) */
int ucg_atoms_download_mask(ucg_ctx *ctx, int *mask);
/* which owned atom each ghost is a periodic image of (host-built ghosts only; the device
 * builder records it itself).  table_ucg_bethe_density needs it to give ghosts their owner's
 * prior and CV force -- the forward_comm the reference declares but never performs
 * (UCG/pair_table_ucg_bethe_density.cpp:280 vs ...density.h:107-110). */
int ucg_ghosts_upload(ucg_ctx *ctx, const int *src, int nghost);
/* ... and, for a single rank, by which box shifts (-1, 0, 1 per dimension: CommBrick's pbc flags).  With ucg_domain_set
 * this lets the device refresh the host-built ghosts itself (ucg_halo_forward) and take the re-neighbour decision against
 * the positions held at this call (ucg_decide_local), so a drop-in run moves nothing per step (ucg_host_bind below). */
int ucg_ghosts_upload_images(ucg_ctx *ctx, const int *src, const int *shift3, int nghost);
/* AtomVecUCG::force_clear (UCG/atom_vec_ucg.cpp:131-135) + Verlet::force_clear */
int ucg_force_clear(ucg_ctx *ctx);

/* ------------------------------------------------------------ neighbour lists
 * (upstream Neighbor; the styles only request lists: ucgld.cpp:868, density.cpp:1135) */

/* host-built FULL list over owned rows; entries as described above */
int ucg_neigh_upload_full(ucg_ctx *ctx, int inum, const int *numneigh, const long long *first,
                          const int *neigh);
/* periodic orthogonal box + neighbour settings for the device builder */
int ucg_domain_set(ucg_ctx *ctx, const double *boxlo, const double *boxhi, double cutforce,
                   double skin, int every, int delay, int check);
/* wrap, sort by (bin, tag), rebuild periodic-image ghosts, bin, build the full list */
int ucg_neigh_rebuild(ucg_ctx *ctx);
/* owner -> periodic images (fields_comm) on the device */
int ucg_halo_forward(ucg_ctx *ctx);
/* device list back to the host (CSR); call with neigh==NULL to get the sizes */
int ucg_neigh_download(ucg_ctx *ctx, int *inum, int *numneigh, long long *first, int *neigh,
                       long long cap, long long *total);
/* ghost map of the device builder: source owned index + periodic shift of each ghost */
int ucg_ghosts_download(ucg_ctx *ctx, int *src, int *shift3, int cap);

/* ----------------------------------------------------- multi-rank (one process per GPU)
 * Spatial decomposition of the periodic box into procgrid[0] x procgrid[1] x procgrid[2]
 * bricks, rank me = ix + px*(iy + py*iz).  These calls only count / pack / unpack on the
 * device; the caller moves the buffers (torch.distributed all_to_all over RCCL, or MPI in a
 * LAMMPS build).  Replaces upstream CommBrick::exchange / borders / forward_comm driven by
 * the field lists of UCG/atom_vec_ucg.cpp:66-82.  There is no reverse halo: the gather
 * kernels accumulate nothing on ghosts.  Call order at a rebuild:
 *   exchange_count -> (all_to_all counts) -> exchange_pack -> (all_to_all data) ->
 *   exchange_unpack -> border_count -> (counts) -> border_pack -> (data) -> border_unpack
 * and every other step: halo_pack -> (all_to_all, same counts) -> halo_unpack.
 * sendcounts arrays are `world` long; buffers are device pointers holding records of
 * ucg_record_bytes() bytes grouped by destination rank in rank order. */
int ucg_decomp_set(ucg_ctx *ctx, const int *procgrid3, int me);
int ucg_record_bytes(int *atom_record_bytes, int *halo_record_bytes);
int ucg_exchange_count(ucg_ctx *ctx, long long *sendcounts);
int ucg_exchange_pack(ucg_ctx *ctx, void *sendbuf);
int ucg_exchange_unpack(ucg_ctx *ctx, const void *recvbuf, long long nrecv);
int ucg_border_count(ucg_ctx *ctx, long long *sendcounts);
int ucg_border_pack(ucg_ctx *ctx, void *sendbuf);
int ucg_border_unpack(ucg_ctx *ctx, const void *recvbuf, long long nrecv);
int ucg_halo_pack(ucg_ctx *ctx, void *sendbuf);
int ucg_halo_unpack(ucg_ctx *ctx, const void *recvbuf);
/* Neighbor::decide(): *due = a check is scheduled this step, *flag = a local bead moved > skin/2 */
/* forward halo of one double2 per bead with the send lists of the last ucg_border_pack
 * (16 bytes per ghost, grouped by destination rank like ucg_halo_pack) */
int ucg_halo_aux_pack(ucg_ctx *ctx, const void *field_dev, void *sendbuf);
int ucg_halo_aux_unpack(ucg_ctx *ctx, void *field_dev, const void *recvbuf);
int ucg_decide_local(ucg_ctx *ctx, int *due, int *flag);

/* ------------------------------------------------- communicator of a decomposed run
 * With a communicator attached, ucg_md_setup / ucg_md_run drive the whole rank-level step loop inside the library
 * (csrc/ucg_comm.hip): exchange / borders at a re-neighbouring, one forward halo per step, the density style's two
 * mid-compute halos, the MPI_Allreduce steps of Neighbor::decide and of fix cluster_switch, the thermo all-reduce --
 * what upstream CommBrick + Verlet do around the styles of this package.  Two kinds of communicator:
 *   ucg_comm_attach_rccl   built in: RCCL called directly on the context's stream (grouped ncclSend / ncclRecv to the
 *                          <= 7 peers, ncclAllReduce for the small host reductions); the caller only distributes the
 *                          128-byte id of rank 0 (ucg_comm_rccl_unique_id) -- MPI_Bcast in a LAMMPS build;
 *                          librccl.so.1 is loaded at the first use (environment UCG_RCCL_LIBRARY: another build of it);
 *   ucg_comm_attach        the caller's callbacks (MPI in a LAMMPS build without RCCL, gloo in the tests):
 *     alltoallv      DEVICE buffers; sendbytes[r] bytes for rank r lie in consecutive blocks of `send` in rank
 *                    order, the block received from rank r goes to `recv` in rank order; must be ordered after the
 *                    work queued on `stream` (a hipStream_t) and complete, or be ordered on it, on return
 *     alltoall_ll    host arrays, one long long per rank each way, blocking
 *     allreduce_ll / allreduce_f64   host arrays of n elements, in place, blocking; op 0 = sum, 1 = max, 2 = min
 * Callbacks return 0 on success.  Errors: a rank-local failure (a HIP error, an overflowing neighbour row, a pair-table
 * range violation) is held by that rank, which keeps taking part in the collectives with empty or stale messages until the
 * next status agreement (the all-reduce of the re-neighbour decision, an error poll, the end of the re-neighbouring), where
 * EVERY rank returns an error code from ucg_md_setup / ucg_md_run together -- no rank is left blocked in a receive.  The
 * simulation state is undefined afterwards: the caller must abort the job (MPI_Abort).  A failure of the communicator
 * itself (UCG_ERR_COMM from RCCL or a callback) cannot be agreed on and is returned at once.
 * fix cluster_switch on a decomposed run: create it on every rank after the beads
 * and molecule ids are uploaded; the library performs its reductions at the next ucg_md_setup. */
typedef struct ucg_comm_ops {
  void *user;
  int rank, world;
  int (*alltoallv)(void *user, const void *send, const long long *sendbytes, void *recv, const long long *recvbytes,
                   void *stream);
  int (*alltoall_ll)(void *user, const long long *send, long long *recv);
  int (*allreduce_ll)(void *user, long long *buf, int n, int op);
  int (*allreduce_f64)(void *user, double *buf, int n, int op);
} ucg_comm_ops;
typedef struct ucg_rccl_id { char internal[128]; } ucg_rccl_id; /* = ncclUniqueId */
int ucg_comm_attach(ucg_ctx *ctx, const ucg_comm_ops *ops);
/* the same callbacks for a caller without device-aware transport (plain MPI in a LAMMPS build: MPI_Alltoallv on bytes): the
 * library stages every message through pinned HOST buffers of its own, so alltoallv receives HOST pointers (`stream` is NULL
 * and everything queued before has completed); nothing else differs from ucg_comm_attach */
int ucg_comm_attach_host(ucg_ctx *ctx, const ucg_comm_ops *ops);
int ucg_comm_rccl_unique_id(ucg_rccl_id *out);
int ucg_comm_attach_rccl(ucg_ctx *ctx, const ucg_rccl_id *id, int rank, int world);
int ucg_comm_detach(ucg_ctx *ctx);
int ucg_comm_info(const ucg_ctx *ctx, int *rank, int *world, int *is_rccl, long long *nrebuild);
/* what the attached communicator really is: out[0] = 1 RCCL / 0 callbacks, out[1] = ranks RCCL itself reports for the
 * communicator (ncclCommCount; 0 without RCCL), out[2] = device RCCL reports (ncclCommCuDevice; -1), out[3] = 1 when the
 * callbacks are host-staged (ucg_comm_attach_host).  bench.py prints it, so that a scaling line cannot pass a host-staged
 * rehearsal off as an RCCL run */
int ucg_comm_transport(const ucg_ctx *ctx, int *out4);
int ucg_comm_allreduce_f64(ucg_ctx *ctx, double *buf, int n, int op);

/* ------------------------------------------------ host mirrors of a drop-in caller
 * The reference's styles work on LAMMPS' host arrays, hook by hook (Pair::compute UCG/pair_table_ucgld.h:22-48, the
 * integrator UCG/fix_nve_ucgld.h:27-36, the thermostat UCG/fix_ucgld_langevin.h:29-47, fix ucgstate
 * UCG/fix_ucgstate.h:15-23).  Instead of copying every array a hook touches in and out (ucg_atoms_upload_owned /
 * ucg_atoms_download), a caller can BIND its arrays of the owned atoms once (again after AtomVec::grow_pointers) and
 * let the library keep the device arrays authoritative between hooks -- upstream's KOKKOS sync / modified protocol:
 *   ucg_host_modified(mask)  the caller has written these fields: the next device hook that reads them uploads them first
 *   ucg_host_sync(mask)      the caller is about to read these fields: those a device hook has written since the last
 *                            synchronisation are downloaded (re-neighbouring, thermo and dump steps; nothing moves on an
 *                            ordinary step)
 * Every hook below marks what it reads and writes.  ucg_atoms_upload leaves both sides equal; a device re-neighbouring
 * (ucg_neigh_rebuild) re-orders the beads, so afterwards every field counts as written by the device.
 * ucg_host_status reports the two masks and {uploads, downloads} that moved data.  Pinned arrays (hipHostRegister)
 * make the copies run at PCIe speed; the AoS <-> device-record repacking is done on the device. */
enum {
  UCG_F_X = 1, UCG_F_V = 2, UCG_F_F = 4, UCG_F_STATE = 8, UCG_F_NSTATES = 16, UCG_F_UCGL = 32, UCG_F_UCGVL = 64,
  UCG_F_UCGP = 128, UCG_F_UCGFORCE = 256, UCG_F_SCORES = 512, UCG_F_ALL = 1023
};
int ucg_host_bind(ucg_ctx *ctx, double *x, double *v, double *f, int *ucgstate, int *num_ucgstates, double *ucgl,
                  double *ucgvl, double *ucgp, double *ucgforce, double *ucgsoftmaxscores);
int ucg_host_modified(ucg_ctx *ctx, int mask);
int ucg_host_sync(ucg_ctx *ctx, int mask);
int ucg_host_status(const ucg_ctx *ctx, int *device_newer, int *host_newer, long long *transfers2);
/* the package's hooks in the order upstream Verlet::run calls them, one C-ABI call per hook (csrc/ucg_host.hip): what a
 * LAMMPS run makes of these styles, without LAMMPS -- the drop-in leg of bench.py and the C caller of the tests.  The
 * re-neighbour decision is taken on the device (the glue's integrator forces LAMMPS' re-neighbouring through
 * Fix::force_reneighbor); bound host mirrors are synchronised before every re-neighbouring when sync_on_reneighbour is set
 * (x v ucgstate ucgl ucgvl ucgp: what LAMMPS' exchange / borders move; 0 = the package re-neighbours on its own, on the
 * device) and completely every `sync_every` steps (0 = never).  use_nve as in ucg_md_attach.
 * stats4: re-neighbourings, host synchronisations, uploads, downloads that moved data. */
int ucg_verlet_hooks_run(ucg_ctx *ctx, ucg_pair *pair, long long nsteps, int use_nve, int use_langevin, int use_ucgstate,
                         int groupbit, int sync_on_reneighbour, int sync_every, long long *stats4);

/* ---------------------------------------------------------------- fix nve/ucgld
 * replaces FixNVE_UCGLD::initial_integrate / final_integrate
 * (UCG/fix_nve_ucgld.cpp:44-101, 104-153), per-type mass branch */
int ucg_fix_nve_initial(ucg_ctx *ctx, int groupbit);
int ucg_fix_nve_final(ucg_ctx *ctx, int groupbit);

/* ---------------------------------------------------------------- fix nve/ucgld/wall/hard
 * FixNVE_UCGLD_Wall_Hard (UCG/fix_nve_ucgld_wall_hard.cpp): the nve/ucgld update, plus
 * ucgstate = (ucgl < 0.5 ? 0 : 1) after the drift (initial_integrate :97-103), reflection of ucgl
 * and ucgvl at 0 and 1 after the second half-kick (final_integrate :171-177), and -- with the
 * keyword `bias_potential [barrier]` (:21-33, default barrier 0.1) -- the bias force
 * (-7980 x^9 + 2 x) * 10 * barrier, x = ucgl - 0.5, added to ucgforce in post_force (:216-241). */
int ucg_fix_nve_wall_hard_set(ucg_ctx *ctx, int bias_potential, double barrier);
int ucg_fix_nve_wall_hard_initial(ucg_ctx *ctx, int groupbit);
int ucg_fix_nve_wall_hard_final(ucg_ctx *ctx, int groupbit);
int ucg_fix_nve_wall_hard_post_force(ucg_ctx *ctx, int groupbit);

/* ----------------------------------------------------------- fix ucgld/langevin
 * replaces Fix_UCGLD_Langevin (UCG/fix_ucgld_langevin.cpp): constructor :54-119,
 * init :149-183, post_force_templated<0> :226-297, end_of_step :303-312,
 * compute_scalar :403-406, extract("t_target") :412-417 */
int ucg_fix_langevin_create(ucg_ctx *ctx, double t_start, double t_stop, double t_period, int seed,
                            int me);
/* per-type prefactors computed by the glue exactly as init() does (it reads
 * atom->ucgml[type index]); arrays are ntypes+1 long */
int ucg_fix_langevin_init(ucg_ctx *ctx, int ntypes, const double *gfactor1, const double *gfactor2);
/* convenience: the reference's init() arithmetic from mλ values indexed by type */
int ucg_fix_langevin_init_from_ucgml(ucg_ctx *ctx, int ntypes, const double *ucgml_by_type_index);
int ucg_fix_langevin_post_force(ucg_ctx *ctx, int groupbit, long long ntimestep,
                                long long beginstep, long long endstep);
int ucg_fix_langevin_end_of_step(ucg_ctx *ctx, int groupbit, double *lambda_temp);
double ucg_fix_langevin_t_target(const ucg_ctx *ctx);
/* reset_target (:358-361): t_target = t_start = t_stop = t_new */
int ucg_fix_langevin_reset_target(ucg_ctx *ctx, double t_new);
/* reset_dt (:366-376) AS SHIPPED: only gfactor2 is rebuilt, from atom->mass[type] (not ucgml) and the context's current
 * dt; gfactor1 keeps its value.  mass_by_type is ntypes+1 long (NULL: the masses of the last
 * ucg_atoms_upload) */
int ucg_fix_langevin_reset_dt(ucg_ctx *ctx, int ntypes, const double *mass_by_type);
/* fix_modify temp with a compute that removes a velocity bias (tbiasflag == BIAS, :162-165): post_force_templated<1>
 * (:283-291) zeroes the random force of a bead whose lambda velocity is exactly 0.  bias = 0 is the default. */
int ucg_fix_langevin_set_bias(ucg_ctx *ctx, int bias);

/* ---------------------------------------------------------------- fix ucgstate
 * replaces FixUCGState::post_force (UCG/fix_ucgstate.cpp:88-132);
 * mode: ld_flag / mc_flag / seed / rate as parsed at :37-67 */
int ucg_fix_ucgstate_create(ucg_ctx *ctx, int ld_flag, int mc_flag, int mc_seed, double mc_rate,
                            int me);
int ucg_fix_ucgstate_post_force(ucg_ctx *ctx);

/* ---------------------------------------------------------------- fix cluster_switch
 * replaces FixClusterSwitch (UCG/fix_cluster_switch.cpp): constructor + read_file + read_contacts
 * :37-344, pre_exchange :452-469, check_cluster :551-719, attempt_switch / confirm_molecule /
 * switch_flag / gather_statistics :721-935, compute_vector :887-897.  Needs atom->molecule
 * (ucg_atoms_upload_molecule, same order as the last ucg_atoms_upload) and the device-built full
 * list; single rank.  In the resident loop (ucg_md_run) it forces a re-neighbour every
 * switch_freq steps, as the reference's force_reneighbor / next_reneighbor do. */
int ucg_atoms_upload_molecule(ucg_ctx *ctx, const int *molecule);
int ucg_atoms_download_molecule(ucg_ctx *ctx, int *molecule);
/* fix ID group cluster_switch mol_seed mol_offset cutoff seed rateFreq switch_freq rateFile F contactFile F */
int ucg_fix_cluster_switch_create(ucg_ctx *ctx, int groupbit, int mol_seed, int mol_offset, double cutoff,
                                  int seed, int switch_freq, const char *rate_file,
                                  const char *contact_file);
int ucg_fix_cluster_switch_check_cluster(ucg_ctx *ctx);
int ucg_fix_cluster_switch_attempt_switch(ucg_ctx *ctx);
int ucg_fix_cluster_switch_maxmol(const ucg_ctx *ctx);
/* which: 0 mol_cluster, 1 mol_state, 2 mol_restrict, 3 mol_accept; out has maxmol+1 entries */
int ucg_fix_cluster_switch_array(ucg_ctx *ctx, int which, int *out);
/* compute_vector: attempts, successes, attempts ON, attempts OFF, successes ON, successes OFF, cluster size */
int ucg_fix_cluster_switch_vector(const ucg_ctx *ctx, double *out7);
/* Decomposed runs.  Molecule ids travel with migrating beads; ghosts get {group mask, molecule id} through
 * ucg_halo_molmask_pack / _unpack (8 bytes per ghost, after ucg_border_unpack).  The caller performs the
 * reductions the reference does with MPI_Allreduce: after _create the survey scalars (_scalars: maxmol MAX,
 * switchable atoms of mol_seed SUM, switchable atoms SUM -> _set_scalars, UCG/fix_cluster_switch.cpp:114-120)
 * and mol_state / mol_restrict / presence (arrays 1, 2, 4: MAX, :157-158); in check_cluster the labels between
 * sweeps (array 5: MIN, :664) until no rank changed anything; in attempt_switch mol_accept (array 3: MAX, :750).
 * _set_array accepts which = 1..5; _array additionally 4 = presence, 5 = the device labels. */
int ucg_halo_molmask_pack(ucg_ctx *ctx, void *sendbuf);
int ucg_halo_molmask_unpack(ucg_ctx *ctx, const void *recvbuf);
int ucg_fix_cluster_switch_scalars(const ucg_ctx *ctx, long long *out3);
int ucg_fix_cluster_switch_set_scalars(ucg_ctx *ctx, long long maxmol, long long nspm, long long nmolatoms);
int ucg_fix_cluster_switch_set_array(ucg_ctx *ctx, int which, const int *in);
int ucg_fix_cluster_switch_sweep(ucg_ctx *ctx, int begin, int *changed);
int ucg_fix_cluster_switch_finalize(ucg_ctx *ctx);
int ucg_fix_cluster_switch_attempt_local(ucg_ctx *ctx);
int ucg_fix_cluster_switch_attempt_apply(ucg_ctx *ctx);
/* forced = a re-neighbour is forced at the step given to ucg_md_set_timestep; switching = the fix also runs */
int ucg_fix_cluster_switch_due(const ucg_ctx *ctx, int *forced, int *switching);
int ucg_fix_cluster_switch_advance(ucg_ctx *ctx);
int ucg_md_set_timestep(ucg_ctx *ctx, long long ntimestep);

/* ------------------------------------------------------- RanMars on the device
 * the upstream generator behind both fixes, exposed for known-answer tests:
 * n draws of RanMars(seed).uniform() (after its constructor warm-up draw) */
int ucg_ranmars_fill(ucg_ctx *ctx, int seed, long long skip, int n, double *out);

/* ------------------------------------------------------ resident Verlet driver
 * step order of upstream Verlet::setup()/run() with the
 * whole state resident in HBM; used by bench.py and the trajectory-parity tests.
 * use_nve: 0 no integrator, 1 fix nve/ucgld, 2 fix nve/ucgld/wall/hard (its bias_potential keyword
 * comes from ucg_fix_nve_wall_hard_set; being the first fix of the deck its post_force runs before
 * the thermostat's) */
int ucg_md_attach(ucg_ctx *ctx, ucg_pair *pair, int use_nve, int use_langevin, int use_ucgstate);
/* the per-bead hooks that follow the pair kernel as ONE launch, in the reference's order:
 * ucgld/langevin post_force -> ucgstate post_force -> nve/ucgld final_integrate [-> the next
 * step's initial_integrate].  Bit-identical to calling the hooks one by one. */
/* ucg_pair_compute + ucg_md_post_fused(.., fuse_next_initial = 1) as ONE launch (the gather kernel's epilogue);
 * UCG_ERR_UNSUPPORTED = not applicable here (table_ucg_bethe_density, no integrator, option off): use the two calls */
int ucg_md_pair_post(ucg_ctx *ctx, ucg_pair *pair, int use_langevin, int use_ucgstate, int use_nve, int groupbit,
                     long long ntimestep, long long beginstep, long long endstep);
int ucg_md_post_fused(ucg_ctx *ctx, int use_langevin, int use_ucgstate, int use_nve, int fuse_next_initial,
                      int groupbit, long long ntimestep, long long beginstep, long long endstep);
int ucg_md_setup(ucg_ctx *ctx, long long nsteps_planned);
int ucg_md_run(ucg_ctx *ctx, long long nsteps, int thermo_every);
/* the same loop for a caller that owns the output schedule (the run_style of the LAMMPS glue): nsteps steps with energy /
 * virial / lambda temperature evaluated on the LAST one iff ev_on_last (an Output::next step) and on no other */
int ucg_md_run_until(ucg_ctx *ctx, long long nsteps, int ev_on_last);
/* update->beginstep / endstep of `run N start S stop E` (the ramp of fix ucgld/langevin's target, :318-330) when they are
 * not the [ntimestep, ntimestep + nsteps_planned] ucg_md_setup assumes; call after ucg_md_setup */
int ucg_md_set_window(ucg_ctx *ctx, long long beginstep, long long endstep);
/* out[0..15]: ntimestep, nrebuild, nlocal, nghost, list entries, pair errors, ... */
int ucg_md_info(ucg_ctx *ctx, long long *out16);
/* last thermo: eng_vdwl, virial[6], lambda_temp, state-1 population */
int ucg_md_thermo(ucg_ctx *ctx, double *out9);

/* --------------------------------------------------------------- measurement
 * HIP-event timing of the pair kernel on the stream it runs on: enable, run,
 * then read launches and summed milliseconds since the last reset */
int ucg_profile_enable(ucg_ctx *ctx, int on);
int ucg_profile_read(ucg_ctx *ctx, long long *pair_launches, double *pair_ms, int reset);

/* ------------------------------------------------------- on-disk formats (host)
 * SURVEY.md section 8 row f4.  Host code on caller-owned arrays: no context, no GPU.
 * Pointers of ucg_io_atoms that are NULL mean "field absent"; x, v, f are [n][3], image is int[n][3] (ix iy iz),
 * mass is [ntypes+1] indexed by type.  err/errcap receive the message of a failing call (may be NULL/0).
 *
 * ucg_io_dump_write   one snapshot of `dump ID group custom N file <columns>` in the native text format, with
 *                     the keywords ucgstate / ucgl / ucgp of dump_custom.cpp:1672-1688 (+ id mol type mass
 *                     x y z xs ys zs xu yu zu xsu ysu zsu ix iy iz vx vy vz fx fy fz q, and ucgvl ucgml ucgforce =
 *                     the remaining property_atom names of UCG/atom_vec_ucg.cpp:172-181).  `modify` holds
 *                     dump_modify keyword groups, one per line: "thresh <attr> <op> <value>"
 *                     (dump_custom.cpp:1182-1209, :2150-2155; op < <= > >= == !=), "sort id|off",
 *                     "format line|int|float|<M> <fmt>", "boundary pp pp pp".
 * ucg_io_dump_scan / _header / _load   list snapshots, read one header, read every column of one snapshot.
 * ucg_io_read_dump    `read_dump file N <fields> [box|replace|trim|scaled|wrapped yes/no]` with the fields
 *                     x y z vx vy vz q ix iy iz fx fy fz ucgstate ucgl ucgp (read_dump.cpp:1344-1349): atoms are
 *                     matched by ID and overwritten (read_dump.cpp:823-921), trimmed like :925-942; purge/add are
 *                     refused as in the reference (:954).  timestep < 0 = first snapshot of the file.
 * ucg_io_write_data / _data_header / _read_data   data-file sections of atom style ucg: Atoms
 *                     "id mol type q x y z ucgstate ucgl ucgml [ix iy iz]", Velocities "id vx vy vz ucgvl"
 *                     (UCG/atom_vec_ucg.cpp:87-90); reading applies data_atom_post (:145-170).
 * ucg_io_write_restart / _restart_header / _read_restart   binary container with the atom style's restart fields
 *                     ucgstate ucgl ucgml ucgvl ucgp (UCG/atom_vec_ucg.cpp:85) next to id type x v q image mol. */
typedef struct ucg_io_atoms {
  long long n;         /* atoms (on input of the read calls: capacity of the arrays) */
  int ntypes;
  double boxlo[3], boxhi[3];
  int *id, *type, *molecule, *ucgstate, *image;
  double *x, *v, *f, *q, *ucgl, *ucgvl, *ucgml, *ucgp, *ucgforce, *mass;
} ucg_io_atoms;

int ucg_io_dump_write(const char *path, int append, long long timestep, const ucg_io_atoms *a, const char *columns,
                      const char *modify, long long *nwritten, char *err, int errcap);
int ucg_io_dump_scan(const char *path, long long *timesteps, long long *natoms, int cap, int *nfound, char *err,
                     int errcap);
int ucg_io_dump_header(const char *path, long long timestep, long long *found_timestep, long long *natoms,
                       double *boxlo, double *boxhi, char *columns, int columns_cap, char *err, int errcap);
int ucg_io_dump_load(const char *path, long long timestep, long long natoms, int ncolumns, double *values, char *err,
                     int errcap);
int ucg_io_read_dump(const char *path, long long timestep, const char *fields, const char *options, ucg_io_atoms *a,
                     long long *stats4, char *err, int errcap);
int ucg_io_write_data(const char *path, long long timestep, const char *units, const ucg_io_atoms *a, char *err,
                      int errcap);
int ucg_io_data_header(const char *path, long long *natoms, int *ntypes, double *boxlo, double *boxhi,
                       int *has_velocities, char *err, int errcap);
int ucg_io_read_data(const char *path, ucg_io_atoms *a, char *err, int errcap);
int ucg_io_write_restart(const char *path, long long timestep, const ucg_io_atoms *a, char *err, int errcap);
int ucg_io_restart_header(const char *path, long long *timestep, long long *natoms, int *ntypes, double *boxlo,
                          double *boxhi, char *err, int errcap);
int ucg_io_read_restart(const char *path, ucg_io_atoms *a, long long *timestep, char *err, int errcap);

#ifdef __cplusplus
}
#endif
#endif /* UCG_HIP_H */
