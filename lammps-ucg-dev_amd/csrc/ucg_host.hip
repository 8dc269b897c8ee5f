// ucg_host.hip -- host mirrors of a drop-in caller, synchronised lazily (ucg_host_bind / _modified / _sync), and the
// hook-by-hook Verlet driver that measures them (ucg_verlet_hooks_run).
//
// The reference's styles are called hook by hook from upstream Verlet (Pair::compute UCG/pair_table_ucgld.h:22-48,
// FixNVE_UCGLD::initial_integrate / final_integrate UCG/fix_nve_ucgld.h:27-36, Fix_UCGLD_Langevin::post_force
// UCG/fix_ucgld_langevin.h:29-47, FixUCGState::post_force UCG/fix_ucgstate.h:15-23) and work on LAMMPS' host arrays.
// A binding that copies every array a hook touches in and out pays PCIe twice per hook.  Here the DEVICE arrays are
// authoritative between hooks and the host arrays are mirrors with two bit masks of fields (what upstream's KOKKOS
// package does with sync / modified):
//   host_newer   the caller wrote them (ucg_host_modified): uploaded right before the next device hook that reads them
//   dev_newer    a device hook wrote them: downloaded when the caller asks for them (ucg_host_sync) -- on re-neighbouring,
//                thermo and dump steps; an ordinary step moves nothing
// Transfers go through a device staging buffer in the HOST layout (x[n][3] etc.): the AoS <-> double4 repacking is done
// by small kernels, the PCIe copies are plain contiguous ones (fast when the caller's arrays are pinned).
#include <cstring>
#include <vector>

#include "../../include/ucg_hip.h"
#include "ucg_ctx.h"

namespace ucg {

namespace {

constexpr int NB = 256;
inline unsigned nblk(long long n) { return (unsigned) ((n + NB - 1) / NB); }

// staging (host layout) -> double4 record; a3: [n][3] or null (xyz kept), w: [n] or null (w kept)
__global__ __launch_bounds__(NB) void k_merge4(const int n, const double *a3, const double *w, double4 *dst)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  double4 r = dst[i];
  if (a3) {
    r.x = a3[3 * (size_t) i];
    r.y = a3[3 * (size_t) i + 1];
    r.z = a3[3 * (size_t) i + 2];
  }
  if (w) r.w = w[i];
  dst[i] = r;
}

__global__ __launch_bounds__(NB) void k_split4(const int n, const double4 *src, double *a3, double *w)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i >= n) return;
  const double4 r = src[i];
  if (a3) {
    a3[3 * (size_t) i] = r.x;
    a3[3 * (size_t) i + 1] = r.y;
    a3[3 * (size_t) i + 2] = r.z;
  }
  if (w) w[i] = r.w;
}

__global__ __launch_bounds__(NB) void k_state_in(const int n, const int *state, int *meta)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i < n) meta[i] = (meta[i] & 0xFFFF) | ((state[i] & 1) << 16);
}

__global__ __launch_bounds__(NB) void k_state_out(const int n, const int *meta, int *state)
{
  const int i = blockIdx.x * NB + threadIdx.x;
  if (i < n) state[i] = (meta[i] >> 16) & 1;
}

template <typename F>
int guarded_host(ucg_ctx *ctx, F &&fn)
{
  try {
    return fn();
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

// staging: doubles [0, 3n) vector part, [3n, 4n) scalar part; ints behind
double *stage_d(ucg_ctx *ctx, size_t n)
{
  ctx->mirror.stage.reserve(5 * n + 16);
  return ctx->mirror.stage.get();
}

void upload_group(ucg_ctx *ctx, DevBuf<double4> &buf, const double *a3, const double *w)
{
  const size_t n = (size_t) ctx->nlocal;
  if ((!a3 && !w) || n == 0) return;
  double *st = stage_d(ctx, n);
  if (a3) UCG_HIP(hipMemcpyAsync(st, a3, 3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (w) UCG_HIP(hipMemcpyAsync(st + 3 * n, w, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_merge4, dim3(nblk(n)), dim3(NB), 0, ctx->stream, (int) n, a3 ? st : nullptr, w ? st + 3 * n : nullptr, buf.get());
  UCG_HIP(hipGetLastError());
  // the staging buffer is reused by the next group: stream order keeps the copies and kernels apart
}

void download_group(ucg_ctx *ctx, const DevBuf<double4> &buf, double *a3, double *w, double *st)
{
  const size_t n = (size_t) ctx->nlocal;
  if ((!a3 && !w) || n == 0) return;
  hipLaunchKernelGGL(k_split4, dim3(nblk(n)), dim3(NB), 0, ctx->stream, (int) n, buf.get(), a3 ? st : nullptr, w ? st + 3 * n : nullptr);
  UCG_HIP(hipGetLastError());
  if (a3) UCG_HIP(hipMemcpyAsync(a3, st, 3 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (w) UCG_HIP(hipMemcpyAsync(w, st + 3 * n, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
}

}  // namespace

// a device hook is about to read `reads`: upload what the caller has modified since
void mirror_need(ucg_ctx *ctx, unsigned reads)
{
  HostMirror &M = ctx->mirror;
  if (!M.bound) return;
  const unsigned up = reads & M.host_newer;
  if (!up || ctx->nlocal == 0) {
    M.host_newer &= ~up;
    return;
  }
  const size_t n = (size_t) ctx->nlocal;
  upload_group(ctx, ctx->pos4, (up & UCG_F_X) ? M.x : nullptr, (up & UCG_F_UCGL) ? M.ucgl : nullptr);
  upload_group(ctx, ctx->vel4, (up & UCG_F_V) ? M.v : nullptr, (up & UCG_F_UCGVL) ? M.ucgvl : nullptr);
  upload_group(ctx, ctx->frc4, (up & UCG_F_F) ? M.f : nullptr, (up & UCG_F_UCGFORCE) ? M.ucgforce : nullptr);
  if (up & UCG_F_STATE) {
    M.istage.reserve(n + 16);
    UCG_HIP(hipMemcpyAsync(M.istage.get(), M.state, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_state_in, dim3(nblk(n)), dim3(NB), 0, ctx->stream, (int) n, M.istage.get(), ctx->meta.get());
    UCG_HIP(hipGetLastError());
  }
  if (up & UCG_F_NSTATES) UCG_HIP(hipMemcpyAsync(ctx->num_ucgstates.get(), M.nstates, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (up & UCG_F_UCGP) {
    UCG_HIP(hipMemcpyAsync(ctx->ucgp.get(), M.ucgp, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    for (size_t i = 0; i < n && !ctx->ucgp_first_possible; i++) ctx->ucgp_first_possible = !(M.ucgp[i] >= 0.0);
  }
  if (up & UCG_F_SCORES) UCG_HIP(hipMemcpyAsync(ctx->scores.get(), M.scores, 2 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  // pageable sources are staged by the runtime before the call returns; pinned ones must not change until the stream
  // has consumed them: drain (uploads happen on re-neighbouring / after host-side edits only)
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  M.host_newer &= ~up;
  M.uploads++;
}

// a device hook has written `writes`
void mirror_wrote(ucg_ctx *ctx, unsigned writes)
{
  HostMirror &M = ctx->mirror;
  if (!M.bound) return;
  M.dev_newer |= writes;
  M.host_newer &= ~writes;
}

}  // namespace ucg

using namespace ucg;

extern "C" {

int ucg_host_bind(ucg_ctx *ctx, double *x, double *v, double *f, int *ucgstate, int *num_ucgstates, double *ucgl,
                  double *ucgvl, double *ucgp, double *ucgforce, double *ucgsoftmaxscores)
{
  if (!ctx) return UCG_ERR_INVALID;
  HostMirror &M = ctx->mirror;
  const bool any = x || v || f || ucgstate || num_ucgstates || ucgl || ucgvl || ucgp || ucgforce || ucgsoftmaxscores;
  if (any && !(x && v && f && ucgstate && num_ucgstates && ucgl && ucgvl && ucgp && ucgforce && ucgsoftmaxscores)) {
    ctx->err = "ucg_host_bind: give all ten arrays, or none to unbind";
    return UCG_ERR_INVALID;
  }
  if (any && (ctx->comm || ctx->dom_world > 1)) {
    // a decomposed run of the library migrates beads between ranks on the device: the bead count behind the caller's
    // arrays would change under them.  (Under MPI the LAMMPS glue stays in copy mode.)
    ctx->err = "ucg_host_bind: host mirrors are for single-rank contexts";
    return UCG_ERR_UNSUPPORTED;
  }
  const bool was = M.bound;
  M.bound = any;
  M.x = x; M.v = v; M.f = f; M.state = ucgstate; M.nstates = num_ucgstates;
  M.ucgl = ucgl; M.ucgvl = ucgvl; M.ucgp = ucgp; M.ucgforce = ucgforce; M.scores = ucgsoftmaxscores;
  if (!was || !any) {  // a fresh binding: the device holds the state, the caller's arrays are unknown (ucg_atoms_upload
    M.dev_newer = any ? (unsigned) UCG_F_ALL : 0u;  // after the binding makes both sides equal)
    M.host_newer = 0;
  }
  return UCG_OK;
}

int ucg_host_modified(ucg_ctx *ctx, int mask)
{
  if (!ctx || !ctx->mirror.bound || (mask & ~UCG_F_ALL)) return UCG_ERR_INVALID;
  if (mask == 0) return UCG_OK;
  ctx->mirror.host_newer |= (unsigned) mask;
  ctx->mirror.dev_newer &= ~(unsigned) mask;
  return UCG_OK;
}

int ucg_host_sync(ucg_ctx *ctx, int mask)
{
  if (!ctx || !ctx->mirror.bound || (mask & ~UCG_F_ALL)) return UCG_ERR_INVALID;
  return guarded_host(ctx, [&]() -> int {
    HostMirror &M = ctx->mirror;
    const unsigned dn = (unsigned) mask & M.dev_newer;
    const size_t n = (size_t) ctx->nlocal;
    if (dn && n) {
      // one staging area per group so that the copies of all groups can be in flight together
      M.stage.reserve(3 * 4 * n + 64);
      double *st = M.stage.get();
      download_group(ctx, ctx->pos4, (dn & UCG_F_X) ? M.x : nullptr, (dn & UCG_F_UCGL) ? M.ucgl : nullptr, st);
      download_group(ctx, ctx->vel4, (dn & UCG_F_V) ? M.v : nullptr, (dn & UCG_F_UCGVL) ? M.ucgvl : nullptr, st + 4 * n);
      download_group(ctx, ctx->frc4, (dn & UCG_F_F) ? M.f : nullptr, (dn & UCG_F_UCGFORCE) ? M.ucgforce : nullptr, st + 8 * n);
      if (dn & UCG_F_STATE) {
        M.istage.reserve(n + 16);
        hipLaunchKernelGGL(k_state_out, dim3(nblk(n)), dim3(NB), 0, ctx->stream, (int) n, ctx->meta.get(), M.istage.get());
        UCG_HIP(hipGetLastError());
        UCG_HIP(hipMemcpyAsync(M.state, M.istage.get(), n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      }
      if (dn & UCG_F_NSTATES) UCG_HIP(hipMemcpyAsync(M.nstates, ctx->num_ucgstates.get(), n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      if (dn & UCG_F_UCGP) UCG_HIP(hipMemcpyAsync(M.ucgp, ctx->ucgp.get(), n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      if (dn & UCG_F_SCORES) UCG_HIP(hipMemcpyAsync(M.scores, ctx->scores.get(), 2 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      UCG_HIP(hipStreamSynchronize(ctx->stream));
      M.downloads++;
    }
    M.dev_newer &= ~dn;
    return UCG_OK;
  });
}

int ucg_host_status(const ucg_ctx *ctx, int *device_newer, int *host_newer, long long *transfers2)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (device_newer) *device_newer = (int) ctx->mirror.dev_newer;
  if (host_newer) *host_newer = (int) ctx->mirror.host_newer;
  if (transfers2) {
    transfers2[0] = ctx->mirror.uploads;
    transfers2[1] = ctx->mirror.downloads;
  }
  return UCG_OK;
}

/* The package's hooks in the order upstream Verlet::run calls them, one C-ABI call per hook --
 * what a LAMMPS run makes of the USER-UCG/GPU styles, without LAMMPS: initial_integrate -> the re-neighbour decision
 * (taken on the device: the glue's integrator forces LAMMPS' re-neighbouring through Fix::force_reneighbor, so no
 * position leaves the GPU on an ordinary step) -> re-neighbouring [the host mirrors are brought up to date first, as
 * LAMMPS' exchange / borders need them] or the forward halo -> Pair::compute -> post_force hooks in deck order ->
 * final_integrate -> end_of_step; the mirrors are also synchronised every `sync_every` steps (thermo / dump steps).
 * sync_on_reneighbour = 0: the package re-neighbours on its own (device builder, single rank) and LAMMPS' host-side
 * exchange / borders are not served.
 * Returns through `stats4`: re-neighbourings, host synchronisations, uploads, downloads. */
int ucg_verlet_hooks_run(ucg_ctx *ctx, ucg_pair *p, long long nsteps, int use_nve, int use_langevin, int use_ucgstate,
                         int groupbit, int sync_on_reneighbour, int sync_every, long long *stats4)
{
  if (!ctx || !p || p->ctx != ctx || nsteps < 0) return UCG_ERR_INVALID;
  long long nre = 0, nsync = 0;
  if (ctx->endstep <= ctx->ntimestep) {  // no Verlet::setup() (ucg_md_setup) framed this run: it is a run of its own
    ctx->beginstep = ctx->ntimestep;
    ctx->endstep = ctx->ntimestep + nsteps;
  }
  const long long up0 = ctx->mirror.uploads, dn0 = ctx->mirror.downloads;
#define UCG_STEP(call)              \
  do {                              \
    const int rc__ = (call);        \
    if (rc__ != UCG_OK) return rc__; \
  } while (0)
  for (long long s = 0; s < nsteps; s++) {
    ctx->ntimestep++;
    if (use_nve) UCG_STEP(use_nve >= 2 ? ucg_fix_nve_wall_hard_initial(ctx, groupbit) : ucg_fix_nve_initial(ctx, groupbit));
    int due = 0, flag = 0;
    UCG_STEP(ucg_decide_local(ctx, &due, &flag));
    if (due && flag) {
      if (ctx->mirror.bound && sync_on_reneighbour) {  // LAMMPS' pbc / exchange / borders work on the host arrays
        UCG_STEP(ucg_host_sync(ctx, UCG_F_X | UCG_F_V | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGP));
        nsync++;
      }
      UCG_STEP(ucg_neigh_rebuild(ctx));
      nre++;
    } else {
      UCG_STEP(ucg_halo_forward(ctx));
    }
    UCG_STEP(ucg_pair_compute(p, 0, 0, nullptr, nullptr));
    if (use_nve >= 2 && ctx->wall_bias) UCG_STEP(ucg_fix_nve_wall_hard_post_force(ctx, groupbit));
    if (use_langevin) UCG_STEP(ucg_fix_langevin_post_force(ctx, groupbit, ctx->ntimestep, ctx->beginstep, ctx->endstep));
    if (use_ucgstate) UCG_STEP(ucg_fix_ucgstate_post_force(ctx));
    if (use_nve) UCG_STEP(use_nve >= 2 ? ucg_fix_nve_wall_hard_final(ctx, groupbit) : ucg_fix_nve_final(ctx, groupbit));
    if (ctx->mirror.bound && sync_every > 0 && ctx->ntimestep % sync_every == 0) {
      UCG_STEP(ucg_host_sync(ctx, UCG_F_ALL));
      nsync++;
    }
  }
#undef UCG_STEP
  const int rc = ucg_pair_check_errors(p);
  if (stats4) {
    stats4[0] = nre;
    stats4[1] = nsync;
    stats4[2] = ctx->mirror.uploads - up0;
    stats4[3] = ctx->mirror.downloads - dn0;
  }
  return rc;
}

}  // extern "C"
