// ucg_capi.hip -- the extern "C" surface declared in include/ucg_hip.h.
// No exception crosses the ABI: every entry point converts failures into an
// error code + message (the LAMMPS glue forwards it to error->one / error->all).
#include <cmath>
#include <cstring>

#include "../../include/ucg_hip.h"
#include "ucg_ctx.h"

using namespace ucg;

namespace {

template <typename F>
int guarded(ucg_ctx *ctx, F &&fn)
{
  try {
    return fn();
  } catch (const InputError &e) {
    if (ctx) ctx->err = e.msg;
    return UCG_ERR_INPUT;
  } catch (const HipFailure &e) {
    if (ctx) ctx->err = std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what;
    return UCG_ERR_HIP;
  } catch (const std::exception &e) {
    if (ctx) ctx->err = e.what();
    return UCG_ERR_INVALID;
  }
}

int fail(ucg_ctx *ctx, int code, const std::string &msg)
{
  if (ctx) ctx->err = msg;
  return code;
}

// pair entry points: input-error messages are kept on the pair as well, so that a
// host-only pair (no context, see ucg_pair_create_host) can report them
template <typename F>
int guarded_pair(ucg_pair *p, F &&fn)
{
  try {
    return fn();
  } catch (const InputError &e) {
    p->err = e.msg;
    if (p->ctx) p->ctx->err = e.msg;
    return UCG_ERR_INPUT;
  } catch (const HipFailure &e) {
    p->err = std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what;
    if (p->ctx) p->ctx->err = p->err;
    return UCG_ERR_HIP;
  } catch (const std::exception &e) {
    p->err = e.what();
    if (p->ctx) p->ctx->err = p->err;
    return UCG_ERR_INVALID;
  }
}

template <typename T>
void h2d(ucg_ctx *ctx, T *dst, const T *src, size_t n)
{
  if (n) UCG_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
}
template <typename T>
void d2h(ucg_ctx *ctx, T *dst, const T *src, size_t n)
{
  if (n) UCG_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
}
void sync(ucg_ctx *ctx) { UCG_HIP(hipStreamSynchronize(ctx->stream)); }

}  // namespace

AtomsDev ucg_ctx::atoms_dev() const
{
  AtomsDev A;
  A.nlocal = nlocal;
  A.nghost = nghost;
  A.pos4 = pos4.get();
  A.vel4 = vel4.get();
  A.frc4 = frc4.get();
  A.scores = scores.get();
  A.meta = meta.get();
  A.ucgp = ucgp.get();
  A.tag = tag.get();
  A.mask = mask.get();
  A.num_ucgstates = num_ucgstates.get();
  A.ucgml = ucgml.get();
  A.mass = mass.get();
  return A;
}

ListDev ucg_ctx::list_dev() const
{
  ListDev L;
  L.inum = list_inum;
  L.pitch = list_pitch;
  L.maxrow = list_maxrow;
  L.neigh = neigh.get();
  L.numneigh = numneigh.get();
  L.blockflag = nullptr;
  L.blockwant = 0;
  // (measured on the gather kernel: 400 -> 392 us at 1 M beads, 292 MB of entries; 101 -> 105 us at 262 144 beads, 76 MB, which
  // the 256 MB last-level cache otherwise keeps from step to step)
  L.stream_rows = stream_rows >= 0 ? stream_rows : ((list_entries * (long long) sizeof(int) > 192ll * 1024 * 1024) ? 1 : 0);
  L.post = PostDev{};
  return L;
}

void ucg_ctx::ensure_rm_jump(int n)
{
  const int need = (n + RANMARS_CHUNK - 1) / RANMARS_CHUNK + 1;
  if (need <= rm_chunks) return;
  const int want = need + need / 4 + 8;
  std::vector<unsigned int> host((size_t) want * 97);
  ranmars_jump_host(want, host.data());
  rm_jump.reserve(host.size());
  UCG_HIP(hipMemcpyAsync(rm_jump.get(), host.data(), host.size() * sizeof(unsigned int), hipMemcpyHostToDevice, stream));
  UCG_HIP(hipStreamSynchronize(stream));
  rm_chunks = want;
}

extern "C" {

int ucg_abi_version(void) { return UCG_ABI_VERSION; }

int ucg_device_count(void)
{
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int ucg_ctx_create(int device, ucg_ctx **out)
{
  if (!out) return UCG_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UCG_ERR_HIP;
  ucg_ctx *ctx = new ucg_ctx();
  int rc = guarded(ctx, [&]() -> int {
    if (device >= 0) UCG_HIP(hipSetDevice(device));
    UCG_HIP(hipGetDevice(&ctx->device));
    UCG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
    ctx->redout.reserve(64);
    return UCG_OK;
  });
  if (rc) {
    delete ctx;
    return rc;
  }
  *out = ctx;
  return UCG_OK;
}

void ucg_ctx_destroy(ucg_ctx *ctx)
{
  if (!ctx) return;
  (void) hipStreamSynchronize(ctx->stream);
  for (auto ev : ctx->prof_ev) (void) hipEventDestroy(ev);
  comm_destroy(ctx);
  domain_destroy(ctx);
  cluster_destroy(ctx);
  if (ctx->own_stream && ctx->stream) (void) hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *ucg_last_error(const ucg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ucg_ctx_set_stream(ucg_ctx *ctx, void *hip_stream)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    UCG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) UCG_HIP(hipStreamDestroy(ctx->stream));
    // the caller's stream exactly as given; NULL is the legacy default stream (what
    // torch.cuda.current_stream().cuda_stream returns for torch's default stream)
    ctx->stream = (hipStream_t) hip_stream;
    ctx->own_stream = false;
    return UCG_OK;
  });
}

int ucg_ctx_synchronize(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_ctx_set_units(ucg_ctx *ctx, double boltz, double ftm2v, double mvv2e, double dt, const double *special_lj)
{
  if (!ctx) return UCG_ERR_INVALID;
  ctx->boltz = boltz;
  ctx->ftm2v = ftm2v;
  ctx->mvv2e = mvv2e;
  ctx->dt = dt;
  if (special_lj)
    for (int i = 0; i < 4; i++) ctx->special_lj[i] = special_lj[i];
  return UCG_OK;
}

/* ------------------------------------------------------------------ pair */

int ucg_pair_create(ucg_ctx *ctx, int style, ucg_pair **out)
{
  if (!ctx || !out) return UCG_ERR_INVALID;
  if (style < 0 || style > 2) return fail(ctx, UCG_ERR_INVALID, "unknown pair style id");
  ucg_pair *p = new ucg_pair(style);
  p->ctx = ctx;
  *out = p;
  return UCG_OK;
}

int ucg_pair_create_host(int style, double boltz, ucg_pair **out)
{
  if (!out || style < 0 || style > 2) return UCG_ERR_INVALID;
  ucg_pair *p = new ucg_pair(style);
  p->ctx = nullptr;
  p->host_boltz = boltz;
  *out = p;
  return UCG_OK;
}

const char *ucg_pair_last_error(const ucg_pair *p)
{
  if (!p) return "null pair";
  if (p->ctx && p->err.empty()) return p->ctx->err.c_str();
  return p->err.c_str();
}

void ucg_pair_destroy(ucg_pair *p)
{
  if (!p) return;
  if (p->ctx) {
    (void) hipStreamSynchronize(p->ctx->stream);
    if (p->ctx->md_pair == p) p->ctx->md_pair = nullptr;
  }
  delete p;
}

int ucg_pair_settings(ucg_pair *p, int narg, const char *const *arg)
{
  if (!p) return UCG_ERR_INVALID;
  return guarded_pair(const_cast<ucg_pair *>(p), [&]() -> int {
    p->uploaded = false;
    p->model.settings(narg, arg);
    return UCG_OK;
  });
}

int ucg_pair_coeff(ucg_pair *p, int ntypes, int narg, const char *const *arg)
{
  if (!p) return UCG_ERR_INVALID;
  return guarded_pair(const_cast<ucg_pair *>(p), [&]() -> int {
    p->uploaded = false;
    p->model.coeff(ntypes, narg, arg);
    return UCG_OK;
  });
}

int ucg_pair_init(ucg_pair *p, int ntypes, double T)
{
  if (!p) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded_pair(p, [&]() -> int {
    PairModel &M = p->model;
    if (!(T > 0.0) || !std::isfinite(T))
      throw InputError{"pair style needs the thermostat target temperature (Fix::extract(\"t_target\")); none given"};
    M.init(ntypes, T, ctx ? ctx->boltz : p->host_boltz);
    if (!ctx) return UCG_OK;  // host-only pair: tables and maps stay on the host
    if (M.n_actual > UCG_MAX_ACTUAL) return fail(ctx, UCG_ERR_UNSUPPORTED, "more actual types than the GPU kernels stage (UCG_MAX_ACTUAL)");
    const int na1 = M.n_actual + 1, nt = M.n_formal + 1, ms = M.max_states;
    for (int t = 1; t <= M.n_actual; t++)
      if (M.n_states_per_type[(size_t) t] != 2)
        return fail(ctx, UCG_ERR_UNSUPPORTED,
                    "GPU UCG styles cover 2-state types only (the reference cannot configure 1-state types either: "
                    "formal type 0 is rejected by coeff())");
    // device tables: only those reachable through tabindex after init_one
    p->tabmap.assign(M.tables.size(), -1);
    std::vector<int> order;
    std::vector<int> pairtab((size_t) na1 * na1 * 4, 0);
    for (int ti = 1; ti <= M.n_actual; ti++)
      for (int tj = 1; tj <= M.n_actual; tj++)
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 2; b++) {
            const int fi = M.formal_from_actual[(size_t) ti * ms + a], fj = M.formal_from_actual[(size_t) tj * ms + b];
            if (fi > ntypes || fj > ntypes) throw InputError{"formal type exceeds atom->ntypes"};
            const int id = M.tabindex[(size_t) fi * nt + fj];
            if (p->tabmap[(size_t) id] < 0) {
              p->tabmap[(size_t) id] = (int) order.size();
              order.push_back(id);
            }
            pairtab[((size_t) ti * na1 + tj) * 4 + a * 2 + b] = p->tabmap[(size_t) id];
          }
    const int ntab = (int) order.size();
    if (ntab > UCG_MAX_TABLES) return fail(ctx, UCG_ERR_UNSUPPORTED, "more tables than the GPU kernels stage (UCG_MAX_TABLES)");
    // BITMAP: two records per bin, {e, de, f, df} and {rsq, drsq, -, -}; the table parameters carry the bit masks
    const bool bitmap = (M.tabstyle == BITMAP);
    const int tl = bitmap ? 2 * (1 << M.tablength) : M.tablength, tlm1 = tl - 1;
    std::vector<double4> tab((size_t) ntab * tl), par((size_t) ntab);
    for (int d = 0; d < ntab; d++) {
      const Table &tb = M.tables[(size_t) order[(size_t) d]];
      par[(size_t) d] = make_double4(tb.innersq, tb.delta, tb.invdelta, tb.deltasq6);
      if (bitmap) {
        par[(size_t) d] = make_double4(tb.innersq, (double) tb.nmask, (double) tb.nshiftbits, 0.0);
        for (int k = 0; k < tl / 2; k++) {
          tab[(size_t) d * tl + 2 * k] = make_double4(tb.e[(size_t) k], tb.de[(size_t) k], tb.f[(size_t) k], tb.df[(size_t) k]);
          tab[(size_t) d * tl + 2 * k + 1] = make_double4(tb.rsq[(size_t) k], tb.drsq[(size_t) k], 0.0, 0.0);
        }
        continue;
      }
      for (int k = 0; k < tl; k++) {
        double4 v = make_double4(0, 0, 0, 0);
        if (M.tabstyle == LOOKUP) {
          if (k < tlm1) v = make_double4(tb.e[(size_t) k], tb.f[(size_t) k], 0, 0);
        } else if (M.tabstyle == LINEAR) {
          v.x = tb.e[(size_t) k];
          v.z = tb.f[(size_t) k];
          if (k < tlm1) {
            v.y = tb.de[(size_t) k];
            v.w = tb.df[(size_t) k];
          }
        } else {
          v = make_double4(tb.e[(size_t) k], tb.f[(size_t) k], tb.e2[(size_t) k], tb.f2[(size_t) k]);
        }
        tab[(size_t) d * tl + k] = v;
      }
    }
    std::vector<double> cutsq((size_t) na1 * na1, 0.0), mu((size_t) na1 * 2, 0.0), prior((size_t) na1 * 2, 0.0);
    for (int ti = 1; ti <= M.n_actual; ti++) {
      for (int tj = 1; tj <= M.n_actual; tj++)
        cutsq[(size_t) ti * na1 + tj] = M.cutsq[(size_t) ti * nt + tj];  // indexed by ACTUAL types, as the reference does (:213)
      for (int s = 0; s < 2; s++) {
        mu[(size_t) ti * 2 + s] = M.chem_pot[(size_t) M.formal_from_actual[(size_t) ti * ms + s]];
        prior[(size_t) ti * 2 + s] = M.prior_prob_from_type[(size_t) ti * ms + s];
      }
    }
    std::vector<int> dflags((size_t) na1 * 2, 0);
    std::vector<double> dpar((size_t) na1 * 2, 0.0);
    for (int ti = 1; ti <= M.n_actual; ti++) {
      dflags[(size_t) ti * 2 + 0] = M.use_density[(size_t) ti];
      dflags[(size_t) ti * 2 + 1] = M.use_state_entropy[(size_t) ti];
      dpar[(size_t) ti * 2 + 0] = M.cv_thresholds[(size_t) ti];
      dpar[(size_t) ti * 2 + 1] = M.threshold_radii[(size_t) ti];
    }
    p->d_densflags.reserve(dflags.size());
    p->d_denspar.reserve(dpar.size());
    h2d(ctx, p->d_densflags.get(), dflags.data(), dflags.size());
    h2d(ctx, p->d_denspar.get(), dpar.data(), dpar.size());
    p->d_tab.reserve(tab.size());
    p->d_tabpar.reserve(par.size());
    p->d_pairtab.reserve(pairtab.size());
    p->d_cutsq.reserve(cutsq.size());
    p->d_mu.reserve(mu.size());
    p->d_prior_type.reserve(prior.size());
    p->d_err.reserve(4);
    p->d_evout.reserve(8);
    h2d(ctx, p->d_tab.get(), tab.data(), tab.size());
    h2d(ctx, p->d_tabpar.get(), par.data(), par.size());
    h2d(ctx, p->d_pairtab.get(), pairtab.data(), pairtab.size());
    h2d(ctx, p->d_cutsq.get(), cutsq.data(), cutsq.size());
    h2d(ctx, p->d_mu.get(), mu.data(), mu.size());
    h2d(ctx, p->d_prior_type.get(), prior.data(), prior.size());
    UCG_HIP(hipMemsetAsync(p->d_err.get(), 0, 4 * sizeof(int), ctx->stream));
    sync(ctx);

    PairDev &D = p->dev;
    D.style = M.style;
    D.tabstyle = M.tabstyle;
    D.tablength = tl;
    D.tlm1 = tlm1;
    D.n_actual = M.n_actual;
    D.ntab = ntab;
    // one 1024-lane workgroup per CU owns the LDS: 160 KB minus the small static arrays
    D.tab_in_lds = ((size_t) ntab * tl * sizeof(double4) <= 152 * 1024) ? 1 : 0;
    D.pseudo_flag = M.pseudo_flag;
    D.prior_flag = M.prior_flag;
    D.method_flag = M.method_flag;
    D.tab = p->d_tab.get();
    D.tabpar = p->d_tabpar.get();
    D.pairtab = p->d_pairtab.get();
    // one actual type whose mixed-state tables are one table (tabindex is symmetric after init_one, App. B #26): what the
    // ONETYPE kernels assume at compile time
    D.onetype_same10 = (M.n_actual == 1 && pairtab[((size_t) 1 * na1 + 1) * 4 + 1] == pairtab[((size_t) 1 * na1 + 1) * 4 + 2]) ? 1 : 0;
    D.cutsq = p->d_cutsq.get();
    D.mu = p->d_mu.get();
    D.prior_type = p->d_prior_type.get();
    D.dens_flags = p->d_densflags.get();
    D.dens_par = p->d_denspar.get();
    D.dens_as_shipped = ctx->density_proximity_as_shipped ? 1 : 0;
    D.kT = M.kT;
    D.rkT = 1.0 / M.kT;
    for (int i = 0; i < 4; i++) D.special_lj[i] = ctx->special_lj[i];
    {
      // FAST kernels: same arithmetic, less work (see ucg_pair.hip).  Conditions checked here.
      bool fast = true;
      for (int d = 1; d < ntab; d++)
        if (std::memcmp(&par[(size_t) d], &par[0], sizeof(double4)) != 0) fast = false;
      for (int i = 0; i < 4; i++)
        if (ctx->special_lj[i] != 1.0) fast = false;
      unsigned long long kb;
      std::memcpy(&kb, &M.kT, sizeof kb);
      const unsigned long long mant = kb & 0xFFFFFFFFFFFFFull, ex = (kb >> 52) & 0x7FF;
      if (mant == 0xFFFFFFFFFFFFFull || ex < 1023 - 200 || ex > 1023 + 200 || (kb >> 63)) fast = false;
      if (ctx->force_generic_kernels || bitmap) fast = false;
      D.kT_pow2 = (mant == 0 && ex > 1023 - 200 && ex < 1023 + 200 && !(kb >> 63)) ? 1 : 0;
      D.fast = fast ? 1 : 0;
      D.fast_stride = 2 * ntab + 1;
      const size_t nslots = (size_t) tl * D.fast_stride;
      std::vector<double2> tf(nslots + 2, make_double2(0, 0));
      for (int k = 0; k < tl; k++)
        for (int d = 0; d < ntab; d++) {
          const double4 v = tab[(size_t) d * tl + k];
          tf[(size_t) k * D.fast_stride + 2 * d] = make_double2(v.x, v.y);
          tf[(size_t) k * D.fast_stride + 2 * d + 1] = make_double2(v.z, v.w);
        }
      p->d_tab_fast.reserve((nslots + 2) / 2 + 1);
      h2d(ctx, (double2 *) p->d_tab_fast.get(), tf.data(), nslots + 2);
      sync(ctx);
      D.tab_fast = p->d_tab_fast.get();
      const size_t bytes = fast ? ((nslots + 1) / 2) * sizeof(double4) : (size_t) ntab * tl * sizeof(double4);
      D.tab_in_lds = (bytes <= 152 * 1024 && !bitmap) ? 1 : 0;  // bitmapped bins: hundreds of KB, read through L2
      // own-bead staging: 36 bytes per bead of the workgroup behind the tables, if the 160 KB allow it
      if (bitmap && ctx->gather_slots > 1)
        return fail(ctx, UCG_ERR_UNSUPPORTED, "bitmap tables run with one lane per bead (option gather_slots 0 or 1)");
      int slots0 = ctx->gather_slots > 0 ? ctx->gather_slots : 1;
      // option pair_vrow (off by default): both gather styles on virtual rows (ucg_pair_vrow.hip) when the tables, 512 own beads
      // and their fixed-point accumulators fit the LDS next to the static model arrays.  Decided here, once: it fixes how
      // a bead's terms are summed (fixed sums, ucg_pair_sum_fixed), which callers comparing bits need to know.
      p->vrow = false;
      p->vr_gen = -1;
      for (int c = 0; c < 3; c++) {
        D.sum_sc[c] = 1.0;
        D.sum_dec[c] = std::ldexp(1.0, -38);
      }
      D.sum_rsq_safe = 1.0e300;
      if (ctx->pair_vrow && (M.style == STYLE_UCGLD || M.style == STYLE_BETHE) && fast && D.tab_in_lds && !bitmap &&
          !ctx->fma_contract) {
        PairDev probe = D;
        probe.fast = 1;
        p->vrow = vrow_lds_bytes(probe) + 4608 <= 160 * 1024;
      }
      if (p->vrow) {
        // the power-of-two units of the three fields -- the rule of the oracle's orc_pair_sum_scales, on the same table
        // values -- and the r^2 beyond which the magic-constant image is valid
        const int nent = (M.tabstyle == LOOKUP) ? M.tablength - 1 : M.tablength;
        double fref = 0.0, uref = 0.0;
        for (int d = 0; d < ntab; d++) {
          const Table &tb = M.tables[(size_t) order[(size_t) d]];
          // the knots of the upper three quarters of the table's r^2 range
          const double from = tb.innersq + 0.25 * ((nent - 1) * tb.delta);
          for (int k = 0; k < nent; k++) {
            const double rs = tb.innersq + k * tb.delta;
            if (!(rs >= from)) continue;
            const double fv = std::fabs(tb.f[(size_t) k]) * std::sqrt(rs > 0.0 ? rs : 0.0), ue = std::fabs(tb.e[(size_t) k]);
            if (fv > fref) fref = fv;
            if (ue > uref) uref = ue;
          }
        }
        const double sref = uref / M.kT;
        const int ex[3] = {(fref > 0.0 && std::isfinite(fref)) ? 4 - std::ilogb(fref) : 0,
                           (uref > 0.0 && std::isfinite(uref)) ? 4 - std::ilogb(uref) : 0,
                           (sref > 0.0 && std::isfinite(sref)) ? 4 - std::ilogb(sref) : 0};
        for (int c = 0; c < 3; c++) {
          D.sum_sc[c] = std::ldexp(1.0, ex[c]);
          D.sum_dec[c] = std::ldexp(1.0, -38 - ex[c]);
        }
        // terms: weight (<= 2.8^2 for |lambda - 0.5| <= 2.3; Bethe: probabilities) x {F = (f/r) r, an energy difference
        // (<= 2 max|e|), an energy / kT}; walk the grid down from the cutoff while all of them stay below 8192 units
        const double wmax = 8.0;
        int ksafe = nent;
        for (int k = nent - 1; k >= 0; k--) {
          double mag = 0.0;
          for (int d = 0; d < ntab; d++) {
            const Table &tb = M.tables[(size_t) order[(size_t) d]];
            const double rs = tb.innersq + k * tb.delta;
            mag = std::fmax(mag, std::fabs(tb.f[(size_t) k]) * std::sqrt(rs > 0.0 ? rs : 0.0) * D.sum_sc[0]);
            mag = std::fmax(mag, 2.0 * std::fabs(tb.e[(size_t) k]) * D.sum_sc[1]);
            mag = std::fmax(mag, 2.0 * std::fabs(tb.e[(size_t) k]) / M.kT * D.sum_sc[2]);
          }
          if (!(wmax * mag < 8192.0)) break;
          ksafe = k;
        }
        const Table &t0 = M.tables[(size_t) order[0]];
        const int kk = ksafe + 2;  // two knots of margin for the interpolant between knots
        if (kk < nent) D.sum_rsq_safe = t0.innersq + kk * t0.delta;
      }
      D.hot_type = 0;
      D.hot_ent = 0;
      D.hot_k0 = -1;
      D.tab_hot = nullptr;
      p->host_tab.clear();
      p->hot_checked = -1;
      if (fast && !D.tab_in_lds && !bitmap && M.n_actual > 1 && ctx->hot_block) {
        p->host_tab = tab;
        p->host_pairtab = pairtab;
      }
      // kind blocks (KindsDev): the tables of every (row type, neighbour type) kind as one
      // compact FAST block for the lanes that read their tables through L1 / L2
      p->kinds = false;
      D.kinds = KindsDev{nullptr, nullptr};
      D.tcache = nullptr;
      if (fast && !D.tab_in_lds && !bitmap && M.n_actual > 1 && ctx->kind_blocks) {
        const int na1 = M.n_actual + 1;
        std::vector<int2> dir((size_t) na1 * na1, make_int2(0, 0));
        std::vector<double4> blocks;
        for (int a = 1; a <= M.n_actual; a++)
          for (int b = 1; b <= M.n_actual; b++) {
            const int *pt = &pairtab[((size_t) a * na1 + b) * 4];
            const bool same = pt[1] == pt[2];
            const int nt = same ? 3 : 4, stride = 2 * nt + 1;
            const int ids[4] = {pt[0], pt[1], same ? pt[3] : pt[2], pt[3]};
            const size_t ent4 = ((size_t) tl * stride + 1) / 2;
            std::vector<double2> blk(ent4 * 2, make_double2(0, 0));
            for (int k = 0; k < tl; k++)
              for (int q = 0; q < nt; q++) {
                const double4 v = tab[(size_t) ids[q] * tl + k];
                blk[(size_t) k * stride + 2 * q] = make_double2(v.x, v.y);
                blk[(size_t) k * stride + 2 * q + 1] = make_double2(v.z, v.w);
              }
            dir[(size_t) a * na1 + b] = make_int2((int) blocks.size(), nt);
            const double4 *b4 = reinterpret_cast<const double4 *>(blk.data());
            blocks.insert(blocks.end(), b4, b4 + ent4);
          }
        p->d_kind_tab.reserve(blocks.size() + 1);
        p->d_kind_dir.reserve(dir.size());
        h2d(ctx, p->d_kind_tab.get(), blocks.data(), blocks.size());
        h2d(ctx, p->d_kind_dir.get(), dir.data(), dir.size());
        sync(ctx);
        p->kinds = true;
        D.kinds.kind_tab = p->d_kind_tab.get();
        D.kinds.kind_dir = p->d_kind_dir.get();
      }
      if (fast && !D.tab_in_lds && !bitmap && M.n_actual == 1 && ctx->hot_block) {
        // one actual type, tables too long for the LDS: the knots of the far end of the r^2 grid -- as many as fit next
        // to 1024 staged beads -- are kept there (PairDev::hot_k0); the window starts at an even knot so that it is a
        // double4-aligned piece of the FAST layout
        const size_t budget = 160 * 1024 - 6 * 1024 - (ctx->stage_own ? (size_t) 1024 * (M.style == STYLE_BETHE ? 44 : 36) : 0) - 1024;
        int nk = (int) (budget / ((size_t) D.fast_stride * sizeof(double2)));
        int k0 = tl - nk;
        if (k0 < 0) k0 = 0;
        k0 += k0 & 1;
        if (tl - k0 >= 64) {
          D.hot_k0 = k0;
          D.hot_ent = (int) (((size_t) (tl - k0) * D.fast_stride + 1) / 2);
          D.tab_hot = p->d_tab_fast.get() + ((size_t) k0 * D.fast_stride) / 2;
        }
      }
      D.gather_slots = slots0;
      p->tab_lds_bytes = D.tab_in_lds ? bytes : (size_t) D.hot_ent * sizeof(double4);
      const size_t own = (size_t) (1024 / slots0) * (M.style == STYLE_BETHE ? 44 : 36);  // pair_own_bytes: + ucgp for table_ucg_bethe
      const size_t used = p->tab_lds_bytes + own + 4 * 1024;  // + the static model arrays (3.5 KB at most)
      D.stage_own = (ctx->stage_own && used <= 160 * 1024) ? 1 : 0;
      D.stage_own_allowed = ctx->stage_own ? 1 : 0;
    }
    if (M.style == STYLE_BETHE && M.prior_flag == PRIOR_CHEMPOT_NOISE)
      return fail(ctx, UCG_ERR_UNSUPPORTED,
                  "prior chemical_potential noise draws RanMars inside the neighbour loop in list order "
                  "(UCG/pair_table_ucg_bethe.cpp:187,235); not reproducible on a parallel device");
    p->uploaded = true;
    return UCG_OK;
  });
}

double ucg_pair_cut(const ucg_pair *p, int i, int j)
{
  if (!p || !p->model.initialized) return -1.0;
  const int nt = p->model.n_formal + 1;
  if (i < 1 || j < 1 || i >= nt || j >= nt) return -1.0;
  return p->model.tables[(size_t) p->model.tabindex[(size_t) i * nt + j]].cut;
}

double ucg_pair_cutforce(const ucg_pair *p) { return (p && p->model.initialized) ? p->model.cutforce : -1.0; }

int ucg_pair_single(const ucg_pair *p, int itype, int jtype, double rsq, double factor_lj, double *fforce, double *energy)
{
  if (!p || !fforce || !energy) return UCG_ERR_INVALID;
  return guarded_pair(const_cast<ucg_pair *>(p), [&]() -> int {
    if (!p->model.initialized) throw InputError{"pair style not initialised"};
    const int nt = p->model.n_formal + 1;
    if (itype < 1 || jtype < 1 || itype >= nt || jtype >= nt) throw InputError{"type out of range in single()"};
    *energy = p->model.single(itype, jtype, rsq, factor_lj, *fforce);
    return UCG_OK;
  });
}

int ucg_pair_gather_slots(const ucg_pair *p) { return (p && p->uploaded) ? p->dev.gather_slots : 1; }
int ucg_pair_sum_fixed(const ucg_pair *p) { return (p && p->uploaded && p->vrow) ? 1 : 0; }

int ucg_pair_table_count(const ucg_pair *p) { return p ? (int) p->model.tables.size() : -1; }

int ucg_pair_table_params(const ucg_pair *p, int m, double *out5)
{
  if (!p || !out5 || m < 0 || m >= (int) p->model.tables.size()) return UCG_ERR_INVALID;
  const Table &tb = p->model.tables[(size_t) m];
  out5[0] = tb.innersq;
  out5[1] = tb.delta;
  out5[2] = tb.invdelta;
  out5[3] = tb.deltasq6;
  out5[4] = tb.cut;
  return UCG_OK;
}

int ucg_pair_table_array(const ucg_pair *p, int m, const char *which, double *out, int cap)
{
  if (!p || !which || m < 0 || m >= (int) p->model.tables.size()) return -1;
  const Table &tb = p->model.tables[(size_t) m];
  const std::vector<double> *v = nullptr;
  const std::string w = which;
  if (w == "rsq") v = &tb.rsq;
  else if (w == "e") v = &tb.e;
  else if (w == "f") v = &tb.f;
  else if (w == "de") v = &tb.de;
  else if (w == "df") v = &tb.df;
  else if (w == "e2") v = &tb.e2;
  else if (w == "f2") v = &tb.f2;
  else if (w == "drsq") v = &tb.drsq;
  else if (w == "rfile") v = &tb.rfile;
  else if (w == "efile") v = &tb.efile;
  else if (w == "ffile") v = &tb.ffile;
  else if (w == "e2file") v = &tb.e2file;
  else if (w == "f2file") v = &tb.f2file;
  else return -1;
  const int n = (int) v->size();
  if (out && cap >= n) std::memcpy(out, v->data(), sizeof(double) * (size_t) n);
  return n;
}

int ucg_pair_tabindex(const ucg_pair *p, int *out, int cap)
{
  if (!p) return -1;
  const int n = (int) p->model.tabindex.size();
  if (out && cap >= n) std::memcpy(out, p->model.tabindex.data(), sizeof(int) * (size_t) n);
  return n;
}

static int pair_compute_impl(ucg_pair *p, int eflag, int vflag, double *eng_vdwl, double *virial, int part,
                             const PostDev *post = nullptr);

namespace {
__global__ __launch_bounds__(256) void k_type_hist(int n, const int *meta, int *hist)
{
  __shared__ int s_h[UCG_MAX_ACTUAL + 1];
  if (threadIdx.x <= UCG_MAX_ACTUAL) s_h[threadIdx.x] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int t = UCG_META_TYPE(meta[i]);
    if (t <= UCG_MAX_ACTUAL) atomicAdd(&s_h[t], 1);
  }
  __syncthreads();
  if (threadIdx.x <= UCG_MAX_ACTUAL && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
}  // namespace

// Tables read through L1 / L2 (several actual types): the block {t00, t01 = t10, t11} of the pairs of the most populous
// actual type with itself goes to LDS (PairDev::hot_type).  Chosen again after every re-neighbouring (fix cluster_switch
// changes the populations there); `own_bytes` is what the calling kernel stages next to it.
static void choose_hot_block(ucg_ctx *ctx, ucg_pair *p, size_t own_bytes)
{
  if (p->host_tab.empty() || p->hot_checked == ctx->list_gen) return;
  p->hot_checked = ctx->list_gen;
  PairDev &D = p->dev;
  const int na1 = D.n_actual + 1, tl = D.tablength;
  const int ent = (tl * 7 + 1) / 2;
  int best = 0;
  if ((size_t) ent * sizeof(double4) + own_bytes + 6 * 1024 <= 160 * 1024 && ctx->nlocal > 0) {
    p->d_typehist.reserve(UCG_MAX_ACTUAL + 1);
    UCG_HIP(hipMemsetAsync(p->d_typehist.get(), 0, (UCG_MAX_ACTUAL + 1) * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_type_hist, dim3(256), dim3(256), 0, ctx->stream, ctx->nlocal, ctx->meta.get(), p->d_typehist.get());
    int h[UCG_MAX_ACTUAL + 1];
    d2h(ctx, h, p->d_typehist.get(), UCG_MAX_ACTUAL + 1);
    sync(ctx);
    for (int a = 1; a <= D.n_actual && a <= UCG_MAX_ACTUAL; a++) {
      const int *pt = &p->host_pairtab[((size_t) a * na1 + a) * 4];
      if (pt[1] != pt[2]) continue;  // (never after init_one: tabindex is symmetric)
      if (h[a] > (best ? h[best] : 0)) best = a;
    }
  }
  if (best == D.hot_type) return;
  if (best) {
    const int *pt = &p->host_pairtab[((size_t) best * na1 + best) * 4];
    const int ids[3] = {pt[0], pt[1], pt[3]};
    std::vector<double2> tf((size_t) ent * 2, make_double2(0, 0));
    for (int k = 0; k < tl; k++)
      for (int q = 0; q < 3; q++) {
        const double4 v = p->host_tab[(size_t) ids[q] * tl + k];
        tf[(size_t) k * 7 + 2 * q] = make_double2(v.x, v.y);
        tf[(size_t) k * 7 + 2 * q + 1] = make_double2(v.z, v.w);
      }
    sync(ctx);  // no launch may still read the block that is replaced
    p->d_tab_hot.reserve((size_t) ent);
    h2d(ctx, (double2 *) p->d_tab_hot.get(), tf.data(), tf.size());
    sync(ctx);
    D.tab_hot = p->d_tab_hot.get();
    D.hot_ent = ent;
  } else {
    D.tab_hot = nullptr;
    D.hot_ent = 0;
  }
  D.hot_type = best;
}

int ucg_pair_compute(ucg_pair *p, int eflag, int vflag, double *eng_vdwl, double *virial)
{
  return pair_compute_impl(p, eflag, vflag, eng_vdwl, virial, 0);
}

int ucg_pair_compute_part(ucg_pair *p, int part)
{
  if (part != 1 && part != 2) return UCG_ERR_INVALID;
  return pair_compute_impl(p, 0, 0, nullptr, nullptr, part);
}

// part 0: all beads; 1: only the workgroups none of whose beads has a ghost neighbour (they need no
// halo); 2: the others.  1 then 2 write exactly what 0 writes.
static int pair_compute_impl(ucg_pair *p, int eflag, int vflag, double *eng_vdwl, double *virial, int part,
                             const PostDev *post)
{
  if (!p || !p->ctx) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded(ctx, [&]() -> int {
    if (!p->uploaded) return fail(ctx, UCG_ERR_INVALID, "ucg_pair_compute before ucg_pair_init");
    if (ctx->list_inum != ctx->nlocal) return fail(ctx, UCG_ERR_INVALID, "neighbour list does not match the resident atoms");
    const bool ev = (eflag || vflag);
    mirror_need(ctx, UCG_F_X | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGP);
    if (ctx->list_from_builder)
      for (int i = 1; i < 4; i++)
        if (ctx->special_lj[i] != 1.0)
          return fail(ctx, UCG_ERR_UNSUPPORTED, "the device list builder knows no bond topology (no special-bond bits): "
                                                "special_lj other than 1 needs a list uploaded with ucg_neigh_upload_full");
    if (ctx->gather_slots == 0 && p->dev.tabstyle != BITMAP) {
      // auto: the kernel's time is (rounds of workgroups over the 256 CUs) x (lifetime of one workgroup),
      // the latter ~ row length / lanes per bead + a fixed part; pick the lanes per bead that minimise it
      // (matches the measured order at 125 k / 250 k / 500 k / 1 M beads per GPU)
      int slots = 1;
      double best = 1e300;
      for (int s2 : {1, 2, 4, 8, 16}) {
        const long long blocks = ((long long) ctx->nlocal * s2 + 1023) / 1024;
        const double cost = (double) ((blocks + 255) / 256) * (73.0 / s2 + 8.0);
        if (cost < best - 1e-9) {
          best = cost;
          slots = s2;
        }
      }
      p->dev.gather_slots = slots;
      const size_t own = (size_t) (1024 / slots) * (p->model.style == STYLE_BETHE ? 44 : 36);
      p->dev.stage_own = (ctx->stage_own && p->tab_lds_bytes + own + 4 * 1024 <= 160 * 1024) ? 1 : 0;
    }
    if (!p->host_tab.empty()) {
      // density: its pass 2 stages 1024 beads; the gather kernels 1024 / lanes-per-bead
      const size_t own = p->model.style == STYLE_BETHE_DENSITY ? (size_t) 1024 * 36
                                                               : (size_t) (1024 / p->dev.gather_slots) * (p->model.style == STYLE_BETHE ? 44 : 36);
      choose_hot_block(ctx, p, ctx->stage_own ? own : 0);
      const size_t hot = (size_t) p->dev.hot_ent * sizeof(double4);
      p->tab_lds_bytes = hot;
      p->dev.stage_own = (ctx->stage_own && hot + own + 4 * 1024 <= 160 * 1024) ? 1 : 0;
    }
    const int nb = p->vrow ? vrow_blocks(ctx->nlocal) : pair_gather_blocks(ctx->nlocal, p->dev.gather_slots);
    if (ev) p->d_evpart.reserve((size_t) nb * 8 + 8);
    if (p->vrow && p->vr_gen != ctx->list_gen && ctx->nlocal > 0) {
      // the virtual rows of this list (one launch, no host round trip: the buffers are sized from the longest full row)
      if (ctx->list_maxrow > vrow_maxrow())
        return fail(ctx, UCG_ERR_UNSUPPORTED, "a neighbour row has more entries than the virtual-row builder takes (128); "
                                              "set option pair_vrow 0 before ucg_pair_init");
      int cap = vrow_capacity(ctx->list_maxrow);
      cap = cap > 32 ? 32 : cap;  // (what the builder's LDS staging takes; a longer list is reported by the builder)
      const int pitch = nb * 1024;
      if (cap > p->vr_cap || pitch > p->vr_pitch) {
        p->vr_cap = cap > p->vr_cap ? cap : p->vr_cap;
        p->vr_pitch = pitch > p->vr_pitch ? pitch : p->vr_pitch;
        p->d_vr_ent.reserve((size_t) p->vr_pitch * (size_t) p->vr_cap * (size_t) vrow_lists());
        p->d_vr_lanemeta.reserve((size_t) p->vr_pitch * (size_t) vrow_lists());
      }
      UCG_HIP(launch_vrow_build(p->dev, ctx->atoms_dev(), ctx->list_dev(), ctx->skin, p->d_vr_ent.get(),
                                (size_t) p->vr_pitch * (size_t) p->vr_cap, p->vr_cap, p->vr_pitch, p->d_vr_lanemeta.get(),
                                p->d_err.get(), ctx->stream));
      p->vr_gen = ctx->list_gen;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->prof_on) {
      UCG_HIP(hipEventCreate(&e0));
      UCG_HIP(hipEventCreate(&e1));
      UCG_HIP(hipEventRecord(e0, ctx->stream));
    }
    if (part != 0 && p->model.style == STYLE_BETHE_DENSITY)
      return fail(ctx, UCG_ERR_INVALID, "ucg_pair_compute_part covers the gather styles (table_ucgld, table_ucg_bethe)");
    if (p->model.style == STYLE_BETHE_DENSITY) {
      if (ctx->nghost > 0 && !ctx->ghost_src_valid)
        return fail(ctx, UCG_ERR_UNSUPPORTED,
                    "table_ucg_bethe_density needs the ghost -> owner map (ucg_ghosts_upload or the device builder); "
                    "decomposed runs are not covered yet");
      const size_t nall = (size_t) ctx->nlocal + ctx->nghost;
      p->d_prior.reserve(nall + 1);
      p->d_cv.reserve(nall + 1);
      p->d_partial.reserve((size_t) ctx->nlocal + 1);
      p->d_evpart.reserve((size_t) density_evpart_doubles(ctx->nlocal));
      p->dev.tcache = nullptr;
      if (ctx->density_tcache && ctx->list_maxrow > 0) {
        p->d_tcache.reserve((size_t) ctx->list_pitch * (size_t) ctx->list_maxrow);
        p->dev.tcache = p->d_tcache.get();
      }
      UCG_HIP(launch_density(p->dev, ctx->atoms_dev(), ctx->list_dev(), ctx->ghost_src.get(), ev, p->d_prior.get(),
                             p->d_partial.get(), p->d_cv.get(), p->d_evpart.get(), p->d_evout.get(), p->d_err.get(),
                             ctx->stream));
    } else {
      ListDev L = ctx->list_dev();
      if (part != 0) {
        const int bslots = p->vrow ? 1024 / vrow_beads() : p->dev.gather_slots;  // beads per workgroup = 1024 / bslots
        if (p->blockflag_build != ctx->list_gen || p->blockflag_slots != bslots) {
          p->d_blockflag.reserve((size_t) nb + 1);
          UCG_HIP(launch_block_classify(ctx->atoms_dev(), L, bslots, p->d_blockflag.get(), ctx->stream));
          p->blockflag_build = ctx->list_gen;
          p->blockflag_slots = bslots;
        }
        L.blockflag = p->d_blockflag.get();
        L.blockwant = part == 1 ? 0 : 1;
      }
      if (post) L.post = *post;
      // (a mirror upload in mirror_need above may just have raised it)
      p->dev.first_possible = ctx->ucgp_first_possible ? 1 : 0;
      if (p->vrow)
        UCG_HIP(launch_pair_vrow(p->dev, ctx->atoms_dev(), L, p->d_vr_ent.get(), (size_t) p->vr_pitch * (size_t) p->vr_cap,
                                 p->d_vr_lanemeta.get(), p->vr_pitch, ev, p->d_evpart.get(), p->d_evout.get(), p->d_err.get(),
                                 ctx->stream));
      else if (ctx->fma_contract)
        UCG_HIP(launch_pair_gather_fused(p->dev, ctx->atoms_dev(), L, ev, p->d_evpart.get(), p->d_evout.get(),
                                         p->d_err.get(), ctx->stream));
      else
        UCG_HIP(launch_pair_gather(p->dev, ctx->atoms_dev(), L, ev, p->d_evpart.get(), p->d_evout.get(), p->d_err.get(),
                                   ctx->stream));
    }
    mirror_wrote(ctx, UCG_F_F | UCG_F_UCGFORCE | UCG_F_SCORES | UCG_F_NSTATES | (p->model.style == STYLE_BETHE_DENSITY ? UCG_F_UCGP : 0));
    if (ctx->prof_on) {
      UCG_HIP(hipEventRecord(e1, ctx->stream));
      ctx->prof_ev.push_back(e0);
      ctx->prof_ev.push_back(e1);
    }
    if (ev) {
      double out[8];
      d2h(ctx, out, p->d_evout.get(), 8);
      sync(ctx);
      if (eng_vdwl) *eng_vdwl = out[0];
      if (virial)
        for (int c = 0; c < 6; c++) virial[c] = out[1 + c];
      ctx->thermo[0] = out[0];
      for (int c = 0; c < 6; c++) ctx->thermo[1 + c] = out[1 + c];
    }
    return UCG_OK;
  });
}

int ucg_pair_density_phase(ucg_pair *p, int phase, int eflag, int vflag, double *eng_vdwl, double *virial)
{
  if (!p || !p->ctx || phase < 1 || phase > 3) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded(ctx, [&]() -> int {
    if (!p->uploaded || p->model.style != STYLE_BETHE_DENSITY) return fail(ctx, UCG_ERR_INVALID, "not an initialised table_ucg_bethe_density style");
    if (ctx->list_inum != ctx->nlocal) return fail(ctx, UCG_ERR_INVALID, "no neighbour list for the current beads");
    const bool ev = (eflag || vflag);
    if (phase == 1) {
      p->dev.tcache = nullptr;
      if (ctx->density_tcache && ctx->list_maxrow > 0) {
        p->d_tcache.reserve((size_t) ctx->list_pitch * (size_t) ctx->list_maxrow);
        p->dev.tcache = p->d_tcache.get();
      }
      const size_t nall = (size_t) ctx->nlocal + ctx->nghost;
      p->d_prior.reserve(nall + 1);
      p->d_cv.reserve(nall + 1);
      p->d_partial.reserve((size_t) ctx->nlocal + 1);
      p->d_evpart.reserve((size_t) density_evpart_doubles(ctx->nlocal));
    }
    if (!p->host_tab.empty()) choose_hot_block(ctx, p, ctx->stage_own ? (size_t) 1024 * 36 : 0);  // (the density style: no ucgp staged)
    mirror_need(ctx, UCG_F_X | UCG_F_STATE | UCG_F_UCGL | UCG_F_UCGP);
    mirror_wrote(ctx, UCG_F_F | UCG_F_UCGFORCE | UCG_F_SCORES | UCG_F_NSTATES | UCG_F_UCGP);
    UCG_HIP(launch_density_phase(p->dev, ctx->atoms_dev(), ctx->list_dev(), phase, ev, p->d_prior.get(), p->d_partial.get(),
                                 p->d_cv.get(), p->d_evpart.get(), p->d_evout.get(), p->d_err.get(), ctx->stream));
    if (phase == 3 && ev) {
      double out[8];
      d2h(ctx, out, p->d_evout.get(), 8);
      sync(ctx);
      if (eng_vdwl) *eng_vdwl = out[0];
      if (virial)
        for (int c = 0; c < 6; c++) virial[c] = out[1 + c];
      ctx->thermo[0] = out[0];
      for (int c = 0; c < 6; c++) ctx->thermo[1 + c] = out[1 + c];
    }
    return UCG_OK;
  });
}

void *ucg_pair_density_buffer(ucg_pair *p, int which)
{
  if (!p) return nullptr;
  return which == 0 ? (void *) p->d_prior.get() : which == 1 ? (void *) p->d_cv.get() : nullptr;
}

int ucg_pair_density_aux_download(ucg_pair *p, int which, double *host, int first, int count)
{
  if (!p || !p->ctx || !host || first < 0 || count < 0 || (which != 0 && which != 1)) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded(ctx, [&]() -> int {
    DevBuf<double2> &B = which == 0 ? p->d_prior : p->d_cv;
    if ((size_t) first + (size_t) count > B.capacity()) return fail(ctx, UCG_ERR_INVALID, "density buffer range");
    if (count) d2h(ctx, (double2 *) host, B.get() + first, (size_t) count);
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_pair_density_aux_upload(ucg_pair *p, int which, const double *host, int first, int count)
{
  if (!p || !p->ctx || !host || first < 0 || count < 0 || (which != 0 && which != 1)) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded(ctx, [&]() -> int {
    DevBuf<double2> &B = which == 0 ? p->d_prior : p->d_cv;
    if ((size_t) first + (size_t) count > B.capacity()) return fail(ctx, UCG_ERR_INVALID, "density buffer range");
    if (count) h2d(ctx, B.get() + first, (const double2 *) host, (size_t) count);
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_pair_check_errors(ucg_pair *p)
{
  if (!p || !p->ctx) return UCG_ERR_INVALID;
  ucg_ctx *ctx = p->ctx;
  return guarded(ctx, [&]() -> int {
    if (!p->uploaded) return UCG_OK;
    int flag = 0;
    d2h(ctx, &flag, p->d_err.get(), 1);
    sync(ctx);
    if (!flag) return UCG_OK;
    UCG_HIP(hipMemsetAsync(p->d_err.get(), 0, sizeof(int), ctx->stream));
    if (flag & 1) return fail(ctx, UCG_ERR_TABLE_INNER, "Pair distance < table inner cutoff");
    if (flag & 2) return fail(ctx, UCG_ERR_TABLE_OUTER, "Pair distance > table outer cutoff");
    if (flag & 16)
      return fail(ctx, UCG_ERR_INVALID, "internal error: the Bethe gather kernel met a first-call marker (ucgp < -0.999) although the "
                                        "library held every bead's ucgp to be set");
    if (flag & 8) p->vr_gen = -1;  // the unusable virtual rows are made (and found wanting) again by the next compute: reported every time
    if (flag & 8)
      return fail(ctx, UCG_ERR_UNSUPPORTED, "the virtual rows of a block of 512 beads do not fit (a row of more than 128 entries, or "
                                            "more than 32 768 kept entries in one of the block's lists): set option pair_vrow 0 "
                                            "before ucg_pair_init");
    if (p->vrow)
      return fail(ctx, UCG_ERR_UNSUPPORTED, "a pair term left the range of the fixed sums (|term| >= 2^24 units of the tables' "
                                            "reference force / energy): the tables or lambda are far outside any physical range");
    return fail(ctx, UCG_ERR_HIP, "a pair kernel reported an unknown error flag");
  });
}

/* ----------------------------------------------------------------- atoms */

int ucg_atoms_upload(ucg_ctx *ctx, int nlocal, int nghost, int ntypes, const double *x, const double *v,
                     const int *type, const int *tag, const int *mask, const int *ucgstate, const double *ucgl,
                     const double *ucgvl, const double *ucgml, const double *ucgp, const double *mass)
{
  if (!ctx || nlocal < 0 || nghost < 0 || !x || !type || !ucgstate || !ucgl) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    const size_t nall = (size_t) nlocal + nghost, nl = (size_t) nlocal;
    if (nall >= (size_t) UCG_NEIGHMASK) return fail(ctx, UCG_ERR_INVALID, "too many atoms for 29-bit neighbour indices");
    ctx->pos4.reserve(nall);
    ctx->meta.reserve(nall);
    ctx->ucgp.reserve(nall);
    ctx->tag.reserve(nall);
    ctx->vel4.reserve(nl);
    ctx->frc4.reserve(nl);
    ctx->scores.reserve(nl);
    ctx->mask.reserve(nl);
    ctx->num_ucgstates.reserve(nl);
    ctx->ucgml.reserve(nl);
    ctx->mass.reserve((size_t) ntypes + 1);
    std::vector<double4> p4(nall), v4(nl);
    std::vector<int> meta(nall), tg(nall), mk(nl);
    std::vector<double> up(nall), ml(nl);
    for (size_t i = 0; i < nall; i++) {
      p4[i] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], ucgl[i]);
      if (type[i] < 1 || type[i] > 0xFFFF) return fail(ctx, UCG_ERR_INVALID, "atom type out of range");
      meta[i] = (type[i] & 0xFFFF) | ((ucgstate[i] & 1) << 16);
      tg[i] = tag ? tag[i] : (int) i + 1;
      up[i] = ucgp ? ucgp[i] : -1.0;
    }
    {
      bool first = false;
      for (size_t i = 0; i < nall && !first; i++) first = !(up[i] >= 0.0);
      ctx->ucgp_first_possible = first;
    }
    for (size_t i = 0; i < nl; i++) {
      v4[i] = make_double4(v ? v[3 * i] : 0.0, v ? v[3 * i + 1] : 0.0, v ? v[3 * i + 2] : 0.0, ucgvl ? ucgvl[i] : 0.0);
      mk[i] = mask ? mask[i] : 1;
      ml[i] = ucgml ? ucgml[i] : 1.0;
    }
    h2d(ctx, ctx->pos4.get(), p4.data(), nall);
    h2d(ctx, ctx->meta.get(), meta.data(), nall);
    h2d(ctx, ctx->tag.get(), tg.data(), nall);
    h2d(ctx, ctx->ucgp.get(), up.data(), nall);
    h2d(ctx, ctx->vel4.get(), v4.data(), nl);
    h2d(ctx, ctx->mask.get(), mk.data(), nl);
    h2d(ctx, ctx->ucgml.get(), ml.data(), nl);
    std::vector<double> ms((size_t) ntypes + 1, 1.0);
    if (mass)
      for (int t = 0; t <= ntypes; t++) ms[(size_t) t] = mass[t];
    h2d(ctx, ctx->mass.get(), ms.data(), ms.size());
    if (nl) {
      UCG_HIP(hipMemsetAsync(ctx->frc4.get(), 0, nl * sizeof(double4), ctx->stream));
      UCG_HIP(hipMemsetAsync(ctx->scores.get(), 0, nl * sizeof(double2), ctx->stream));
      UCG_HIP(hipMemsetAsync(ctx->num_ucgstates.get(), 0, nl * sizeof(int), ctx->stream));
    }
    sync(ctx);
    ctx->nlocal = nlocal;
    ctx->nghost = nghost;
    ctx->ntypes = ntypes;
    ctx->list_inum = 0;
    ctx->ghost_src_valid = (nghost == 0);
    ctx->mirror.dev_newer = ctx->mirror.host_newer = 0;  // the caller's arrays and the device hold the same values
    return UCG_OK;
  });
}

int ucg_atoms_upload_comm(ucg_ctx *ctx, const double *x, const int *ucgstate, const double *ucgl, const double *ucgp)
{
  if (!ctx || !x || !ucgstate || !ucgl) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    const size_t nall = (size_t) ctx->nlocal + ctx->nghost;
    std::vector<double4> p4(nall);
    std::vector<int> meta(nall);
    d2h(ctx, meta.data(), ctx->meta.get(), nall);
    sync(ctx);
    for (size_t i = 0; i < nall; i++) {
      p4[i] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], ucgl[i]);
      meta[i] = (meta[i] & 0xFFFF) | ((ucgstate[i] & 1) << 16);
    }
    h2d(ctx, ctx->pos4.get(), p4.data(), nall);
    h2d(ctx, ctx->meta.get(), meta.data(), nall);
    if (ucgp) {
      h2d(ctx, ctx->ucgp.get(), ucgp, nall);
      bool first = false;
      for (size_t i = 0; i < nall && !first; i++) first = !(ucgp[i] >= 0.0);
      ctx->ucgp_first_possible = first;
    }
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_atoms_upload_owned(ucg_ctx *ctx, const double *x, const double *v, const double *f, const int *ucgstate,
                           const int *num_ucgstates, const double *ucgl, const double *ucgvl, const double *ucgp,
                           const double *ucgforce, const double *ucgsoftmaxscores)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    // drop-in fix hooks: LAMMPS owns the host arrays, so what a hook reads is refreshed from them
    // first.  x/ucgl, v/ucgvl and f/ucgforce share a double4 record: the missing half is kept.
    const size_t nl = (size_t) ctx->nlocal;
    if (nl == 0) return UCG_OK;
    auto merge4 = [&](DevBuf<double4> &buf, const double *a3, const double *w) {
      if (!a3 && !w) return;
      std::vector<double4> h(nl);
      if (!a3 || !w) {
        d2h(ctx, h.data(), buf.get(), nl);
        sync(ctx);
      }
      for (size_t i = 0; i < nl; i++) {
        if (a3) {
          h[i].x = a3[3 * i];
          h[i].y = a3[3 * i + 1];
          h[i].z = a3[3 * i + 2];
        }
        if (w) h[i].w = w[i];
      }
      h2d(ctx, buf.get(), h.data(), nl);
      sync(ctx);
    };
    merge4(ctx->pos4, x, ucgl);
    merge4(ctx->vel4, v, ucgvl);
    merge4(ctx->frc4, f, ucgforce);
    if (ucgstate) {
      std::vector<int> meta(nl);
      d2h(ctx, meta.data(), ctx->meta.get(), nl);
      sync(ctx);
      for (size_t i = 0; i < nl; i++) meta[i] = (meta[i] & 0xFFFF) | ((ucgstate[i] & 1) << 16);
      h2d(ctx, ctx->meta.get(), meta.data(), nl);
    }
    if (num_ucgstates) h2d(ctx, ctx->num_ucgstates.get(), num_ucgstates, nl);
    if (ucgp) {
      h2d(ctx, ctx->ucgp.get(), ucgp, nl);
      for (size_t i = 0; i < nl && !ctx->ucgp_first_possible; i++) ctx->ucgp_first_possible = !(ucgp[i] >= 0.0);
    }
    if (ucgsoftmaxscores) h2d(ctx, (double *) ctx->scores.get(), ucgsoftmaxscores, 2 * nl);
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_atoms_download(ucg_ctx *ctx, int with_ghosts, double *x, double *v, double *f, int *type, int *tag,
                       int *ucgstate, int *num_ucgstates, double *ucgl, double *ucgvl, double *ucgml, double *ucgp,
                       double *ucgforce, double *ucgsoftmaxscores)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    const size_t nl = (size_t) ctx->nlocal, nall = nl + (with_ghosts ? (size_t) ctx->nghost : 0);
    std::vector<double4> p4, v4, f4;
    std::vector<int> meta;
    if (x || ucgl) {
      p4.resize(nall);
      d2h(ctx, p4.data(), ctx->pos4.get(), nall);
    }
    if (v || ucgvl) {
      v4.resize(nl);
      d2h(ctx, v4.data(), ctx->vel4.get(), nl);
    }
    if (f || ucgforce) {
      f4.resize(nl);
      d2h(ctx, f4.data(), ctx->frc4.get(), nl);
    }
    if (type || ucgstate) {
      meta.resize(nall);
      d2h(ctx, meta.data(), ctx->meta.get(), nall);
    }
    if (tag) d2h(ctx, tag, ctx->tag.get(), nall);
    if (num_ucgstates) d2h(ctx, num_ucgstates, ctx->num_ucgstates.get(), nl);
    if (ucgml) d2h(ctx, ucgml, ctx->ucgml.get(), nl);
    if (ucgp) d2h(ctx, ucgp, ctx->ucgp.get(), nall);
    if (ucgsoftmaxscores) d2h(ctx, (double2 *) ucgsoftmaxscores, ctx->scores.get(), nl);
    sync(ctx);
    for (size_t i = 0; i < nall; i++) {
      if (x) { x[3 * i] = p4[i].x; x[3 * i + 1] = p4[i].y; x[3 * i + 2] = p4[i].z; }
      if (ucgl) ucgl[i] = p4[i].w;
      if (type) type[i] = meta[i] & 0xFFFF;
      if (ucgstate) ucgstate[i] = (meta[i] >> 16) & 1;
    }
    for (size_t i = 0; i < nl; i++) {
      if (v) { v[3 * i] = v4[i].x; v[3 * i + 1] = v4[i].y; v[3 * i + 2] = v4[i].z; }
      if (ucgvl) ucgvl[i] = v4[i].w;
      if (f) { f[3 * i] = f4[i].x; f[3 * i + 1] = f4[i].y; f[3 * i + 2] = f4[i].z; }
      if (ucgforce) ucgforce[i] = f4[i].w;
    }
    return UCG_OK;
  });
}

int ucg_ghosts_upload(ucg_ctx *ctx, const int *src, int nghost)
{
  if (!ctx || nghost < 0 || (nghost > 0 && !src)) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (nghost != ctx->nghost) return fail(ctx, UCG_ERR_INVALID, "ghost map length differs from the resident ghost count");
    for (int g = 0; g < nghost; g++)
      if (src[g] < 0 || src[g] >= ctx->nlocal) return fail(ctx, UCG_ERR_INVALID, "ghost source index out of range");
    ctx->ghost_src.reserve((size_t) nghost + 1);
    h2d(ctx, ctx->ghost_src.get(), src, (size_t) nghost);
    sync(ctx);
    ctx->ghost_src_valid = true;
    return UCG_OK;
  });
}

int ucg_atoms_counts(const ucg_ctx *ctx, int *nlocal, int *nghost)
{
  if (!ctx) return UCG_ERR_INVALID;
  if (nlocal) *nlocal = ctx->nlocal;
  if (nghost) *nghost = ctx->nghost;
  return UCG_OK;
}

int ucg_atoms_download_mask(ucg_ctx *ctx, int *mask)
{
  if (!ctx || !mask) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (ctx->nlocal) d2h(ctx, mask, ctx->mask.get(), (size_t) ctx->nlocal);
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_force_clear(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    UCG_HIP(launch_force_clear(ctx->atoms_dev(), ctx->stream));
    return UCG_OK;
  });
}

/* ------------------------------------------------------------- neighbours */

int ucg_neigh_upload_full(ucg_ctx *ctx, int inum, const int *numneigh, const long long *first, const int *neigh)
{
  if (!ctx || inum < 0 || !numneigh || !first || (!neigh && inum)) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (inum != ctx->nlocal) return fail(ctx, UCG_ERR_INVALID, "full list must have one row per owned atom");
    int maxrow = 0;
    long long total = 0;
    const int nall = ctx->nlocal + ctx->nghost;
    for (int i = 0; i < inum; i++) {
      if (numneigh[i] > maxrow) maxrow = numneigh[i];
      total += numneigh[i];
    }
    const int pitch = ((inum + 63) / 64) * 64;
    std::vector<int> tr((size_t) pitch * (size_t) (maxrow > 0 ? maxrow : 1), 0);
    for (int i = 0; i < inum; i++)
      for (int e = 0; e < numneigh[i]; e++) {
        const int ent = neigh[first[i] + e];
        if ((ent & UCG_NEIGHMASK) >= nall) return fail(ctx, UCG_ERR_INVALID, "neighbour index beyond nlocal+nghost");
        tr[(size_t) e * pitch + i] = ent;
      }
    ctx->neigh.reserve(tr.size());
    ctx->numneigh.reserve((size_t) pitch);
    h2d(ctx, ctx->neigh.get(), tr.data(), tr.size());
    h2d(ctx, ctx->numneigh.get(), numneigh, (size_t) inum);
    sync(ctx);
    ctx->list_inum = inum;
    ctx->list_pitch = pitch;
    ctx->list_maxrow = maxrow;
    ctx->list_entries = total;
    ctx->list_from_builder = false;
    ctx->list_gen++;
    return UCG_OK;
  });
}

int ucg_neigh_download(ucg_ctx *ctx, int *inum, int *numneigh, long long *first, int *neigh, long long cap, long long *total)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (inum) *inum = ctx->list_inum;
    if (total) *total = ctx->list_entries;
    if (!numneigh && !neigh) return UCG_OK;
    const int n = ctx->list_inum;
    std::vector<int> nn((size_t) n);
    d2h(ctx, nn.data(), ctx->numneigh.get(), (size_t) n);
    sync(ctx);
    if (numneigh) std::memcpy(numneigh, nn.data(), sizeof(int) * (size_t) n);
    if (first || neigh) {
      std::vector<long long> fst((size_t) n);
      long long pos = 0;
      for (int i = 0; i < n; i++) {
        fst[(size_t) i] = pos;
        pos += nn[(size_t) i];
      }
      if (first) std::memcpy(first, fst.data(), sizeof(long long) * (size_t) n);
      if (neigh) {
        if (cap < pos) return fail(ctx, UCG_ERR_INVALID, "neighbour buffer too small");
        std::vector<int> tr((size_t) ctx->list_pitch * (size_t) (ctx->list_maxrow > 0 ? ctx->list_maxrow : 1));
        d2h(ctx, tr.data(), ctx->neigh.get(), tr.size());
        sync(ctx);
        for (int i = 0; i < n; i++)
          for (int e = 0; e < nn[(size_t) i]; e++) neigh[fst[(size_t) i] + e] = tr[(size_t) e * ctx->list_pitch + i];
      }
    }
    return UCG_OK;
  });
}

/* ------------------------------------------------------------------ fixes */

int ucg_fix_nve_initial(ucg_ctx *ctx, int groupbit)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    // dtv = dt, dtf = 0.5*dt*ftm2v  (UCG/fix_nve_ucgld.cpp:36-38)
    mirror_need(ctx, UCG_F_X | UCG_F_V | UCG_F_F | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGFORCE);
    UCG_HIP(launch_nve_initial(ctx->atoms_dev(), ctx->dt, 0.5 * ctx->dt * ctx->ftm2v, groupbit, 0, ctx->stream));
    mirror_wrote(ctx, UCG_F_X | UCG_F_V | UCG_F_UCGL | UCG_F_UCGVL);
    return UCG_OK;
  });
}

int ucg_fix_nve_final(ucg_ctx *ctx, int groupbit)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    mirror_need(ctx, UCG_F_V | UCG_F_F | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGFORCE);
    UCG_HIP(launch_nve_final(ctx->atoms_dev(), 0.5 * ctx->dt * ctx->ftm2v, groupbit, 0, ctx->stream));
    mirror_wrote(ctx, UCG_F_V | UCG_F_UCGVL);
    return UCG_OK;
  });
}

/* fix nve/ucgld/wall/hard (UCG/fix_nve_ucgld_wall_hard.cpp) */
int ucg_fix_nve_wall_hard_set(ucg_ctx *ctx, int bias_potential, double barrier)
{
  if (!ctx) return UCG_ERR_INVALID;
  // the constructor's keyword loop (:21-33): "bias_potential [barrier]", default barrier 0.1
  ctx->wall_bias = bias_potential != 0;
  ctx->wall_barrier = barrier;
  return UCG_OK;
}

int ucg_fix_nve_wall_hard_initial(ucg_ctx *ctx, int groupbit)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    mirror_need(ctx, UCG_F_X | UCG_F_V | UCG_F_F | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGFORCE | UCG_F_STATE);
    UCG_HIP(launch_nve_initial(ctx->atoms_dev(), ctx->dt, 0.5 * ctx->dt * ctx->ftm2v, groupbit, 2, ctx->stream));
    mirror_wrote(ctx, UCG_F_X | UCG_F_V | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_STATE);
    return UCG_OK;
  });
}

int ucg_fix_nve_wall_hard_final(ucg_ctx *ctx, int groupbit)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    mirror_need(ctx, UCG_F_V | UCG_F_F | UCG_F_UCGL | UCG_F_UCGVL | UCG_F_UCGFORCE);
    UCG_HIP(launch_nve_final(ctx->atoms_dev(), 0.5 * ctx->dt * ctx->ftm2v, groupbit, 2, ctx->stream));
    mirror_wrote(ctx, UCG_F_V | UCG_F_UCGVL | UCG_F_UCGL);
    return UCG_OK;
  });
}

int ucg_fix_nve_wall_hard_post_force(ucg_ctx *ctx, int groupbit)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    // setmask() adds POST_FORCE only with bias_potential (:41-52)
    if (ctx->wall_bias) {
      mirror_need(ctx, UCG_F_UCGL | UCG_F_UCGFORCE | UCG_F_F);
      UCG_HIP(launch_wall_bias(ctx->atoms_dev(), ctx->wall_barrier, groupbit, ctx->stream));
      mirror_wrote(ctx, UCG_F_UCGFORCE);
    }
    return UCG_OK;
  });
}

static void rng_setup(ucg_ctx *ctx, RanMarsDev &R, DevBuf<unsigned int> &h0, DevBuf<unsigned int> &h1, int seed)
{
  unsigned int hist[97];
  long long count = 0;
  ranmars_seed_host(seed, hist, &count);
  h0.reserve(128);
  h1.reserve(128);
  UCG_HIP(hipMemcpyAsync(h0.get(), hist, sizeof(hist), hipMemcpyHostToDevice, ctx->stream));
  UCG_HIP(hipStreamSynchronize(ctx->stream));
  R.hist[0] = h0.get();
  R.hist[1] = h1.get();
  R.cur = 0;
  R.count = count;
}

static void rng_draw(ucg_ctx *ctx, RanMarsDev &R, DevBuf<unsigned int> &draws, int n)
{
  ctx->ensure_rm_jump(n);
  R.jump = ctx->rm_jump.get();
  R.nchunks_max = ctx->rm_chunks;
  draws.reserve((size_t) n + 1);
  UCG_HIP(launch_ranmars(R, n, draws.get(), ctx->stream));
}

// The next n draws of a per-bead stream (one per owned bead, in local order).  The generator is sequential, so the
// draws of the following steps are known in advance: one launch produces rng_batch steps' worth and the steps take
// their slices.  When the bead count changes before the window is used up (migration in a decomposed run), the stream
// is taken back to the window's start and advanced by what was actually consumed -- the values every step sees are
// those of one launch per step.
static const unsigned int *rng_next(ucg_ctx *ctx, RanMarsDev &R, DevBuf<unsigned int> &draws, RngBatch &B, int n)
{
  if (n <= 0) return draws.get();
  if (B.unit == n && B.total - B.used >= n) {
    const unsigned int *p = draws.get() + B.used;
    B.used += n;
    return p;
  }
  if (B.used < B.total) {
    UCG_HIP(hipMemcpyAsync(R.hist[R.cur], B.hist_save.get(), 97 * sizeof(unsigned int), hipMemcpyDeviceToDevice, ctx->stream));
    R.count = B.count0;
    long long left = B.used;
    while (left > 0) {  // replay the consumed part (into the window itself: it is regenerated below)
      const int piece = (int) std::min<long long>(left, B.total);
      rng_draw(ctx, R, draws, piece);
      left -= piece;
    }
  }
  const long long k = std::max(1, ctx->rng_batch);
  const long long want = std::min<long long>(k * n, 1ll << 30);
  B.hist_save.reserve(128);
  UCG_HIP(hipMemcpyAsync(B.hist_save.get(), R.hist[R.cur], 97 * sizeof(unsigned int), hipMemcpyDeviceToDevice, ctx->stream));
  B.count0 = R.count;
  rng_draw(ctx, R, draws, (int) want);
  B.total = want;
  B.used = n;
  B.unit = n;
  return draws.get();
}

int ucg_fix_langevin_create(ucg_ctx *ctx, double t_start, double t_stop, double t_period, int seed, int me)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    if (t_period <= 0.0) throw InputError{"Fix langevin period must be > 0.0"};
    if (seed <= 0) throw InputError{"Illegal fix langevin command"};
    if (seed + me > 900000000) throw InputError{"Invalid seed for Marsaglia random # generator"};
    FixLangevin &L = ctx->lang;
    L.active = true;
    L.t_start = t_start;
    L.t_target = t_start;
    L.t_stop = t_stop;
    L.t_period = t_period;
    L.seed = seed;
    L.inited = false;
    rng_setup(ctx, L.rng, L.hist0, L.hist1, seed + me);
    L.batch.total = L.batch.used = 0;
    L.batch.unit = 0;
    return UCG_OK;
  });
}

int ucg_fix_langevin_init(ucg_ctx *ctx, int ntypes, const double *gfactor1, const double *gfactor2)
{
  if (!ctx || !gfactor1 || !gfactor2 || ntypes < 1) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    FixLangevin &L = ctx->lang;
    if (!L.active) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not created");
    L.ntypes = ntypes;
    L.gf1.reserve((size_t) ntypes + 1);
    L.gf2.reserve((size_t) ntypes + 1);
    h2d(ctx, L.gf1.get(), gfactor1, (size_t) ntypes + 1);
    h2d(ctx, L.gf2.get(), gfactor2, (size_t) ntypes + 1);
    sync(ctx);
    L.inited = true;
    return UCG_OK;
  });
}

int ucg_fix_langevin_init_from_ucgml(ucg_ctx *ctx, int ntypes, const double *ml)
{
  if (!ctx || !ml || ntypes < 1) return UCG_ERR_INVALID;
  // Fix_UCGLD_Langevin::init(), UCG/fix_ucgld_langevin.cpp:164-171 (ratio[i] = 1)
  std::vector<double> g1((size_t) ntypes + 1, 0.0), g2((size_t) ntypes + 1, 0.0);
  const FixLangevin &L = ctx->lang;
  for (int i = 1; i <= ntypes; i++) {
    g1[(size_t) i] = -ml[i] / L.t_period / ctx->ftm2v;
    g2[(size_t) i] = std::sqrt(ml[i]) / ctx->ftm2v;
    g2[(size_t) i] *= std::sqrt(24.0 * ctx->boltz / L.t_period / ctx->dt / ctx->mvv2e);
    g1[(size_t) i] *= 1.0 / 1.0;
    g2[(size_t) i] *= 1.0 / std::sqrt(1.0);
  }
  return ucg_fix_langevin_init(ctx, ntypes, g1.data(), g2.data());
}

int ucg_fix_langevin_post_force(ucg_ctx *ctx, int groupbit, long long ntimestep, long long beginstep, long long endstep)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    FixLangevin &L = ctx->lang;
    if (!L.active || !L.inited) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not initialised");
    // compute_target(), :318-330
    double delta = (double) (ntimestep - beginstep);
    if (delta != 0.0) delta /= (double) (endstep - beginstep);
    L.t_target = L.t_start + delta * (L.t_stop - L.t_start);
    L.tsqrt = std::sqrt(L.t_target);
    LangevinDev Lg;
    Lg.gfactor1 = L.gf1.get();
    Lg.gfactor2 = L.gf2.get();
    Lg.tsqrt = L.tsqrt;
    Lg.bias = L.bias ? 1 : 0;
    Lg.draws = rng_next(ctx, L.rng, L.draws, L.batch, ctx->nlocal);
    mirror_need(ctx, UCG_F_UCGVL | UCG_F_UCGFORCE | UCG_F_F);
    UCG_HIP(launch_langevin(ctx->atoms_dev(), Lg, groupbit, ctx->stream));
    mirror_wrote(ctx, UCG_F_UCGFORCE);
    return UCG_OK;
  });
}

int ucg_fix_langevin_end_of_step(ucg_ctx *ctx, int groupbit, double *lambda_temp)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    const int nb = (ctx->nlocal + 255) / 256 + 1;
    ctx->redpart.reserve((size_t) nb * 4);
    mirror_need(ctx, UCG_F_UCGVL | UCG_F_STATE);  // sum of ml vl^2 (:303-312) + the state-1 population of the thermo line
    UCG_HIP(launch_lambda_ke(ctx->atoms_dev(), groupbit, ctx->mvv2e, ctx->redpart.get(), ctx->redout.get(), ctx->stream));
    double out[4];
    d2h(ctx, out, ctx->redout.get(), 4);
    sync(ctx);
    // lambda_temp = lmd_ek / (0.5 * boltz * nlocal)   (:311)
    ctx->lang.lambda_temp = out[0] / (0.5 * ctx->boltz * ctx->nlocal);
    ctx->thermo[7] = ctx->lang.lambda_temp;
    ctx->thermo[8] = out[1];
    if (lambda_temp) *lambda_temp = ctx->lang.lambda_temp;
    return UCG_OK;
  });
}

double ucg_fix_langevin_t_target(const ucg_ctx *ctx) { return ctx ? ctx->lang.t_target : 0.0; }

int ucg_fix_langevin_reset_target(ucg_ctx *ctx, double t_new)
{
  if (!ctx) return UCG_ERR_INVALID;
  FixLangevin &L = ctx->lang;
  if (!L.active) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not created");
  L.t_target = L.t_start = L.t_stop = t_new;  // UCG/fix_ucgld_langevin.cpp:358-361
  return UCG_OK;
}

int ucg_fix_langevin_reset_dt(ucg_ctx *ctx, int ntypes, const double *mass_by_type)
{
  if (!ctx || ntypes < 1) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    FixLangevin &L = ctx->lang;
    if (!L.active || !L.inited) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not initialised");
    if (ntypes != L.ntypes) return fail(ctx, UCG_ERR_INVALID, "ucg_fix_langevin_reset_dt: ntypes differs from the one given to init");
    std::vector<double> mass((size_t) ntypes + 1, 0.0), g2((size_t) ntypes + 1, 0.0);
    if (mass_by_type) {
      for (int i = 0; i <= ntypes; i++) mass[(size_t) i] = mass_by_type[i];
    } else {
      if (ctx->ntypes != ntypes) return fail(ctx, UCG_ERR_INVALID, "ucg_fix_langevin_reset_dt: no masses on the device for this ntypes");
      d2h(ctx, mass.data(), ctx->mass.get(), (size_t) ntypes + 1);
      sync(ctx);
    }
    // reset_dt() as shipped (UCG/fix_ucgld_langevin.cpp:366-376): `if (atom->mass)` -- always true for atom style ucg --
    // gfactor2 from atom->mass[i] (init() used atom->ucgml[i]), ratio[i] = 1; gfactor1 is left alone
    for (int i = 1; i <= ntypes; i++) {
      g2[(size_t) i] = std::sqrt(mass[(size_t) i]) / ctx->ftm2v;
      g2[(size_t) i] *= std::sqrt(24.0 * ctx->boltz / L.t_period / ctx->dt / ctx->mvv2e);
      g2[(size_t) i] *= 1.0 / std::sqrt(1.0);
    }
    h2d(ctx, L.gf2.get(), g2.data(), (size_t) ntypes + 1);
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_fix_langevin_set_bias(ucg_ctx *ctx, int bias)
{
  if (!ctx) return UCG_ERR_INVALID;
  ctx->lang.bias = bias != 0;
  return UCG_OK;
}

int ucg_fix_ucgstate_create(ucg_ctx *ctx, int ld_flag, int mc_flag, int mc_seed, double mc_rate, int me)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    FixUcgState &S = ctx->ucgst;
    S.active = true;
    S.ld_flag = ld_flag;
    S.mc_flag = mc_flag;
    S.mc_seed = mc_seed;
    S.mc_rate = mc_rate;
    if (mc_flag) {
      if (mc_seed + me <= 0 || mc_seed + me > 900000000) throw InputError{"Invalid seed for Marsaglia random # generator"};
      rng_setup(ctx, S.rng, S.hist0, S.hist1, mc_seed + me);
      S.batch.total = S.batch.used = 0;
      S.batch.unit = 0;
    }
    return UCG_OK;
  });
}

int ucg_fix_ucgstate_post_force(ucg_ctx *ctx)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    FixUcgState &S = ctx->ucgst;
    if (!S.active) return fail(ctx, UCG_ERR_INVALID, "fix ucgstate not created");
    const unsigned int *draws = nullptr;
    if (S.mc_flag && !S.ld_flag) {
      // one uniform() per 2-state owned bead in index order (:117); every bead is 2-state here
      draws = rng_next(ctx, S.rng, S.draws, S.batch, ctx->nlocal);
    }
    mirror_need(ctx, UCG_F_SCORES | UCG_F_NSTATES | UCG_F_STATE | UCG_F_X | UCG_F_UCGL);
    UCG_HIP(launch_ucgstate(ctx->atoms_dev(), S.ld_flag, S.mc_flag, S.mc_rate, draws, ctx->stream));
    ctx->ucgp_first_possible = false;  // every owned bead's ucgp now lies in [1e-6, 1 - 1e-6] (or is 1 for a one-state bead)
    mirror_wrote(ctx, UCG_F_UCGP | UCG_F_STATE | UCG_F_UCGL);
    return UCG_OK;
  });
}

int ucg_md_post_fused(ucg_ctx *ctx, int use_langevin, int use_ucgstate, int use_nve, int fuse_next_initial, int groupbit,
                      long long ntimestep, long long beginstep, long long endstep)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    LangevinDev Lg{};
    if (use_langevin) {
      FixLangevin &L = ctx->lang;
      if (!L.active || !L.inited) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not initialised");
      double delta = (double) (ntimestep - beginstep);
      if (delta != 0.0) delta /= (double) (endstep - beginstep);
      L.t_target = L.t_start + delta * (L.t_stop - L.t_start);
      L.tsqrt = std::sqrt(L.t_target);
      Lg.gfactor1 = L.gf1.get();
      Lg.gfactor2 = L.gf2.get();
      Lg.tsqrt = L.tsqrt;
      Lg.bias = L.bias ? 1 : 0;
      Lg.draws = rng_next(ctx, L.rng, L.draws, L.batch, ctx->nlocal);
    }
    const unsigned int *mc_draws = nullptr;
    FixUcgState &S = ctx->ucgst;
    if (use_ucgstate) {
      if (!S.active) return fail(ctx, UCG_ERR_INVALID, "fix ucgstate not created");
      if (S.mc_flag && !S.ld_flag) {
        mc_draws = rng_next(ctx, S.rng, S.draws, S.batch, ctx->nlocal);
      }
    }
    // reads and writes every per-bead field: what the caller announced with ucg_host_modified goes up first (mirror_wrote
    // below would otherwise discard the announcement)
    mirror_need(ctx, UCG_F_ALL);
    UCG_HIP(launch_post_fused(ctx->atoms_dev(), use_langevin != 0, Lg, use_ucgstate != 0, S.ld_flag, S.mc_flag, S.mc_rate,
                              mc_draws, use_nve != 0, fuse_next_initial != 0, ctx->dt, 0.5 * ctx->dt * ctx->ftm2v,
                              groupbit, use_nve >= 2 ? (ctx->wall_bias ? 3 : 2) : 0, ctx->wall_barrier, ctx->stream));
    if (use_ucgstate) ctx->ucgp_first_possible = false;
    mirror_wrote(ctx, UCG_F_ALL);
    return UCG_OK;
  });
}

/* pair force + the per-bead hooks of a step whose next initial_integrate is fused in, as ONE launch of the
 * gather kernel (its epilogue, PostDev): same statements, same order, same bits as ucg_pair_compute followed by
 * ucg_md_post_fused(..., fuse_next_initial = 1).  Returns UCG_ERR_UNSUPPORTED (nothing done) where the two-launch
 * form has to be used: table_ucg_bethe_density, or the option "post_in_pair" switched off. */
int ucg_md_pair_post(ucg_ctx *ctx, ucg_pair *p, int use_langevin, int use_ucgstate, int use_nve, int groupbit,
                     long long ntimestep, long long beginstep, long long endstep)
{
  if (!ctx || !p || p->ctx != ctx) return UCG_ERR_INVALID;
  if (!ctx->post_in_pair || p->model.style == STYLE_BETHE_DENSITY || !use_nve) return UCG_ERR_UNSUPPORTED;
  PostDev Q{};
  int rc = guarded(ctx, [&]() -> int {
    Q.enabled = 1;
    Q.lang = use_langevin != 0;
    Q.ucgst = use_ucgstate != 0;
    Q.nve = use_nve >= 2 ? (ctx->wall_bias ? 3 : 2) : 1;
    Q.groupbit = groupbit;
    Q.dtv = ctx->dt;
    Q.dtf = 0.5 * ctx->dt * ctx->ftm2v;
    Q.barrier = ctx->wall_barrier;
    if (use_langevin) {
      FixLangevin &L = ctx->lang;
      if (!L.active || !L.inited) return fail(ctx, UCG_ERR_INVALID, "fix ucgld/langevin not initialised");
      double delta = (double) (ntimestep - beginstep);
      if (delta != 0.0) delta /= (double) (endstep - beginstep);
      L.t_target = L.t_start + delta * (L.t_stop - L.t_start);
      L.tsqrt = std::sqrt(L.t_target);
      Q.gfactor1 = L.gf1.get();
      Q.gfactor2 = L.gf2.get();
      Q.tsqrt = L.tsqrt;
      Q.lang_bias = L.bias ? 1 : 0;
      Q.lang_draws = rng_next(ctx, L.rng, L.draws, L.batch, ctx->nlocal);  // the draws do not depend on the forces
    }
    FixUcgState &S = ctx->ucgst;
    if (use_ucgstate) {
      if (!S.active) return fail(ctx, UCG_ERR_INVALID, "fix ucgstate not created");
      Q.ld_flag = S.ld_flag;
      Q.mc_flag = S.mc_flag;
      Q.mc_rate = S.mc_rate;
      if (S.mc_flag && !S.ld_flag) {
        Q.mc_draws = rng_next(ctx, S.rng, S.draws, S.batch, ctx->nlocal);
      }
    }
    mirror_need(ctx, UCG_F_ALL);  // as ucg_md_post_fused: host-side edits go up before the launch that overwrites everything
    ctx->pos4_alt.reserve_exact(ctx->pos4.capacity());
    ctx->meta_alt.reserve_exact(ctx->meta.capacity());
    Q.pos_out = ctx->pos4_alt.get();
    Q.meta_out = ctx->meta_alt.get();
    if (Q.ucgst) {
      ctx->ucgp_alt.reserve_exact(ctx->ucgp.capacity());
      Q.ucgp_out = ctx->ucgp_alt.get();
    }
    return UCG_OK;
  });
  if (rc) return rc;
  rc = pair_compute_impl(p, 0, 0, nullptr, nullptr, 0, &Q);
  if (rc) return rc;
  // the next step's positions / states become the current ones (ghost entries are refreshed by the halo)
  ctx->pos4.swap(ctx->pos4_alt);
  ctx->meta.swap(ctx->meta_alt);
  if (Q.ucgst) {
    ctx->ucgp.swap(ctx->ucgp_alt);
    ctx->ucgp_first_possible = false;
  }
  mirror_wrote(ctx, UCG_F_ALL);
  return UCG_OK;
}

int ucg_ranmars_fill(ucg_ctx *ctx, int seed, long long skip, int n, double *out)
{
  if (!ctx || !out || n < 0 || skip < 0) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    RanMarsDev R{};
    DevBuf<unsigned int> h0, h1, draws;
    rng_setup(ctx, R, h0, h1, seed);
    // skip in pieces so that the state carry-over between launches is exercised too
    long long left = skip;
    while (left > 0) {
      const int piece = (int) (left > 1000003 ? 1000003 : left);
      rng_draw(ctx, R, draws, piece);
      left -= piece;
    }
    if (n) {
      rng_draw(ctx, R, draws, n);
      std::vector<unsigned int> host((size_t) n);
      d2h(ctx, host.data(), draws.get(), (size_t) n);
      sync(ctx);
      for (int i = 0; i < n; i++) out[i] = (double) host[(size_t) i] * 5.9604644775390625e-08;
    }
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_selftest_div(ucg_ctx *ctx, double b, long long seed, int n, long long *mismatches)
{
  if (!ctx || !mismatches || n < 0) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    DevBuf<unsigned long long> d;
    d.reserve(2);
    UCG_HIP(hipMemsetAsync(d.get(), 0, sizeof(unsigned long long), ctx->stream));
    UCG_HIP(launch_selftest_div(b, (unsigned long long) seed, n, d.get(), ctx->stream));
    unsigned long long h = 0;
    d2h(ctx, &h, d.get(), 1);
    sync(ctx);
    *mismatches = (long long) h;
    return UCG_OK;
  });
}

int ucg_selftest_div_core(ucg_ctx *ctx, long long seed, int n, long long *mismatches)
{
  if (!ctx || !mismatches || n < 0) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    DevBuf<unsigned long long> d;
    d.reserve(2);
    UCG_HIP(hipMemsetAsync(d.get(), 0, sizeof(unsigned long long), ctx->stream));
    UCG_HIP(launch_selftest_div_core((unsigned long long) seed, n, d.get(), ctx->stream));
    unsigned long long h = 0;
    d2h(ctx, &h, d.get(), 1);
    sync(ctx);
    *mismatches = (long long) h;
    return UCG_OK;
  });
}

int ucg_selftest_sqrt_core(ucg_ctx *ctx, long long seed, int n, long long *mismatches)
{
  if (!ctx || !mismatches || n < 0) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    DevBuf<unsigned long long> d;
    d.reserve(2);
    UCG_HIP(hipMemsetAsync(d.get(), 0, sizeof(unsigned long long), ctx->stream));
    UCG_HIP(launch_selftest_sqrt_core((unsigned long long) seed, n, d.get(), ctx->stream));
    unsigned long long h = 0;
    d2h(ctx, &h, d.get(), 1);
    sync(ctx);
    *mismatches = (long long) h;
    return UCG_OK;
  });
}

int ucg_selftest_stream(ucg_ctx *ctx, long long nbytes, int wide, int repeats)
{
  if (!ctx || nbytes < 16 || repeats < 1) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    DevBuf<char> buf;
    DevBuf<int> sink;
    buf.reserve((size_t) nbytes);
    sink.reserve(4);
    UCG_HIP(hipMemsetAsync(buf.get(), 1, (size_t) nbytes, ctx->stream));
    for (int r = 0; r < repeats; r++) UCG_HIP(launch_stream(buf.get(), (size_t) nbytes, wide, sink.get(), ctx->stream));
    sync(ctx);
    return UCG_OK;
  });
}

int ucg_ctx_set_option(ucg_ctx *ctx, const char *name, int value)
{
  if (!ctx || !name) return UCG_ERR_INVALID;
  if (std::strcmp(name, "generic_kernels") == 0) {
    ctx->force_generic_kernels = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "md_no_fuse") == 0) {
    ctx->md_no_fuse = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "fma_contract") == 0) {
    ctx->fma_contract = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "post_in_pair") == 0) {
    ctx->post_in_pair = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "rng_batch") == 0) {
    if (value < 1 || value > 64) {
      ctx->err = "rng_batch must be 1 .. 64";
      return UCG_ERR_INVALID;
    }
    ctx->rng_batch = value;
    return UCG_OK;
  }
  if (std::strcmp(name, "fault_inject_step") == 0) {
    ctx->fault_step = value;
    return UCG_OK;
  }
  if (std::strcmp(name, "fault_inject_setup") == 0) {
    ctx->fault_setup = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "rows_sort_r2") == 0) {
    ctx->rows_sort_r2 = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "rows_untiled") == 0) {
    ctx->rows_untiled = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "pair_vrow") == 0) {
    ctx->pair_vrow = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "hot_block") == 0) {
    ctx->hot_block = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "kind_blocks") == 0) {
    ctx->kind_blocks = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "density_tcache") == 0) {
    ctx->density_tcache = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "stream_rows") == 0) {
    ctx->stream_rows = value < 0 ? -1 : (value != 0 ? 1 : 0);
    return UCG_OK;
  }
  if (std::strcmp(name, "stage_own") == 0) {
    ctx->stage_own = value != 0;
    return UCG_OK;
  }
  if (std::strcmp(name, "gather_slots") == 0) {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) {
      ctx->err = "gather_slots must be 0 (auto), 1, 2, 4, 8 or 16";
      return UCG_ERR_INVALID;
    }
    ctx->gather_slots = value;
    return UCG_OK;
  }
  if (std::strcmp(name, "density_proximity_as_shipped") == 0) {
    ctx->density_proximity_as_shipped = value != 0;
    return UCG_OK;
  }
  ctx->err = std::string("unknown option ") + name;
  return UCG_ERR_INVALID;
}

/* ------------------------------------------------------------- measurement */

int ucg_profile_enable(ucg_ctx *ctx, int on)
{
  if (!ctx) return UCG_ERR_INVALID;
  ctx->prof_on = on != 0;
  return UCG_OK;
}

int ucg_profile_read(ucg_ctx *ctx, long long *pair_launches, double *pair_ms, int reset)
{
  if (!ctx) return UCG_ERR_INVALID;
  return guarded(ctx, [&]() -> int {
    sync(ctx);
    for (size_t i = 0; i + 1 < ctx->prof_ev.size(); i += 2) {
      float ms = 0.f;
      UCG_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
      ctx->prof_ms += ms;
      ctx->prof_launches++;
      (void) hipEventDestroy(ctx->prof_ev[i]);
      (void) hipEventDestroy(ctx->prof_ev[i + 1]);
    }
    ctx->prof_ev.clear();
    if (pair_launches) *pair_launches = ctx->prof_launches;
    if (pair_ms) *pair_ms = ctx->prof_ms;
    if (reset) {
      ctx->prof_launches = 0;
      ctx->prof_ms = 0;
    }
    return UCG_OK;
  });
}

}  // extern "C"
