// ucg_density.hip -- table_ucg_bethe_density on gfx950: the three passes of
// PairTable_UCG_Bethe_Density::compute (UCG/pair_table_ucg_bethe_density.cpp:133-758,
// helpers :107-127), Scenario 4 (:529-658), as gathers over the FULL neighbour list.
//
//   k_density_pass1   :219-274   rho_k = sum_m w(r) -> prior p_k0 = 1/2 + 1/2 tanh((rho-rho_th)/(0.1 rho_th))
//   (halo)                       ghosts take their owner's prior   (the reference's forward_comm moves
//                                0 bytes, App. B #7: ghost priors = 0 -> log(0/0) = NaN back-forces)
//   k_density_pass2   :284-696   tables, pseudo-likelihood scores, Bethe closure (as shipped, unguarded),
//                                pair forces (x 1/2 when the neighbour is owned: newton off, full list),
//                                entropic accumulators G, posterior -> ucgp, CV forces G * dp/drho
//   (halo)                       ghosts take their owner's CV forces
//   k_density_pass3   :698-733   back-force of the density CV over the neighbours
//
// Pass 1 leaves tanh of every in-cutoff entry's proximity argument in PairDev::tcache (one double per list entry); pass 3,
// which needs the same value for the same pair, reads it back.  Decks of several actual types whose tables do not fit the
// LDS: pass 2's cold lanes read a compact table block per (row type, neighbour type) kind (KindsDev).
//
// The reference scatters to owned neighbours (f[j] -= ...); here a bead also evaluates what
// each neighbour's own visit of the pair would send it (the closure with the roles swapped,
// the neighbour's CV force), so nothing is scattered and the sums are formed in row order.
// Decisions where the shipped text is undefined are listed in DESIGN.md (App. B #7-#12).
// Same arithmetic contract as ucg_pair.hip (-ffp-contract=off, ucg_math.h for exp/log/tanh).
#include "ucg_pair_dev.h"

namespace ucg {

namespace {

constexpr int DENS_BLOCK = 256;

// (r - rth) / (0.1 rth): the divisor is a per-type constant, so the quotient is formed with its
// reciprocal and two FMA residual steps (exactly the IEEE quotient, see div_by_const) when the
// divisor qualifies; w = 0.1 * rth and rw = 1 / w come from the caller
__device__ __forceinline__ double prox_arg(const double r, const double rth, const double w, const double rw, const bool ok)
{
  return ok ? div_by_const(r - rth, w, rw) : (r - rth) / w;
}

__device__ __forceinline__ double prox_fn_t(const double t) { return 0.5 * (1.0 - t); }
__device__ __forceinline__ double prox_der_t(const double t, const double w) { return 0.5 * (1.0 - t * t) / w; }

// the workgroup's own beads staged in LDS (same idea as k_pair_gather: with Morton-sorted beads most
// neighbours of a bead are beads of its own workgroup; the rest go through L1/L2)
struct OwnBlock {
  const double4 *pos;
  const int *meta;
  int k0;
  unsigned nown;
};

__device__ __forceinline__ void stage_own_block(const AtomsDev &A, const int k0, double4 *s_pos, int *s_meta)
{
  for (int t = threadIdx.x; t < PAIR_BLOCK; t += blockDim.x)
    if (k0 + t < A.nlocal) {
      s_pos[t] = A.pos4[k0 + t];
      s_meta[t] = A.meta[k0 + t];
    }
  __syncthreads();
}

__device__ __forceinline__ void gather_bead(const AtomsDev &A, const OwnBlock &O, const int m, double4 &p, int &mt)
{
  const unsigned ml = (unsigned) (m - O.k0);
  if (ml < O.nown) {
    p = O.pos[ml];
    mt = O.meta[ml];
  } else {
    p = A.pos4[m];
    mt = A.meta[m];
  }
}

// per-type constants of a deck of several actual types, in LDS: the squared cutoffs [tk][tm] and, per type, {threshold radius,
// w = 0.1 * radius, 1 / w, flags as a small number: 1 = use_density, 2 = 1 / w qualifies for div_by_const} -- instead of loads through L1 per pair
constexpr int NA1MAX = UCG_MAX_ACTUAL + 1;

__device__ __forceinline__ void stage_type_pars(const PairDev &P, double *s_cutsq, double4 *s_tp)
{
  const int na1 = P.n_actual + 1;
  for (int t = threadIdx.x; t < na1 * na1; t += blockDim.x) s_cutsq[t] = P.cutsq[t];
  for (int t = threadIdx.x; t < na1; t += blockDim.x) {
    const double rth = P.dens_par[t * 2 + 1];
    const double w = 0.1 * rth;
    const int bits = (P.dens_flags[t * 2 + 0] == 1 ? 1 : 0) | (recip_ok(w) ? 2 : 0);
    s_tp[t] = make_double4(rth, w, 1.0 / w, (double) bits);
  }
}

__global__ __launch_bounds__(PAIR_BLOCK) void k_density_pass1(const PairDev P, const AtomsDev A, const ListDev Lst,
                                                             double2 *prior, double *partial0)
{
  __shared__ double4 s_pos[PAIR_BLOCK];
  __shared__ int s_meta[PAIR_BLOCK];
  __shared__ double s_cutsq[NA1MAX * NA1MAX];
  __shared__ double4 s_tp[NA1MAX];
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int k0 = chunk * PAIR_BLOCK;
  stage_type_pars(P, s_cutsq, s_tp);
  stage_own_block(A, k0, s_pos, s_meta);
  OwnBlock O{s_pos, s_meta, k0, (unsigned) max(0, min(PAIR_BLOCK, A.nlocal - k0))};
  const int k = k0 + threadIdx.x;
  if (k >= A.nlocal) return;
  const double4 pk = s_pos[threadIdx.x];
  const int tk = UCG_META_TYPE(s_meta[threadIdx.x]);
  const int na1 = P.n_actual + 1;
  if (P.dens_flags[tk * 2 + 0] == 1) {
    const double rth = P.dens_par[tk * 2 + 1];
    const double w = 0.1 * rth, rw = 1.0 / w;
    const bool wok = recip_ok(w);
    const bool onetype = P.n_actual == 1;
    const double cut11 = P.cutsq[na1 + 1];
    const int n = Lst.numneigh[k];
    const int *rp = Lst.neigh + k;
    const size_t pitch = (size_t) Lst.pitch;
    double *const tc = P.tcache ? P.tcache + k : nullptr;
    double rho = 0.0;
    // one-deep software pipeline: the next entry's bead is in flight while this one is evaluated
    int ent = n > 0 ? row_load(rp, Lst.stream_rows) : 0, ent_n = n > 1 ? row_load(rp + pitch, Lst.stream_rows) : ent;
    double4 pm;
    int mm;
    gather_bead(A, O, ent & 0x1FFFFFFF, pm, mm);
    rp += pitch;
    for (int e = 0; e < n; e++) {
      rp += pitch;
      const int ent_nn = (e + 2 < n) ? row_load(rp, Lst.stream_rows) : ent_n;
      double4 pm_n;
      int mm_n;
      gather_bead(A, O, ent_n & 0x1FFFFFFF, pm_n, mm_n);
      const int tm = UCG_META_TYPE(mm);
      const double dx = pk.x - pm.x, dy = pk.y - pm.y, dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      double cutv = cut11;
      if (!onetype) cutv = s_cutsq[tk * na1 + tm];  // (uniform branch)
      if (rsq < cutv) {
        const double t = ucg_tanh(prox_arg(sqrt(rsq), rth, w, rw, wok));
        rho += prox_fn_t(t);
        if (tc) tc[(size_t) e * pitch] = t;  // pass 3 needs tanh of the same argument for this entry
      }
      ent_n = ent_nn;
      pm = pm_n;
      mm = mm_n;
    }
    const double cth = P.dens_par[tk * 2 + 0];
    const double th = ucg_tanh((rho - cth) / (0.1 * cth));
    const double p0 = 0.5 + 0.5 * th;
    prior[k] = make_double2(p0, 1.0 - p0);
    partial0[k] = 0.5 * (1.0 - th * th) / (0.1 * cth);
  } else {
    const double e0 = ucg_exp_nb(-P.mu[tk * 2 + 0] / P.kT);
    const double e1 = ucg_exp_nb(-P.mu[tk * 2 + 1] / P.kT);
    double den = 0.0;
    den += e0;
    den += e1;
    prior[k] = make_double2(e0 / den, e1 / den);
    partial0[k] = 0.0;
  }
}

__global__ __launch_bounds__(DENS_BLOCK) void k_ghost_copy2(int ng, int nlocal, const int *ghost_src, double2 *arr)
{
  const int g = blockIdx.x * DENS_BLOCK + threadIdx.x;
  if (g < ng) arr[nlocal + g] = arr[ghost_src[g]];
}

// closure as shipped (:608-622): a = b - 1, no guards.  In two parts: the coupling (b = exp(-J / kT), a = b - 1) depends on
// the four energies only, the pair probabilities on it and on the two priors.
template <bool FAST>
__device__ __forceinline__ void closure_coupling(const double kT, const double rkT, const int kTp2, const double u00,
                                                 const double u01, const double u10, const double u11, double &aij, double &bij)
{
  const double Jij = u11 + u00 - u01 - u10;
  bij = ucg_exp_nb(FAST ? div_kT(-Jij, kT, rkT, kTp2) : -Jij / kT);
  aij = bij - 1.;
}

__device__ __forceinline__ void closure_probs(const double aij, const double bij, const double pi1, const double pj1, double &p00,
                                              double &p01, double &p10, double &p11)
{
  const double Qij = (pi1 + pj1) * aij + 1.;
  // (the discriminant is used as shipped, unguarded: a negative one gives NaN through the hardware form; in [2^-700, 2^700)
  // the bare rsq iteration returns the same bits -- ucg_sqrt_core, chosen per wavefront as in the Bethe kernel)
  const double disc = Qij * Qij - 4. * aij * bij * pi1 * pj1;
  double Dij;
  if (__builtin_amdgcn_ballot_w64(!(disc >= 0x1p-700 && disc < 0x1p+700)) != 0ull) Dij = sqrt(disc);
  else Dij = ucg_sqrt_core(disc);
  p11 = (Qij - Dij) / 2. / aij;
  p00 = 1. + p11 - pi1 - pj1;
  p10 = pi1 - p11;
  p01 = pj1 - p11;
}

// what one in-cutoff entry adds to its row owner (:597-676): scores, the closure, the pair force (x 1/2 when the neighbour is
// owned) and what the neighbour's own visit of the pair sends back, the energy / virial terms, the entropic accumulators
struct DensAcc {
  double fx, fy, fz, s0, s1, G0, G1;
};

template <bool FAST, bool EV>
__device__ __forceinline__ void density_pair_terms(const Quad &q, const bool mixed_same, const double kT, const double rkT,
                                                   const int kTp2, const int sm, const double pk1, const double pm1,
                                                   const bool m_owned, const bool dens_k, const double dx, const double dy,
                                                   const double dz, DensAcc &a, double (&ev)[8])
{
  // scores: only the row owner's (:597-603)
  if (FAST) {
    a.s0 -= div_kT(sm ? q.u01 : q.u00, kT, rkT, kTp2);
    a.s1 -= div_kT(sm ? q.u11 : q.u10, kT, rkT, kTp2);
  } else {
    a.s0 -= (sm ? q.u01 : q.u00) / kT;
    a.s1 -= (sm ? q.u11 : q.u10) / kT;
  }
  double p00, p01, p10, p11;
  double aij, bij;
  closure_coupling<FAST>(kT, rkT, kTp2, q.u00, q.u01, q.u10, q.u11, aij, bij);
  closure_probs(aij, bij, pk1, pm1, p00, p01, p10, p11);
  double evdwl = p00 * q.u00 + p01 * q.u01 + p10 * q.u10 + p11 * q.u11;
  double fpair = p00 * q.f00 + p01 * q.f01 + p10 * q.f10 + p11 * q.f11;
  if (m_owned) {
    evdwl = evdwl * 0.5;
    fpair = fpair * 0.5;
  }
  a.fx += dx * fpair;
  a.fy += dy * fpair;
  a.fz += dz * fpair;
  if (m_owned) {
    // what m's own visit of this pair sends to k: roles swapped (its u[a][b] is our u[b][a])
    // With ONE table for both mixed states (pairs of one actual type: eval_quad copies u10 = u01) its J is our J
    // bit for bit -- (u11 + u00 - x) - x either way -- and so are b = exp(-J / kT) and a: only a pair of two types
    // evaluates the exponential a second time.
    double t00, t01, t10, t11, am = aij, bm = bij;
    if (!mixed_same) closure_coupling<FAST>(kT, rkT, kTp2, q.u00, q.u10, q.u01, q.u11, am, bm);
    closure_probs(am, bm, pm1, pk1, t00, t01, t10, t11);
    double fpj = t00 * q.f00 + t01 * q.f10 + t10 * q.f01 + t11 * q.f11;
    fpj = fpj * 0.5;
    const double djx = -dx, djy = -dy, djz = -dz;
    a.fx -= djx * fpj;
    a.fy -= djy * fpj;
    a.fz -= djz * fpj;
  }
  if (EV) {
    const double sc = m_owned ? 1.0 : 0.5;
    ev[0] += m_owned ? evdwl : 0.5 * evdwl;
    ev[1] += sc * (dx * dx * fpair);
    ev[2] += sc * (dy * dy * fpair);
    ev[3] += sc * (dz * dz * fpair);
    ev[4] += sc * (dx * dy * fpair);
    ev[5] += sc * (dx * dz * fpair);
    ev[6] += sc * (dy * dz * fpair);
  }
  if (dens_k) {
    a.G0 -= (q.u10 - q.u00 + kT * ucg_log_nb(p10 / p00));
    a.G1 -= (q.u11 - q.u01 + kT * ucg_log_nb(p11 / p01));
  }
}

// KTP2: 1 = kT is a power of two (known on the host): u / kT is the exact product u * (1 / kT) and the general quotient's
// branch goes at compile time; -1 = P.kT_pow2 decides at run time
template <int TS, bool EV, bool LDS_TAB, bool FAST, int KTP2 = -1>
__global__ __launch_bounds__(PAIR_BLOCK) void k_density_pass2(const PairDev P, const AtomsDev A, const ListDev Lst,
                                                             const double2 *prior, const double *partial0, double2 *cv,
                                                             double *evpart, int *errflag)
{
  extern __shared__ double4 s_tab[];
  __shared__ double s_red[(PAIR_BLOCK / 64) * 8];
  __shared__ double4 s_par[UCG_MAX_TABLES];
  __shared__ int s_pairtab[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1) * 4];
  __shared__ double s_cutsq[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1)];
  __shared__ int2 s_kdir[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1)];
  // tables through L1 / L2: the cold lanes read the compact block of their pair's kind (KindsDev::kind_tab) when there is one
  const bool kcold = !LDS_TAB && FAST && TS != 3 && P.kinds.kind_tab != nullptr;

  const int ntabent = FAST ? (P.tablength * P.fast_stride + 1) / 2 : P.ntab * P.tablength;
  {
    const int na1sq = (P.n_actual + 1) * (P.n_actual + 1);
    if (kcold)
      for (int t = threadIdx.x; t < na1sq; t += blockDim.x) s_kdir[t] = P.kinds.kind_dir[t];
    for (int t = threadIdx.x; t < P.ntab; t += blockDim.x) s_par[t] = P.tabpar[t];
    for (int t = threadIdx.x; t < na1sq * 4; t += blockDim.x) s_pairtab[t] = P.pairtab[t];
    for (int t = threadIdx.x; t < na1sq; t += blockDim.x) s_cutsq[t] = P.cutsq[t];
    if (LDS_TAB)
      for (int t = threadIdx.x; t < ntabent; t += blockDim.x) s_tab[t] = (FAST ? P.tab_fast : P.tab)[t];
  }
  // tables through L1 / L2 with one actual type's block in LDS all the same (PairDev::hot_type)
  const int hot_ent = (!LDS_TAB && FAST && TS != 3) ? P.hot_ent : 0;
  for (int t = threadIdx.x; t < hot_ent; t += blockDim.x) s_tab[t] = P.tab_hot[t];
  // the workgroup's own beads behind the tables, as in k_pair_gather; the 4th component carries
  // the bead's prior p1 (lambda is not used by this style)
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int k0 = chunk * PAIR_BLOCK;
  const bool stage_own = P.stage_own != 0;
  double4 *s_ownpos = s_tab + (LDS_TAB ? ntabent : hot_ent);
  int *s_ownmeta = reinterpret_cast<int *>(s_ownpos + PAIR_BLOCK);
  if (stage_own) {
    for (int t = threadIdx.x; t < PAIR_BLOCK; t += blockDim.x)
      if (k0 + t < A.nlocal) {
        double4 p = A.pos4[k0 + t];
        p.w = prior[k0 + t].y;
        s_ownpos[t] = p;
        s_ownmeta[t] = A.meta[k0 + t];
      }
  }
  __syncthreads();
  const unsigned nown = stage_own ? (unsigned) max(0, min(PAIR_BLOCK, A.nlocal - k0)) : 0u;

  const int k = k0 + threadIdx.x;
  const int nlocal = A.nlocal;
  const int na1 = P.n_actual + 1;
  const double kT = P.kT, rkT = P.rkT;
  const int kTp2 = KTP2 >= 0 ? KTP2 : P.kT_pow2;
  double ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int err = 0;
  RangeTrack rtrack = range_track_init();

  if (k < nlocal) {
    const double4 pk = A.pos4[k];
    const int tk = UCG_META_TYPE(A.meta[k]);
    const int n = Lst.numneigh[k];
    const int *rp = Lst.neigh + k;
    const size_t pitch = (size_t) Lst.pitch;
    const double2 prk = prior[k];
    auto gather = [&](const int entw, double4 &p, int &mt) {
      const int m1 = entw & 0x1FFFFFFF;
      const unsigned ml = (unsigned) (m1 - k0);
      if (ml < nown) {
        p = s_ownpos[ml];
        mt = s_ownmeta[ml];
      } else {
        p = A.pos4[m1];
        p.w = prior[m1].y;
        mt = A.meta[m1];
      }
    };
    const bool dens_k = P.dens_flags[tk * 2 + 0] == 1;

    DensAcc acc{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (dens_k) {
      // one-body terms (:302-314); jnum is the whole row, skin included, as shipped
      const double jnum_f = 1. - n;
      const double mu0 = P.mu[tk * 2 + 0], mu1 = P.mu[tk * 2 + 1];
      if (P.dens_flags[tk * 2 + 1]) {
        acc.G0 -= kT * ucg_log_nb(prk.x) * jnum_f;
        acc.G1 -= kT * ucg_log_nb(prk.y) * jnum_f;
      }
      acc.G0 -= mu0;
      acc.s0 -= mu0 / kT;
      acc.G1 -= mu1;
      acc.s1 -= mu1 / kT;
    }

    // the shared grid's parameters (FAST) and, for one actual type, the cutoff as scalars instead of LDS reads per pair
    const bool onetype = P.n_actual == 1;
    const double cut11 = P.cutsq[na1 + 1];
    double4 parF = make_double4(0, 0, 0, 0);
    if (FAST) {
      const double4 pg = P.tabpar[0];
      parF = make_double4(uniform_f64(pg.x), uniform_f64(pg.y), uniform_f64(pg.z), uniform_f64(pg.w));
    }
    // one-deep software pipeline: the next entry's bead is in flight while this one is evaluated
    int ent = n > 0 ? row_load(rp, Lst.stream_rows) : 0, ent_n = n > 1 ? row_load(rp + pitch, Lst.stream_rows) : ent;
    double4 pm;
    int mm;
    gather(ent, pm, mm);
    rp += pitch;
    for (int e = 0; e < n; e++) {
      rp += pitch;
      const int ent_nn = (e + 2 < n) ? row_load(rp, Lst.stream_rows) : ent_n;
      double4 pm_n;
      int mm_n;
      gather(ent_n, pm_n, mm_n);
      const int m = ent & 0x1FFFFFFF;
      double factor_lj = 1.0;
      if (!FAST) {
        const int sb = (ent >> 30) & 3;
        factor_lj = sb == 0 ? P.special_lj[0] : sb == 1 ? P.special_lj[1] : sb == 2 ? P.special_lj[2] : P.special_lj[3];
      }
      const int tm = UCG_META_TYPE(mm);
      const int sm = UCG_META_STATE(mm);
      const double dx = pk.x - pm.x, dy = pk.y - pm.y, dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      double cutv = cut11;
      if (!onetype) cutv = s_cutsq[tk * na1 + tm];  // (uniform branch: a deck of several types keeps its per-pair LDS read)
      if (rsq < cutv) {
        const int *pt = s_pairtab + (tk * na1 + tm) * 4;
        Quad q;
        if (LDS_TAB) eval_quad<TS, FAST>(s_tab, s_par, pt, P.tablength, P.tlm1, P.fast_stride, rsq, factor_lj, q, err, rtrack,
                                         nullptr, false, -1, FAST ? &parF : nullptr);
        else if (kcold) {
          // per lane: the LDS hot block (stride 7, {t00, t01 = t10, t11}) or the kind's block in global memory
          const bool hot = hot_ent && tk == P.hot_type && tm == P.hot_type;
          const int2 kd = s_kdir[tk * na1 + tm];
          const int stride = hot ? 7 : 2 * kd.y + 1;
          const double2 *base = hot ? reinterpret_cast<const double2 *>(s_tab)
                                    : reinterpret_cast<const double2 *>(P.kinds.kind_tab) + 2 * (size_t) kd.x;
          const bool three = hot || kd.y == 3;
          eval_quad_kind<TS>(base, stride, three ? 2 : 4, three ? 4 : 6, parF, P.tlm1, rsq, q, rtrack);
        } else eval_quad<TS, FAST>(FAST ? P.tab_fast : P.tab, s_par, pt, P.tablength, P.tlm1, P.fast_stride, rsq, factor_lj, q, err, rtrack,
                                 hot_ent ? reinterpret_cast<const double2 *>(s_tab) : nullptr, tk == P.hot_type && tm == P.hot_type, P.hot_k0);
        density_pair_terms<FAST, EV>(q, pt[1] == pt[2], kT, rkT, kTp2, sm, prk.y, pm.w, m < nlocal, dens_k, dx, dy, dz, acc, ev);
      }
      ent = ent_n;
      ent_n = ent_nn;
      pm = pm_n;
      mm = mm_n;
    }
    A.frc4[k] = make_double4(acc.fx, acc.fy, acc.fz, 0.0);
    A.scores[k] = make_double2(acc.s0, acc.s1);
    A.num_ucgstates[k] = 2;
    // posterior (:678-689), index fixed to the bead's type (App. B #8)
    {
      const double e0 = ucg_exp_nb(acc.s0), e1 = ucg_exp_nb(acc.s1);
      double den = 0.0;
      den += e0;
      den += e1;
      A.ucgp[k] = e1 / den;
    }
    double2 c = make_double2(0.0, 0.0);
    if (dens_k) {
      const double pa = partial0[k];
      c.x = acc.G0 * pa;
      c.y = acc.G1 * (-pa);
    }
    cv[k] = c;
  }
  if (FAST) err |= range_flags(s_par[0], P.tlm1, rtrack);
  if (err) atomicOr(errflag, err);
  if (EV) block_sum_store<8>(ev, s_red, evpart);
}

template <bool EV>
__global__ __launch_bounds__(PAIR_BLOCK) void k_density_pass3(const PairDev P, const AtomsDev A, const ListDev Lst,
                                                             const double2 *cv, double *evpart)
{
  __shared__ double s_red[(PAIR_BLOCK / 64) * 8];
  __shared__ double4 s_pos[PAIR_BLOCK];
  __shared__ int s_meta[PAIR_BLOCK];
  __shared__ double2 s_cv[PAIR_BLOCK];
  __shared__ double s_cutsq[NA1MAX * NA1MAX];
  __shared__ double4 s_tp[NA1MAX];
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int k0 = chunk * PAIR_BLOCK;
  stage_type_pars(P, s_cutsq, s_tp);
  if (k0 + (int) threadIdx.x < A.nlocal) s_cv[threadIdx.x] = cv[k0 + threadIdx.x];
  stage_own_block(A, k0, s_pos, s_meta);
  OwnBlock O{s_pos, s_meta, k0, (unsigned) max(0, min(PAIR_BLOCK, A.nlocal - k0))};
  const int k = k0 + threadIdx.x;
  const int na1 = P.n_actual + 1;
  double ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (k < A.nlocal) {
    const double4 pk = s_pos[threadIdx.x];
    const int tk = UCG_META_TYPE(s_meta[threadIdx.x]);
    const bool dens_k = P.dens_flags[tk * 2 + 0] == 1;
    const double rth_k = P.dens_par[tk * 2 + 1];
    const double w_k = 0.1 * rth_k, rw_k = 1.0 / w_k;
    const bool wok_k = recip_ok(w_k);
    const bool onetype = P.n_actual == 1;
    const double cut11 = P.cutsq[na1 + 1];
    const double2 cvk = s_cv[threadIdx.x];
    double4 f = A.frc4[k];
    const int n = Lst.numneigh[k];
    const int *rp = Lst.neigh + k;
    const size_t pitch = (size_t) Lst.pitch;
    int ent = n > 0 ? row_load(rp, Lst.stream_rows) : 0, ent_n = n > 1 ? row_load(rp + pitch, Lst.stream_rows) : ent;
    double4 pm;
    int mm;
    gather_bead(A, O, ent & 0x1FFFFFFF, pm, mm);
    // pass 1's tanh of this bead's in-cutoff entries (valid where pass 1 evaluated the entry: dens_k and inside the cutoff --
    // the conditions of `in_k` below, on the same positions), fetched one entry ahead like the beads
    const double *tc = (P.tcache && dens_k) ? P.tcache + k : nullptr;
    const int strm = Lst.stream_rows;
    double tc_cur = (tc && n > 0) ? (strm ? __builtin_nontemporal_load(tc) : tc[0]) : 0.0;
    rp += pitch;
    for (int e = 0; e < n; e++) {
      rp += pitch;
      const int ent_nn = (e + 2 < n) ? row_load(rp, Lst.stream_rows) : ent_n;
      double4 pm_n;
      int mm_n;
      gather_bead(A, O, ent_n & 0x1FFFFFFF, pm_n, mm_n);
      const double *tcp = tc + (size_t) (e + 1) * pitch;
      const double tc_n = (tc && e + 1 < n) ? (strm ? __builtin_nontemporal_load(tcp) : *tcp) : 0.0;
      const int m = ent & 0x1FFFFFFF;
      const int tm = UCG_META_TYPE(mm);
      const double dx = pk.x - pm.x, dy = pk.y - pm.y, dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      double cut_km = cut11, cut_mk = cut11;
      bool dens_m = dens_k;
      double4 tpm = make_double4(rth_k, w_k, rw_k, 0.0);
      if (!onetype) {  // (uniform branch)
        cut_km = s_cutsq[tk * na1 + tm];
        cut_mk = s_cutsq[tm * na1 + tk];
        tpm = s_tp[tm];
        dens_m = ((int) tpm.w & 1) != 0;
      }
      const bool in_k = dens_k && rsq < cut_km;
      const bool in_m = dens_m && rsq < cut_mk;
      if (in_k || in_m) {
        const double distance = sqrt(rsq);
        // the four quotients (cv * w) / distance share the divisor: its reciprocal + two FMA residual
        // steps give the IEEE quotient (div_by_const); other divisors keep the hardware division
        const double rdist = 1.0 / distance;
        const bool dok = recip_ok(distance);
        double w_own = 0.0;
        if (in_k) {
          const double t = tc ? tc_cur : ucg_tanh(prox_arg(distance, rth_k, w_k, rw_k, wok_k));
          w_own = P.dens_as_shipped ? prox_fn_t(t) : prox_der_t(t, w_k);
          for (int s = 0; s < 2; s++) {
            const double num = (s ? cvk.y : cvk.x) * w_own;
            const double fpair = dok ? div_by_const(num, distance, rdist) : num / distance;
            f.x += fpair * dx;
            f.y += fpair * dy;
            f.z += fpair * dz;
            if (EV) {
              ev[1] += dx * dx * fpair;
              ev[2] += dy * dy * fpair;
              ev[3] += dz * dz * fpair;
              ev[4] += dx * dy * fpair;
              ev[5] += dx * dz * fpair;
              ev[6] += dy * dz * fpair;
            }
          }
        }
        if (in_m) {
          double w;
          if (tpm.x == rth_k && in_k) {
            w = w_own;  // same threshold radius: same arithmetic (prox_arg is the IEEE quotient), same value
          } else {
            const bool wok_m = onetype ? wok_k : ((int) tpm.w & 2) != 0;
            const double t = ucg_tanh(prox_arg(distance, tpm.x, tpm.y, tpm.z, wok_m));
            w = P.dens_as_shipped ? prox_fn_t(t) : prox_der_t(t, tpm.y);
          }
          const unsigned ml = (unsigned) (m - k0);
          const double2 cvm = ml < O.nown ? s_cv[ml] : cv[m];
          const double djx = pm.x - pk.x, djy = pm.y - pk.y, djz = pm.z - pk.z;
          for (int s = 0; s < 2; s++) {
            const double num = (s ? cvm.y : cvm.x) * w;
            const double fpair = dok ? div_by_const(num, distance, rdist) : num / distance;
            f.x -= fpair * djx;
            f.y -= fpair * djy;
            f.z -= fpair * djz;
          }
        }
      }
      ent = ent_n;
      ent_n = ent_nn;
      pm = pm_n;
      mm = mm_n;
      tc_cur = tc_n;
    }
    A.frc4[k] = f;
  }
  if (EV) block_sum_store<8>(ev, s_red, evpart);
}

__global__ void k_ev_final2(const double *part, int nb1, const double *part3, int nb3, double *out)
{
  const int c = threadIdx.x;
  if (c < 8) {
    double s = 0.0;
    for (int b = 0; b < nb1; b++) s += part[(size_t) b * 8 + c];
    for (int b = 0; b < nb3; b++) s += part3[(size_t) b * 8 + c];
    out[c] = s;
  }
}

template <int TS>
hipError_t launch_pass2(const PairDev &Pin, const AtomsDev &A, const ListDev &L, bool ev, const double2 *prior,
                        const double *partial0, double2 *cv, double *evpart, int *errflag, hipStream_t st, int nblocks)
{
  PairDev P = Pin;
  {
    // own-bead staging needs the full 1024-bead block behind the tables (the gather kernels may run with
    // fewer beads per block)
    const size_t tb = P.fast ? ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4)
                             : (size_t) P.ntab * P.tablength * sizeof(double4);
    const size_t hot = (!P.tab_in_lds && P.fast) ? (size_t) P.hot_ent * sizeof(double4) : 0;
    size_t used = (P.tab_in_lds ? tb : hot) + (size_t) PAIR_BLOCK * (sizeof(double4) + sizeof(int)) + 6 * 1024;
    if (hot && used > 160 * 1024) {  // the hot block and the full 1024-bead staging do not both fit: the staging stays
      P.hot_type = 0;
      P.hot_ent = 0;
      used -= hot;
    }
    P.stage_own = (Pin.stage_own_allowed && used <= 160 * 1024) ? 1 : 0;
  }
  const size_t tabbytes = P.fast ? ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4)
                                 : (size_t) P.ntab * P.tablength * sizeof(double4);
  const size_t ownbytes = P.stage_own ? (size_t) PAIR_BLOCK * (sizeof(double4) + sizeof(int)) : 0;
  const size_t ldsbytes = (P.tab_in_lds ? tabbytes : (P.fast ? (size_t) P.hot_ent * sizeof(double4) : 0)) + ownbytes;
#define UCG_LAUNCH(EVF, LDSF, FASTF)                                                                     \
  do {                                                                                                   \
    auto kern = k_density_pass2<TS, EVF, LDSF, FASTF>;                                                   \
    if constexpr (FASTF) {                                                                               \
      if (P.kT_pow2) kern = k_density_pass2<TS, EVF, LDSF, FASTF, 1>;                                    \
    }                                                                                                    \
    if (ldsbytes > 48 * 1024) {                                                                          \
      hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int) ldsbytes);                                                \
      if (e != hipSuccess) return e;                                                                     \
    }                                                                                                    \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, prior, partial0, cv, evpart, \
                       errflag);                                                                         \
  } while (0)
  const int sel = (P.tab_in_lds ? 4 : 0) | (ev ? 2 : 0) | (P.fast ? 1 : 0);
  switch (sel) {
    case 0: UCG_LAUNCH(false, false, false); break;
    case 1: UCG_LAUNCH(false, false, true); break;
    case 2: UCG_LAUNCH(true, false, false); break;
    case 3: UCG_LAUNCH(true, false, true); break;
    case 4: UCG_LAUNCH(false, true, false); break;
    case 5: UCG_LAUNCH(false, true, true); break;
    case 6: UCG_LAUNCH(true, true, false); break;
    default: UCG_LAUNCH(true, true, true); break;
  }
#undef UCG_LAUNCH
  return hipGetLastError();
}

// BITMAP tables: generic path only (see launch_style_bitmap in ucg_pair.hip)
hipError_t launch_pass2_bitmap(const PairDev &Pin, const AtomsDev &A, const ListDev &L, bool ev, const double2 *prior,
                               const double *partial0, double2 *cv, double *evpart, int *errflag, hipStream_t st, int nblocks)
{
  if (Pin.tab_in_lds || Pin.fast) return hipErrorInvalidValue;
  PairDev P = Pin;
  P.stage_own = Pin.stage_own_allowed ? 1 : 0;
  const size_t ldsbytes = P.stage_own ? (size_t) PAIR_BLOCK * (sizeof(double4) + sizeof(int)) : 0;
  if (ev) hipLaunchKernelGGL((k_density_pass2<3, true, false, false>), dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, prior, partial0, cv, evpart, errflag);
  else hipLaunchKernelGGL((k_density_pass2<3, false, false, false>), dim3(nblocks), dim3(PAIR_BLOCK), ldsbytes, st, P, A, L, prior, partial0, cv, evpart, errflag);
  return hipGetLastError();
}

}  // namespace

// prior / cv: [nlocal + nghost] double2; partial0: [nlocal]; evpart: 8 * (blocks2 + blocks3) doubles
hipError_t launch_density(const PairDev &P, const AtomsDev &A, const ListDev &L, const int *ghost_src, bool ev,
                          double2 *prior, double *partial0, double2 *cv, double *evpart, double *evout, int *errflag,
                          hipStream_t st)
{
  const int n = A.nlocal;
  if (n == 0) return hipSuccess;
  const int nb2 = (n + PAIR_BLOCK - 1) / PAIR_BLOCK;
  const int ngb = (A.nghost + DENS_BLOCK - 1) / DENS_BLOCK;
  hipLaunchKernelGGL(k_density_pass1, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, prior, partial0);
  if (A.nghost > 0) hipLaunchKernelGGL(k_ghost_copy2, dim3(ngb), dim3(DENS_BLOCK), 0, st, A.nghost, n, ghost_src, prior);
  hipError_t e;
  switch (P.tabstyle) {
    case 0: e = launch_pass2<0>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
    case 1: e = launch_pass2<1>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
    case 3: e = launch_pass2_bitmap(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
    default: e = launch_pass2<2>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
  }
  if (e != hipSuccess) return e;
  if (A.nghost > 0) hipLaunchKernelGGL(k_ghost_copy2, dim3(ngb), dim3(DENS_BLOCK), 0, st, A.nghost, n, ghost_src, cv);
  double *evpart3 = evpart + (size_t) nb2 * 8;
  if (ev) {
    hipLaunchKernelGGL(k_density_pass3<true>, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, cv, evpart3);
    hipLaunchKernelGGL(k_ev_final2, dim3(1), dim3(64), 0, st, evpart, nb2, evpart3, nb2, evout);
  } else {
    hipLaunchKernelGGL(k_density_pass3<false>, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, cv, evpart3);
  }
  return hipGetLastError();
}

// the same three passes one at a time, for decomposed runs: the caller refreshes the ghosts' entries of
// `prior` (after phase 1) and of `cv` (after phase 2) from their owner ranks in between
hipError_t launch_density_phase(const PairDev &P, const AtomsDev &A, const ListDev &L, int phase, bool ev, double2 *prior,
                                double *partial0, double2 *cv, double *evpart, double *evout, int *errflag,
                                hipStream_t st)
{
  const int n = A.nlocal;
  if (n == 0) return hipSuccess;
  const int nb2 = (n + PAIR_BLOCK - 1) / PAIR_BLOCK;
  if (phase == 1) {
    hipLaunchKernelGGL(k_density_pass1, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, prior, partial0);
  } else if (phase == 2) {
    hipError_t e;
    switch (P.tabstyle) {
      case 0: e = launch_pass2<0>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
      case 1: e = launch_pass2<1>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
      case 3: e = launch_pass2_bitmap(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
      default: e = launch_pass2<2>(P, A, L, ev, prior, partial0, cv, evpart, errflag, st, nb2); break;
    }
    if (e != hipSuccess) return e;
  } else {
    double *evpart3 = evpart + (size_t) nb2 * 8;
    if (ev) {
      hipLaunchKernelGGL(k_density_pass3<true>, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, cv, evpart3);
      hipLaunchKernelGGL(k_ev_final2, dim3(1), dim3(64), 0, st, evpart, nb2, evpart3, nb2, evout);
    } else {
      hipLaunchKernelGGL(k_density_pass3<false>, dim3(nb2), dim3(PAIR_BLOCK), 0, st, P, A, L, cv, evpart3);
    }
  }
  return hipGetLastError();
}

int density_evpart_doubles(int nlocal)
{
  return 8 * ((nlocal + DENS_BLOCK - 1) / DENS_BLOCK + (nlocal + PAIR_BLOCK - 1) / PAIR_BLOCK) + 16;
}

}  // namespace ucg
