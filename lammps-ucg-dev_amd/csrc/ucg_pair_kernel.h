// ucg_pair_kernel.h -- the neighbour-loop kernel template of table_ucgld and table_ucg_bethe (gfx950), instantiated by
// ucg_pair_hot.hip (tables in LDS on one shared r^2 grid, one or two lanes per bead: the tuned variants) and
// ucg_pair.hip (every other variant).
//
// What is computed: Scenario 4 of PairTable_UCGLD::compute
// (UCG/pair_table_ucgld.cpp:424-533) and of PairTable_UCG_Bethe::compute
// (UCG/pair_table_ucg_bethe.cpp:457-620), with the prologues :170-180 / :155-162.
//
// How: one lane per owned bead, gathering over that bead's row of a FULL,
// row-transposed neighbour list; nothing is scattered, so there are no atomics
// and the per-bead sums are formed in one fixed order (row order).  Each pair is
// evaluated in the reference's (i,j) orientation -- bit 29 of the entry says
// whether the row owner is "i" -- so the numbers added to a bead are exactly the
// numbers the reference's half-list sweep adds to it; only the order of the
// additions is the canonical one ("determinism contract").
//
// All tables are staged in LDS (<= 160 KB per CU) when they fit; one workgroup
// of 1024 lanes per CU then owns the whole LDS and 4 waves per SIMD hide the
// gather latency.  Tables that do not fit are read through L1/L2.
//
// FAST variants (chosen on the host when every table shares one r^2 grid, all
// special_lj are 1 and kT is a usable divisor) do the same arithmetic with less
// work: the knot index and the interpolation basis are computed once per pair
// instead of once per table (identical inputs give identical bits), the
// multiplications by factor_lj = 1.0 are dropped (x*1.0 == x), and u/kT uses a
// host-computed reciprocal with two fused residual corrections, which returns the
// correctly rounded quotient (Markstein): same bits as the IEEE division.
//
// Compiled with -ffp-contract=off: every product and sum below rounds exactly
// where the reference's scalar x86-64 code rounds.
#pragma once

#include "ucg_pair_dev.h"

namespace ucg {

namespace {

// A neighbour's record {x, y, z, lambda | type, state}: from the workgroup's LDS copy when it is one of its own beads,
// else through L1 / L2.  ONE instruction stream: the LDS or the global address is selected per lane as a generic
// pointer and read with generic-address (flat) loads.  The alternative -- an LDS read and a global load, each under its
// lane mask -- is SLOWER (round 2: 422 -> 472 us at 1 M beads; round 3's build fell into it unnoticed, 407 -> 455 us on one
// box, profiles/r04_ab_r02_vs_r03.json): both branches write the same registers, so every LDS read waits for the other
// lanes' outstanding global loads (vmcnt), and the two serial branch bodies cost more than the flat path.  The compiler
// turns a plain `cond ? lds[i] : global[j]` into either form depending on the surrounding code, so the selected pointers
// are passed through an empty asm statement: behind it their address space is unknown and flat loads are the only choice.
__device__ __forceinline__ void gather_bead_split(const AtomsDev &A, const double4 *s_ownpos, const int *s_ownmeta, const int k0,
                                                  const unsigned nown, const int m, double4 &pm, int &mm)
{
  const unsigned ml = (unsigned) (m - k0);
  const bool own = ml < nown;
  const double4 *pp = own ? s_ownpos + ml : A.pos4 + m;
  const int *mp = own ? s_ownmeta + ml : A.meta + m;
  asm volatile("" : "+v"(pp), "+v"(mp));
  pm = *pp;
  mm = *mp;
}

// SCE (table_ucg_bethe): -1 = P.pseudo_flag decides at run time; 0 = pseudo-likelihood scores only (`pseudo yes`), the
// full-SCE code and the per-row reciprocals it keeps in registers are compiled out; 1 = full SCE (`pseudo no`)
// KTP2: 1 = kT is a power of two (known on the host): u / kT is the exact product u * (1 / kT) and the general quotient's
// code (a uniform branch and a range test per division) is compiled out; -1 = P.kT_pow2 decides at run time
// STREAM: the rows are read with non-temporal loads (lists larger than the last-level cache, ListDev::stream_rows: read once
// per launch, they would displace the beads the gathers re-read; 400 -> 392 us at 1 M beads).  A compile-time choice: the same
// choice behind a uniform run-time flag costs the gain.
template <int STYLE, int TS, bool EV, bool LDS_TAB, bool FAST, int SLOTS, bool ONETYPE = false, int SCE = -1, int KTP2 = -1,
          bool STREAM = false>
__global__ __launch_bounds__(PAIR_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pair_gather(const PairDev P, const AtomsDev A,
                                                           const ListDev Lst, double *evpart,
                                                           int *errflag)
{
  extern __shared__ double4 s_tab[];
  __shared__ double s_red[(PAIR_BLOCK / 64) * 8];
  // the small per-model arrays (bounded by UCG_MAX_ACTUAL / UCG_MAX_TABLES at upload time)
  __shared__ double4 s_par[UCG_MAX_TABLES];
  __shared__ int s_pairtab[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1) * 4];
  __shared__ double s_cutsq[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1)];

  // filtered launches map workgroups round-robin (their live workgroups are neighbours in Morton order and
  // would otherwise pile up on few XCDs)
  const int chunk_id = Lst.blockflag ? (int) blockIdx.x : xcd_chunk(blockIdx.x, gridDim.x);
  if (Lst.blockflag && Lst.blockflag[chunk_id] != Lst.blockwant) return;  // whole workgroup
  // in double4 units; the FAST layout is tablength * (2*ntab+1) 16-byte slots
  const int ntabent = FAST ? (P.tablength * P.fast_stride + 1) / 2 : P.ntab * P.tablength;
  // the workgroup's own beads, staged in LDS behind the tables (when they fit): beads are sorted
  // along a Morton curve, so ~3/4 of a bead's neighbours are beads of its own workgroup and are then
  // read from LDS instead of through the vector L1, whose tag rate (one line per lane per load) is
  // what bounds this kernel otherwise.  Same values either way.
  const bool stage_own = P.stage_own != 0;
  // tables through L1 / L2 with one actual type's block in LDS all the same (PairDev::hot_type)
  const int hot_ent = (!LDS_TAB && FAST && TS != 3) ? P.hot_ent : 0;
  const bool kcold = !LDS_TAB && FAST && TS != 3 && P.kinds.kind_tab != nullptr;
  double4 *s_ownpos = s_tab + (LDS_TAB ? ntabent : hot_ent);
  int *s_ownmeta = reinterpret_cast<int *>(s_ownpos + PAIR_BLOCK / SLOTS);
  // table_ucg_bethe: ucgp of the workgroup's own beads as well (8 bytes per bead behind the meta words)
  double *s_ownucgp = reinterpret_cast<double *>(s_ownmeta + PAIR_BLOCK / SLOTS);
  const int k0 = chunk_id * (PAIR_BLOCK / SLOTS);
  if (stage_own) {
    for (int t = threadIdx.x; t < PAIR_BLOCK / SLOTS; t += blockDim.x) {
      if (k0 + t < A.nlocal) {
        s_ownpos[t] = A.pos4[k0 + t];
        s_ownmeta[t] = A.meta[k0 + t];
        if (STYLE == 1) s_ownucgp[t] = A.ucgp[k0 + t];
      }
    }
  }
  const unsigned nown = stage_own ? (unsigned) min(PAIR_BLOCK / SLOTS, A.nlocal - k0) : 0u;
  {
    const int na1sq = (P.n_actual + 1) * (P.n_actual + 1);
    for (int t = threadIdx.x; t < P.ntab; t += blockDim.x) s_par[t] = P.tabpar[t];
    // (tables through L1 / L2 on several actual types: the cold lanes read the compact block of their pair's kind, KindsDev,
    // and need its directory instead of the table ids -- in the same LDS words)
    if (kcold)
      for (int t = threadIdx.x; t < na1sq; t += blockDim.x) reinterpret_cast<int2 *>(s_pairtab)[t] = P.kinds.kind_dir[t];
    else
      for (int t = threadIdx.x; t < na1sq * 4; t += blockDim.x) s_pairtab[t] = P.pairtab[t];
    for (int t = threadIdx.x; t < na1sq; t += blockDim.x) s_cutsq[t] = P.cutsq[t];
    if (LDS_TAB)
      for (int t = threadIdx.x; t < ntabent; t += blockDim.x) s_tab[t] = (FAST ? P.tab_fast : P.tab)[t];
    for (int t = threadIdx.x; t < hot_ent; t += blockDim.x) s_tab[t] = P.tab_hot[t];
    __syncthreads();
  }

  // SLOTS lanes share one bead: lane `slot` takes the row entries e = slot, slot+SLOTS, ...;
  // adjacent lanes then gather adjacent list entries (mostly adjacent beads: shared cache lines),
  // and the SLOTS partial sums are combined by a fixed shuffle tree (the canonical order).
  const int chunk = chunk_id;
  const int gtid = chunk * PAIR_BLOCK + threadIdx.x;
  const int k = gtid / SLOTS;
  const int slot = gtid % SLOTS;
  const int nlocal = A.nlocal;
  const int na1 = P.n_actual + 1;
  const double kT = P.kT, rkT = P.rkT;
  const int kTp2 = KTP2 >= 0 ? KTP2 : P.kT_pow2;
  const int pseudo_flag = SCE < 0 ? P.pseudo_flag : SCE;
  double ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int err = 0;
  RangeTrack rtrack = range_track_init();
  const bool active = k < nlocal;
  double4 pk = make_double4(0, 0, 0, 0);
  int mk = 0;
  double fx = 0.0, fy = 0.0, fz = 0.0, uf = 0.0, s0 = 0.0, s1 = 0.0;

  if (active) {
    pk = A.pos4[k];
    mk = A.meta[k];
    const int tk = UCG_META_TYPE(mk);
    const double lk = pk.w;
    const int n = Lst.numneigh[k];
    const size_t pitch = (size_t) Lst.pitch;
    const size_t rstep = pitch * SLOTS;
    const int *rp = Lst.neigh + k + (size_t) slot * pitch;

    const double mu0 = P.mu[tk * 2 + 0], mu1 = P.mu[tk * 2 + 1];
    if (slot == 0) {  // the prologue values (:170-180 / bethe :155-162) start slot 0's sums
      if (STYLE == 0) {
        const double mui = mu1 - mu0;
        uf -= mui;
        s1 -= mui / kT;
      } else {
        s0 = -mu0 / kT;
        s1 = -mu1 / kT;
      }
    }
    // priors of k for the Bethe closure
    double pk_as_i1 = 0.0, pk_as_j1 = 0.0, pk_as_i0 = 0.0, pk_as_j0 = 0.0;
    bool k_first_chempot = false;
    // first-call rules (:179-205, :227-253) only while the host says a marker may still be around (uniform): afterwards
    // the per-pair test and the selects behind it are skipped, and a marker met anyway is reported (error bit 16)
    // (the tuned Bethe variants -- ONETYPE with the score mode fixed -- are launched only while the flag is down: compiled out)
    constexpr bool STEADY = STYLE == 1 && ONETYPE && SCE >= 0;
    const bool firstp = STYLE == 1 && !STEADY && P.first_possible != 0;
    int upm_sign_or = 0;
    if (STYLE == 1) {
      const double upk = A.ucgp[k];
      const bool first = firstp && upk < -0.999;
      if (!firstp) upm_sign_or |= __double2hiint(upk);  // (a set ucgp lies in [1e-6, 1]: any sign bit is a marker)
      if (first && P.prior_flag == 0) {  // CHEMICAL_POTENTIAL
        pk_as_i0 = P.prior_type[tk * 2 + 0];
        pk_as_i1 = P.prior_type[tk * 2 + 1];
      } else {
        pk_as_i1 = lk;
        pk_as_i0 = 1.0 - lk;
      }
      if (first) {
        if (P.prior_flag == 0) {
          // as shipped the neighbour's first-call prior is looked up with the ROW owner's
          // type (UCG/pair_table_ucg_bethe.cpp:229-232); resolved per pair below
          k_first_chempot = true;
        } else {
          pk_as_j0 = 1.0 - lk;
          pk_as_j1 = lk;
        }
      } else {
        pk_as_j1 = upk;
        pk_as_j0 = 1.0 - upk;
      }
    }

    // FAST Bethe rows: the full-SCE scores divide by the ROW bead's priors (as "i" or as "j"), so
    // their reciprocals are formed once per row and each division becomes an exact
    // reciprocal-multiply with two FMA residual steps (div_by_const); denominators outside its
    // proven range keep the hardware division.
    double rk_i0 = 0.0, rk_i1 = 0.0, rk_j0 = 0.0, rk_j1 = 0.0;
    bool rk_ok = false;
    if (STYLE == 1 && FAST && pseudo_flag == 1) {
      rk_ok = !k_first_chempot && recip_ok(pk_as_i0) && recip_ok(pk_as_i1) && recip_ok(pk_as_j0) && recip_ok(pk_as_j1);
      if (rk_ok) {
        rk_i0 = 1.0 / pk_as_i0;
        rk_i1 = 1.0 / pk_as_i1;
        rk_j0 = 1.0 / pk_as_j0;
        rk_j1 = 1.0 / pk_as_j1;
      }
    }

    // one actual type (the usual UCG deck): cutoff and table ids are the same for every pair (ONETYPE: known at
    // compile time, so they stay scalars)
    const bool onetype = ONETYPE || (P.n_actual == 1);
    // read through the kernel-argument pointers (uniform addresses: scalar loads, the values live in SGPRs)
    const double cut11 = P.cutsq[na1 + 1];
    // the shared r^2 grid of the FAST kernels {innersq, delta, invdelta, deltasq6}: uniform, so it lives in scalar
    // registers for the whole row (read per pair from the LDS copy it cost two ds_read_b128 and an LDS round trip at the
    // head of every iteration)
    double4 parF = make_double4(0, 0, 0, 0);
    if (FAST) {
      const double4 pg = P.tabpar[0];
      parF = make_double4(uniform_f64(pg.x), uniform_f64(pg.y), uniform_f64(pg.z), uniform_f64(pg.w));
    }
    const int pt11_0 = P.pairtab[(na1 + 1) * 4 + 0], pt11_1 = P.pairtab[(na1 + 1) * 4 + 1];
    const int pt11_2 = P.pairtab[(na1 + 1) * 4 + 2], pt11_3 = P.pairtab[(na1 + 1) * 4 + 3];

    // two-stage software pipeline: while entry e is evaluated, the gather of entry e+SLOTS is in
    // flight and the list word of entry e+2*SLOTS is being fetched (no exposed index-load latency)
    int ent = (slot < n) ? (STREAM ? __builtin_nontemporal_load(rp) : rp[0]) : 0;
    int ent_n = (slot + SLOTS < n) ? (STREAM ? __builtin_nontemporal_load(rp + rstep) : rp[rstep]) : ent;
    double4 pm;
    int mm;
    gather_bead_split(A, s_ownpos, s_ownmeta, k0, nown, ent & 0x1FFFFFFF, pm, mm);
    // table_ucg_bethe: the neighbour's ucgp travels with its record through the software pipeline (read at its use, inside
    // the cutoff branch, its latency was exposed: 709 -> 684 us at 1 M beads), from the LDS copy for an own bead
    auto gather_ucgp = [&](const int mi) {
      const unsigned mlu = (unsigned) (mi - k0);
      const double *upp = (mlu < nown) ? s_ownucgp + mlu : A.ucgp + mi;
      asm volatile("" : "+v"(upp));
      return *upp;
    };
    double up_cur = STYLE == 1 ? gather_ucgp(ent & 0x1FFFFFFF) : 0.0;
    rp += rstep;
    for (int e = slot; e < n; e += SLOTS) {
      rp += rstep;
      const int ent_nn = (e + 2 * SLOTS < n) ? (STREAM ? __builtin_nontemporal_load(rp) : rp[0]) : ent_n;
      double4 pm_n;
      int mm_n;
      gather_bead_split(A, s_ownpos, s_ownmeta, k0, nown, ent_n & 0x1FFFFFFF, pm_n, mm_n);
      const double up_n = STYLE == 1 ? gather_ucgp(ent_n & 0x1FFFFFFF) : 0.0;

      const bool k_is_i = (ent >> 29) & 1;
      double factor_lj = 1.0;
      if (!FAST) {
        const int sb = (ent >> 30) & 3;
        factor_lj = sb == 0 ? P.special_lj[0] : sb == 1 ? P.special_lj[1] : sb == 2 ? P.special_lj[2] : P.special_lj[3];
      }
      const int tm = UCG_META_TYPE(mm);
      const int sm = UCG_META_STATE(mm);
      const double lm = pm.w;
      const double dx = pk.x - pm.x;
      const double dy = pk.y - pm.y;
      const double dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const double cutv = onetype ? cut11 : s_cutsq[tk * na1 + tm];
      if (rsq < cutv) {
        int pt[4];
        if (onetype) {
          pt[0] = pt11_0; pt[1] = pt11_1; pt[2] = pt11_2; pt[3] = pt11_3;
        } else if (kcold) {
          pt[0] = pt[1] = pt[2] = pt[3] = 0;  // (the kind's directory stands where the table ids were)
        } else {
          const int *ps = s_pairtab + (tk * na1 + tm) * 4;
          pt[0] = ps[0]; pt[1] = ps[1]; pt[2] = ps[2]; pt[3] = ps[3];
        }
        Quad q;
        if (LDS_TAB) eval_quad<TS, FAST, ONETYPE>(s_tab, s_par, pt, P.tablength, P.tlm1, P.fast_stride, rsq, factor_lj, q, err, rtrack,
                                                  nullptr, false, -1, FAST ? &parF : nullptr);
        else if (kcold) {
          // per lane: the LDS hot block (stride 7, {t00, t01 = t10, t11}) or the kind's block in global memory
          const bool hot = hot_ent && tk == P.hot_type && tm == P.hot_type;
          const int2 kd = reinterpret_cast<const int2 *>(s_pairtab)[tk * na1 + tm];
          const int stride = hot ? 7 : 2 * kd.y + 1;
          const double2 *base = hot ? reinterpret_cast<const double2 *>(s_tab)
                                    : reinterpret_cast<const double2 *>(P.kinds.kind_tab) + 2 * (size_t) kd.x;
          const bool three = hot || kd.y == 3;
          eval_quad_kind<TS>(base, stride, three ? 2 : 4, three ? 4 : 6, parF, P.tlm1, rsq, q, rtrack);
        } else eval_quad<TS, FAST>(FAST ? P.tab_fast : P.tab, s_par, pt, P.tablength, P.tlm1, P.fast_stride, rsq, factor_lj, q, err, rtrack,
                                 hot_ent ? reinterpret_cast<const double2 *>(s_tab) : nullptr, tk == P.hot_type && tm == P.hot_type, P.hot_k0);

        double evdwl = 0.0, fpair;
        if (STYLE == 0 || pseudo_flag == 0) {
          // pseudo-likelihood scores (:492-502): S[k][a] -= u[a][state of the neighbour] / kT
          const double ua = sm ? q.u01 : q.u00, ub = sm ? q.u11 : q.u10;
          if (FAST) {
            s0 -= div_kT(ua, kT, rkT, kTp2);
            s1 -= div_kT(ub, kT, rkT, kTp2);
          } else {
            s0 -= ua / kT;
            s1 -= ub / kT;
          }
        }
        if (STYLE == 0) {
          // lambda-bilinear mix (:507-517).  In the reference's orientation the sum is
          // ((t00 + t01) + t10) + t11; seen from the "j" bead the middle terms swap.
          const double w00 = (1. - lk) * (1. - lm);
          const double w11 = lk * lm;
          const double wA = (1. - lk) * lm;  // weight of own-frame (0,1)
          const double wB = (1. - lm) * lk;  // weight of own-frame (1,0)
          const double fA = wA * q.f01, fB = wB * q.f10;
          const double f1st = k_is_i ? fA : fB, f2nd = k_is_i ? fB : fA;
          fpair = w00 * q.f00 + f1st + f2nd + w11 * q.f11;
          if (EV) {
            const double eA = wA * q.u01, eB = wB * q.u10;
            const double e1st = k_is_i ? eA : eB, e2nd = k_is_i ? eB : eA;
            evdwl = w00 * q.u00 + e1st + e2nd + w11 * q.u11;
          }
          uf -= lm * (q.u11 - q.u01) + (1. - lm) * (q.u10 - q.u00);
        } else {
          // Bethe closure in the reference's orientation (UCG/pair_table_ucg_bethe.cpp:544-604)
          const double cu01 = k_is_i ? q.u01 : q.u10, cu10 = k_is_i ? q.u10 : q.u01;
          const double cf01 = k_is_i ? q.f01 : q.f10, cf10 = k_is_i ? q.f10 : q.f01;
          double pm_as_i1, pm_as_i0, pm_as_j1, pm_as_j0;
          {
            // the usual case first (every bead has been through fix ucgstate: ucgp is set), the first-call rules
            // (:179-205, :227-253) in ONE rarely taken branch behind it
            const double upm = up_cur;
            pm_as_i1 = lm;
            pm_as_i0 = 1.0 - lm;
            pm_as_j1 = upm;
            pm_as_j0 = 1.0 - upm;
            if (!firstp) upm_sign_or |= __double2hiint(upm);
            if (firstp && upm < -0.999) {
              if (P.prior_flag == 0) {
                pm_as_i0 = P.prior_type[tm * 2 + 0];
                pm_as_i1 = P.prior_type[tm * 2 + 1];
                pm_as_j0 = P.prior_type[tk * 2 + 0];  // row owner's type, as shipped
                pm_as_j1 = P.prior_type[tk * 2 + 1];
              } else {
                pm_as_j0 = 1.0 - lm;
                pm_as_j1 = lm;
              }
            }
          }
          double kj0 = pk_as_j0, kj1 = pk_as_j1;
          if (k_first_chempot) {  // k is "j" on its first call with the chemical-potential prior
            kj0 = P.prior_type[tm * 2 + 0];
            kj1 = P.prior_type[tm * 2 + 1];
          }
          const double pi0 = k_is_i ? pk_as_i0 : pm_as_i0, pi1 = k_is_i ? pk_as_i1 : pm_as_i1;
          const double pj0 = k_is_i ? pm_as_j0 : kj0, pj1 = k_is_i ? pm_as_j1 : kj1;

          double Jij = q.u11 + q.u00 - cu01 - cu10;
          if ((FAST ? div_kT(Jij, kT, rkT, kTp2) : Jij / kT) < -709.0) Jij = -700.0 * kT;
          const double mJkT = FAST ? div_kT(-Jij, kT, rkT, kTp2) : -Jij / kT;
          double bij, aij;
          ucg_exp_expm1(mJkT, &bij, &aij);  // = ucg_exp, ucg_expm1 bit for bit: one argument reduction, no k branches
          const double Qij = (pi1 + pj1) * aij + 1.;
          double Dij = Qij * Qij - 4. * aij * bij * pi1 * pj1;
          Dij = (Dij > 0.0) ? Dij : 0.0;
          double pij11 = pi1 * pj1;
          if (P.method_flag == 1) {
            // the closure's two quotient forms (:566-575) share the square root and ONE division: numerator and
            // denominator are selected, not the branch (same operations on the selected operands: same bits)
            // the root: the bare iteration where the hardware form's operand scaling and special cases are the identity
            // (ucg_sqrt_core) -- chosen per wavefront, so that a wavefront runs one of the two sequences
            double sD;
            if (__builtin_amdgcn_ballot_w64(!(Dij >= 0x1p-700 && Dij < 0x1p+700)) != 0ull) sD = sqrt(Dij);
            else sD = ucg_sqrt_core(Dij);
            const bool neg = Qij < 0.0;
            const double num = neg ? (Qij - sD) : (2. * bij * pi1 * pj1);
            const double den = neg ? (2. * aij) : (Qij + sD);
            const double quo = num / den;
            pij11 = (fabs(aij) < 1.0e-6) ? pij11 : quo;
          }
          const double pij00 = 1. + pij11 - pi1 - pj1;
          const double pij10 = pi1 - pij11;
          const double pij01 = pj1 - pij11;
          if (pseudo_flag == 1) {
            // full-SCE scores exactly as shipped (:583-601)
            if (FAST && rk_ok) {
              // same quotients; the row bead's priors are pi when it is "i" and pj when it is "j"
              const double d0 = k_is_i ? pi0 : pj0, d1 = k_is_i ? pi1 : pj1;
              const double r0 = k_is_i ? rk_i0 : rk_j0, r1 = k_is_i ? rk_i1 : rk_j1;
              const double n01 = k_is_i ? pij01 : pij10, n10 = k_is_i ? pij10 : pij01;
              const double qa = div_by_const(pij00, d0, r0), qb = div_by_const(n01, d0, r0);
              const double qc = div_by_const(n10, d1, r1), qd = div_by_const(pij11, d1, r1);
              if (k_is_i) {
                s0 -= div_kT(qa * q.u00 + qc * cu01, kT, rkT, kTp2);
                s1 -= div_kT(qb * cu10 + qd * q.u11, kT, rkT, kTp2);
              } else {
                s0 -= div_kT(qa * q.u00 + qb * cu01, kT, rkT, kTp2);
                s1 -= div_kT(qc * cu10 + qd * q.u11, kT, rkT, kTp2);
              }
            } else if (k_is_i) {
              const double pj0i0 = pij00 / pi0, pj0i1 = pij01 / pi0, pj1i0 = pij10 / pi1, pj1i1 = pij11 / pi1;
              s0 -= (pj0i0 * q.u00 + pj1i0 * cu01) / kT;
              s1 -= (pj0i1 * cu10 + pj1i1 * q.u11) / kT;
            } else {
              const double pi0j0 = pij00 / pj0, pi0j1 = pij10 / pj0, pi1j0 = pij01 / pj1, pi1j1 = pij11 / pj1;
              s0 -= (pi0j0 * q.u00 + pi0j1 * cu01) / kT;
              s1 -= (pi1j0 * cu10 + pi1j1 * q.u11) / kT;
            }
          }
          fpair = pij00 * q.f00 + pij01 * cf01 + pij10 * cf10 + pij11 * q.f11;
          if (EV) evdwl = pij00 * q.u00 + pij01 * cu01 + pij10 * cu10 + pij11 * q.u11;
        }
        fx += dx * fpair;
        fy += dy * fpair;
        fz += dz * fpair;
        if (EV) {
          ev[0] += 0.5 * evdwl;
          ev[1] += 0.5 * (dx * dx * fpair);
          ev[2] += 0.5 * (dy * dy * fpair);
          ev[3] += 0.5 * (dz * dz * fpair);
          ev[4] += 0.5 * (dx * dy * fpair);
          ev[5] += 0.5 * (dx * dz * fpair);
          ev[6] += 0.5 * (dy * dz * fpair);
        }
      }
      ent = ent_n;
      ent_n = ent_nn;
      pm = pm_n;
      mm = mm_n;
      up_cur = up_n;
    }
    if (STYLE == 1 && upm_sign_or < 0) err |= 16;
  }
  if (active) {
    if (SLOTS > 1) {
      // fixed tree over the bead's lanes: s[l] += s[l + off], off = SLOTS/2 ... 1
#pragma unroll
      for (int off = SLOTS / 2; off > 0; off >>= 1) {
        fx += __shfl_down(fx, off, SLOTS);
        fy += __shfl_down(fy, off, SLOTS);
        fz += __shfl_down(fz, off, SLOTS);
        if (STYLE == 0) uf += __shfl_down(uf, off, SLOTS);
        s0 += __shfl_down(s0, off, SLOTS);
        s1 += __shfl_down(s1, off, SLOTS);
      }
    }
    if (slot == 0) {
      const PostDev &Q = Lst.post;
      if (!EV && Q.enabled) {
        // Epilogue (pair_epilogue, ucg_pair_dev.h): [wall/hard bias ->] ucgld/langevin -> ucgstate -> final_integrate ->
        // the next step's initial_integrate on the sums this lane holds; f, ucgforce and the scores never reach HBM
        pair_epilogue<STYLE>(A, Q, k, mk, pk, fx, fy, fz, uf, s0, s1);
      } else {
        if (STYLE == 0) {
          A.frc4[k] = make_double4(fx, fy, fz, uf);
        } else {
          // table_ucg_bethe never touches ucgforce: it stays at its cleared value
          A.frc4[k] = make_double4(fx, fy, fz, 0.0);
        }
        A.scores[k] = make_double2(s0, s1);
        A.num_ucgstates[k] = 2;
      }
    }
  }
  if (FAST) err |= range_flags(s_par[0], P.tlm1, rtrack);
  if (err) atomicOr(errflag, err);
  if (EV) block_sum_store<8>(ev, s_red, evpart);
}

}  // namespace
}  // namespace ucg
