// ucg_pair_vrow.hip -- the neighbour loop of table_ucgld / table_ucg_bethe with every pair of two beads of one
// workgroup evaluated ONCE, on balanced "virtual rows" (gfx950).
//
// What is computed: Scenario 4 of PairTable_UCGLD::compute (UCG/pair_table_ucgld.cpp:424-533) and of
// PairTable_UCG_Bethe::compute (UCG/pair_table_ucg_bethe.cpp:457-620) -- the same per-pair arithmetic as
// k_pair_gather (ucg_pair.hip), in the reference's (i, j) orientation.
//
// How the work is laid out.  k_pair_gather gives every owned bead one lane that walks the bead's row of a FULL list:
// every pair is evaluated twice, and a wavefront runs as long as its longest row.  Here a workgroup owns a block of
// VR_BEADS = 512 beads (tables + the block's beads + their accumulators fill the 160 KB of LDS) and its 1024 lanes
// walk 1024 VIRTUAL rows of equal length:
//   * the block's entries are taken from its beads' rows in bead order; an entry whose neighbour is a bead of the
//     same block is kept only in the row of the lower-indexed bead (the pair is then evaluated once and what the
//     reference's half-list sweep adds to the partner -- UCG/pair_table_ucgld.cpp:500-502, 514-517, 523-530 -- is
//     formed by the same lane);
//   * the kept entries of a bead are padded to a multiple of VR_ALIGN, the padded lists of the block's beads are
//     concatenated and cut into 1024 pieces of equal length: piece l is virtual row l, stored transposed
//     ([slot][lane], 64 consecutive ints per wavefront and slot).  The first entry of a bead carries VR_FLAG.
//     Entries inside the force cutoff at build time (pass A) and skin entries (pass B) form two such lists, so that
//     the lanes of a wavefront run the heavy body together;
//   * a lane keeps the running sums of the bead it is working on in registers and adds them to that bead's LDS
//     accumulators when its row moves on to the next bead (possible at slots that are multiples of VR_ALIGN only:
//     one uniform test per VR_ALIGN slots) and at the end of the row.
// Sums: every term is added as a 64-bit INTEGER, its value rounded to nearest-even at 2^-40 (bits(v + 6144.0) -
// bits(6144.0)); integer addition is associative and commutative, so a bead's sums do not depend on the order or the
// lane in which its terms arrive: bit-reproducible, and independent of how the rows were cut.  |term| < 2048, else
// error bit 4.  The oracle's statement of this order: orc_pair_set_sum_fixed (oracle/orc_compute.c).
//
// Compiled with -ffp-contract=off like ucg_pair.hip: every product and sum of the pair arithmetic rounds where the
// reference's scalar x86-64 code rounds.
#include "ucg_pair_dev.h"

namespace ucg {

namespace {

constexpr int VR_BEADS = 512;
constexpr int VR_LANES = PAIR_BLOCK;
constexpr int VR_ALIGN = 4;
constexpr int VR_FLAG = 1 << 30;
constexpr int VR_DUMMY = (int) 0x80000000u;  // bit 31: a padding slot (its index field holds the block's first bead)
constexpr double VR_MAGIC = 24576.0;                // 1.5 * 2^14: ulp = 2^-38
constexpr double VR_UNIT = 3.637978807091713e-12;   // 2^-38
constexpr long long VR_MAGIC_BITS = 0x40D8000000000000ll;

// ------------------------------------------------------------------------------------------------ building the rows

__device__ __forceinline__ int roundup_align(int n) { return ((n + VR_ALIGN - 1) / VR_ALIGN) * VR_ALIGN; }

// is the entry (k, m) of bead k's row kept in the block's lists?  (an own-block pair lives in the lower bead's row)
__device__ __forceinline__ bool vrow_keeps(const int k, const int m, const int k0, const int nlocal)
{
  const bool own = m >= k0 && m < k0 + VR_BEADS && m < nlocal;
  return !own || m > k;
}

// pass 1: per bead the padded numbers of kept entries inside / outside the cutoff, their exclusive prefix sums within
// the block, per block the row lengths; pass 2 (FILL): the entries go to their slots
template <bool FILL>
__global__ __launch_bounds__(VR_BEADS) void k_vrow_build(const int nlocal, const int na1, const double4 *pos4, const int *meta,
                                                        const double *cutsq, const int *numneigh, const int *neigh,
                                                        const int pitch, int2 *beadoff, int4 *blockinfo, int *maxlen,
                                                        int *entA, int *entB, const int vpitch, int2 *lanemeta)
{
  __shared__ int s_scanA[VR_BEADS], s_scanB[VR_BEADS];
  const int blk = blockIdx.x, i = threadIdx.x, k0 = blk * VR_BEADS, k = k0 + i;
  int nA = 0, nB = 0, offA = 0, offB = 0, TA = 0, TB = 0;
  if (FILL) {
    if (k < nlocal) {
      offA = beadoff[k].x;
      offB = beadoff[k].y;
    }
    const int4 bi = blockinfo[blk];
    TA = bi.x;
    TB = bi.y;
  }
  if (k < nlocal) {
    const double4 pk = pos4[k];
    const int tk = UCG_META_TYPE(meta[k]);
    const int n = numneigh[k];
    for (int e = 0; e < n; e++) {
      const int ent = neigh[(size_t) e * pitch + k];
      const int m = ent & 0x1FFFFFFF;
      if (!vrow_keeps(k, m, k0, nlocal)) continue;
      const double4 pm = pos4[m];
      const int tm = UCG_META_TYPE(meta[m]);
      const double dx = pk.x - pm.x, dy = pk.y - pm.y, dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const bool in = rsq < cutsq[tk * na1 + tm];
      if (FILL) {
        const int cnt = in ? nA : nB;
        const int q = (in ? offA : offB) + cnt;
        const int T = in ? TA : TB;
        const int lane = q / T, slot = q - lane * T;
        (in ? entA : entB)[(size_t) slot * vpitch + blk * VR_LANES + lane] = (ent & 0x3FFFFFFF & ~VR_FLAG) | (cnt == 0 ? VR_FLAG : 0);
      }
      if (in) nA++;
      else nB++;
    }
    if (FILL) {
      // padding slots; a bead without entries still announces itself (the rows rely on consecutive beads)
      for (int pass = 0; pass < 2; pass++) {
        const int nn = pass ? nB : nA, T = pass ? TB : TA, off = pass ? offB : offA;
        const int pn = roundup_align(nn > 0 ? nn : 1);
        for (int c = nn; c < pn; c++) {
          const int q = off + c;
          const int lane = q / T, slot = q - lane * T;
          (pass ? entB : entA)[(size_t) slot * vpitch + blk * VR_LANES + lane] = VR_DUMMY | k0 | (c == 0 ? VR_FLAG : 0);
        }
      }
    }
  }
  const int pA = k < nlocal ? roundup_align(nA > 0 ? nA : 1) : 0, pB = k < nlocal ? roundup_align(nB > 0 ? nB : 1) : 0;
  // inclusive scan over the block (Hillis-Steele in LDS; 512 values)
  s_scanA[i] = pA;
  s_scanB[i] = pB;
  __syncthreads();
  for (int d = 1; d < VR_BEADS; d <<= 1) {
    const int a = i >= d ? s_scanA[i - d] : 0, b = i >= d ? s_scanB[i - d] : 0;
    __syncthreads();
    s_scanA[i] += a;
    s_scanB[i] += b;
    __syncthreads();
  }
  const int PA = s_scanA[VR_BEADS - 1], PB = s_scanB[VR_BEADS - 1];
  if (!FILL) {
    if (k < nlocal) beadoff[k] = make_int2(s_scanA[i] - pA, s_scanB[i] - pB);
    if (i == 0) {
      const int ta = roundup_align((PA + VR_LANES - 1) / VR_LANES), tb = roundup_align((PB + VR_LANES - 1) / VR_LANES);
      blockinfo[blk] = make_int4(ta > 0 ? ta : VR_ALIGN, tb > 0 ? tb : VR_ALIGN, PA, PB);
      atomicMax(&maxlen[0], ta);
      atomicMax(&maxlen[1], tb);
    }
  } else {
    // per virtual row: the bead its first slot belongs to and its length, both passes: lanes 2 i and 2 i + 1
    for (int l = 2 * i; l < 2 * i + 2; l++) {
      int2 mt[2];
      for (int pass = 0; pass < 2; pass++) {
        const int *sc = pass ? s_scanB : s_scanA;  // inclusive sums: bead b covers [sc[b] - p_b, sc[b])
        const int T = pass ? TB : TA, P = pass ? PB : PA;
        const int q0 = l * T;
        int cnt = P - q0;
        cnt = cnt < 0 ? 0 : (cnt > T ? T : cnt);
        int lo = 0, hi = VR_BEADS - 1;  // first bead with sc[b] > q0
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (sc[mid] > q0) hi = mid;
          else lo = mid + 1;
        }
        mt[pass] = make_int2(lo, cnt);
      }
      lanemeta[(size_t) (blk * VR_LANES + l) * 2] = mt[0];
      lanemeta[(size_t) (blk * VR_LANES + l) * 2 + 1] = mt[1];
    }
  }
}

// ------------------------------------------------------------------------------------------------------ the sweep

__device__ __forceinline__ long long vr_image(const double v) { return __double_as_longlong(v + VR_MAGIC); }
__device__ __forceinline__ double vr_decode(const unsigned long long s) { return (double) (long long) s * VR_UNIT; }
// (the range of the terms is guaranteed by PairDev::fixed_rsq_safe and the lambda test of load_bead, not per term)
__device__ __forceinline__ void vr_partner_add(unsigned long long *acc, const double v, int &err)
{
  atomicAdd(acc, (unsigned long long) (vr_image(v) - VR_MAGIC_BITS));
}

struct VrowDev {
  const int *entA, *entB;
  const int2 *lanemeta;
  int vpitch;
};

template <int STYLE, int TS, bool EV, bool ONETYPE, int SCE>
__global__ __launch_bounds__(PAIR_BLOCK) void k_pair_vrow(const PairDev P, const AtomsDev A, const ListDev Lst, const VrowDev V,
                                                         double *evpart, int *errflag)
{
  extern __shared__ double4 s_tab[];
  __shared__ double s_red[(PAIR_BLOCK / 64) * 8];
  __shared__ double4 s_par[UCG_MAX_TABLES];
  __shared__ int s_pairtab[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1) * 4];
  __shared__ double s_cutsq[(UCG_MAX_ACTUAL + 1) * (UCG_MAX_ACTUAL + 1)];
  constexpr int NACC = STYLE == 0 ? 6 : 5;  // fx fy fz [ucgforce] s0 s1

  const int blk = Lst.blockflag ? (int) blockIdx.x : xcd_chunk(blockIdx.x, gridDim.x);
  if (Lst.blockflag && Lst.blockflag[blk] != Lst.blockwant) return;  // whole workgroup
  const int ntabent = (P.tablength * P.fast_stride + 1) / 2;
  double4 *s_ownpos = s_tab + ntabent;
  double *s_ownp = reinterpret_cast<double *>(s_ownpos + VR_BEADS);                // Bethe: ucgp of the block's beads
  int *s_ownmeta = reinterpret_cast<int *>(s_ownp + (STYLE == 1 ? VR_BEADS : 0));
  unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_ownmeta + VR_BEADS);  // [field][bead]
  const int k0 = blk * VR_BEADS;
  const int nlocal = A.nlocal;
  const unsigned nown = (unsigned) min(VR_BEADS, nlocal - k0);
  for (int t = threadIdx.x; t < NACC * VR_BEADS; t += blockDim.x) s_acc[t] = 0ull;
  for (int t = threadIdx.x; t < VR_BEADS; t += blockDim.x) {
    if (k0 + t < nlocal) {
      s_ownpos[t] = A.pos4[k0 + t];
      s_ownmeta[t] = A.meta[k0 + t];
      if (STYLE == 1) s_ownp[t] = A.ucgp[k0 + t];
    }
  }
  {
    const int na1sq = (P.n_actual + 1) * (P.n_actual + 1);
    for (int t = threadIdx.x; t < P.ntab; t += blockDim.x) s_par[t] = P.tabpar[t];
    for (int t = threadIdx.x; t < na1sq * 4; t += blockDim.x) s_pairtab[t] = P.pairtab[t];
    for (int t = threadIdx.x; t < na1sq; t += blockDim.x) s_cutsq[t] = P.cutsq[t];
    for (int t = threadIdx.x; t < ntabent; t += blockDim.x) s_tab[t] = P.tab_fast[t];
    __syncthreads();
  }

  const int na1 = P.n_actual + 1;
  const double kT = P.kT, rkT = P.rkT;
  const int kTp2 = P.kT_pow2;
  const int pseudo_flag = SCE < 0 ? P.pseudo_flag : SCE;
  const bool onetype = ONETYPE || (P.n_actual == 1);
  const double cut11 = P.cutsq[na1 + 1];
  const int pt11_0 = P.pairtab[(na1 + 1) * 4 + 0], pt11_1 = P.pairtab[(na1 + 1) * 4 + 1];
  const int pt11_2 = P.pairtab[(na1 + 1) * 4 + 2], pt11_3 = P.pairtab[(na1 + 1) * 4 + 3];
  double ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int err = 0;
  RangeTrack rtrack = range_track_init();
  const size_t vpitch = (size_t) V.vpitch;
  const int vlane = blk * VR_LANES + (int) threadIdx.x;

  // one pass over a list of virtual rows (A: inside the cutoff at build time, B: the skin)
  auto sweep = [&](const int *ve, const int2 mt) {
    const int cnt = mt.y;
    if (cnt <= 0) return;
    int ib = mt.x;  // the bead (block-local) whose entries the row is at
    // ---- state of the current bead
    double4 pk;
    int mk, tk, sk;
    double lk;
    double pk_as_i1 = 0.0, pk_as_j1 = 0.0, pk_as_i0 = 0.0, pk_as_j0 = 0.0;
    bool k_first_chempot = false;
    auto load_bead = [&]() {
      pk = s_ownpos[ib];
      mk = s_ownmeta[ib];
      tk = UCG_META_TYPE(mk);
      sk = UCG_META_STATE(mk);
      lk = pk.w;
      if (!(fabs(lk - 0.5) <= 2.3)) err |= 4;  // the bound on the terms (fixed_rsq_safe) assumes mixing weights up to 2.8^2
      if (STYLE == 1) {  // priors of the bead for the Bethe closure (as in k_pair_gather)
        const double upk = s_ownp[ib];
        const bool first = upk < -0.999;
        k_first_chempot = false;
        if (first && P.prior_flag == 0) {
          pk_as_i0 = P.prior_type[tk * 2 + 0];
          pk_as_i1 = P.prior_type[tk * 2 + 1];
        } else {
          pk_as_i1 = lk;
          pk_as_i0 = 1.0 - lk;
        }
        if (first) {
          if (P.prior_flag == 0) {
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
    };
    load_bead();
    long long ax = 0, ay = 0, az = 0, au = 0, a0 = 0, a1 = 0;  // integer images of the running sums (raw: + MAGIC bits per term)
    int nterm = 0;
    auto flush = [&]() {
      if (nterm) {
        // n x bits(MAGIC): the low word of bits(MAGIC) is zero, so one 32-bit product (modulo 2^64 like the sums)
        const long long corr = (long long) ((unsigned long long) ((unsigned) nterm * 0x40D80000u) << 32);
        unsigned long long *ap = s_acc + ib;
        atomicAdd(ap, (unsigned long long) (ax - corr));
        atomicAdd(ap + VR_BEADS, (unsigned long long) (ay - corr));
        atomicAdd(ap + 2 * VR_BEADS, (unsigned long long) (az - corr));
        if (STYLE == 0) atomicAdd(ap + 3 * VR_BEADS, (unsigned long long) (au - corr));
        atomicAdd(ap + (NACC - 2) * VR_BEADS, (unsigned long long) (a0 - corr));
        atomicAdd(ap + (NACC - 1) * VR_BEADS, (unsigned long long) (a1 - corr));
        ax = ay = az = au = a0 = a1 = 0;
        nterm = 0;
      }
    };

    const int *rp = ve + vlane;
    int ent = rp[0];
    int ent_n = (1 < cnt) ? rp[vpitch] : ent;
    auto gather = [&](const int e, double4 &pm, int &mm) {
      const int m = e & 0x1FFFFFFF;  // (a padding slot gathers the block's first bead: never used)
      const unsigned ml = (unsigned) (m - k0);
      if (ml < nown) {
        pm = s_ownpos[ml];
        mm = s_ownmeta[ml];
      } else {
        pm = A.pos4[m];
        mm = A.meta[m];
      }
    };
    double4 pm;
    int mm;
    gather(ent, pm, mm);
    rp += vpitch;
    for (int t = 0; t < cnt; t++) {
      rp += vpitch;
      const int ent_nn = (t + 2 < cnt) ? rp[0] : ent_n;
      double4 pm_n;
      int mm_n;
      gather(ent_n, pm_n, mm_n);

      if ((t % VR_ALIGN) == 0 && t > 0) {  // uniform test; the row may move on to its next bead here
        if (ent & VR_FLAG) {
          flush();
          ib++;
          load_bead();
        }
      }
      const int m = ent & 0x1FFFFFFF;
      const bool k_is_i = (ent >> 29) & 1;
      const int tm = UCG_META_TYPE(mm);
      const int sm = UCG_META_STATE(mm);
      const double lm = pm.w;
      const double dx = pk.x - pm.x;
      const double dy = pk.y - pm.y;
      const double dz = pk.z - pm.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const double cutv = onetype ? cut11 : s_cutsq[tk * na1 + tm];
      if (rsq < cutv && ent >= 0) {
        int pt[4];
        if (onetype) {
          pt[0] = pt11_0; pt[1] = pt11_1; pt[2] = pt11_2; pt[3] = pt11_3;
        } else {
          const int *ps = s_pairtab + (tk * na1 + tm) * 4;
          pt[0] = ps[0]; pt[1] = ps[1]; pt[2] = ps[2]; pt[3] = ps[3];
        }
        Quad q;
        eval_quad<TS, true>(s_tab, s_par, pt, P.tablength, P.tlm1, P.fast_stride, rsq, 1.0, q, err, rtrack);
        const unsigned ml = (unsigned) (m - k0);
        const bool partner = ml < nown;  // the pair is evaluated here only: the partner's terms too
        unsigned long long *ap = s_acc + (partner ? ml : 0u);
        double evdwl = 0.0, fpair;
        double t0 = 0.0, t1 = 0.0;  // this bead's score terms
        if (STYLE == 0 || pseudo_flag == 0) {
          // pseudo-likelihood scores (:492-502): S[k][a] -= u[a][state of the neighbour] / kT
          t0 = -div_kT(sm ? q.u01 : q.u00, kT, rkT, kTp2);
          t1 = -div_kT(sm ? q.u11 : q.u10, kT, rkT, kTp2);
          if (partner) {
            // ... and S[m][b] -= u[state of this bead][b] / kT (:500-502)
            vr_partner_add(ap + (NACC - 2) * VR_BEADS, -div_kT(sk ? q.u10 : q.u00, kT, rkT, kTp2), err);
            vr_partner_add(ap + (NACC - 1) * VR_BEADS, -div_kT(sk ? q.u11 : q.u01, kT, rkT, kTp2), err);
          }
        }
        if (STYLE == 0) {
          // lambda-bilinear mix (:507-517) in the reference's orientation
          const double w00 = (1. - lk) * (1. - lm);
          const double w11 = lk * lm;
          const double wA = (1. - lk) * lm;
          const double wB = (1. - lm) * lk;
          const double fA = wA * q.f01, fB = wB * q.f10;
          const double f1st = k_is_i ? fA : fB, f2nd = k_is_i ? fB : fA;
          fpair = w00 * q.f00 + f1st + f2nd + w11 * q.f11;
          if (EV) {
            const double eA = wA * q.u01, eB = wB * q.u10;
            const double e1st = k_is_i ? eA : eB, e2nd = k_is_i ? eB : eA;
            evdwl = w00 * q.u00 + e1st + e2nd + w11 * q.u11;
          }
          const double tu = -(lm * (q.u11 - q.u01) + (1. - lm) * (q.u10 - q.u00));
          au += vr_image(tu);
          if (partner) vr_partner_add(ap + 3 * VR_BEADS, -(lk * (q.u11 - q.u10) + (1. - lk) * (q.u01 - q.u00)), err);
        } else {
          // Bethe closure in the reference's orientation (UCG/pair_table_ucg_bethe.cpp:544-604), as in k_pair_gather
          const double cu01 = k_is_i ? q.u01 : q.u10, cu10 = k_is_i ? q.u10 : q.u01;
          const double cf01 = k_is_i ? q.f01 : q.f10, cf10 = k_is_i ? q.f10 : q.f01;
          double pm_as_i1, pm_as_i0, pm_as_j1, pm_as_j0;
          {
            const double upm = partner ? s_ownp[ml] : A.ucgp[m];
            pm_as_i1 = lm;
            pm_as_i0 = 1.0 - lm;
            pm_as_j1 = upm;
            pm_as_j0 = 1.0 - upm;
            if (upm < -0.999) {
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
          if (k_first_chempot) {
            kj0 = P.prior_type[tm * 2 + 0];
            kj1 = P.prior_type[tm * 2 + 1];
          }
          const double pi0 = k_is_i ? pk_as_i0 : pm_as_i0, pi1 = k_is_i ? pk_as_i1 : pm_as_i1;
          const double pj0 = k_is_i ? pm_as_j0 : kj0, pj1 = k_is_i ? pm_as_j1 : kj1;

          double Jij = q.u11 + q.u00 - cu01 - cu10;
          if (div_kT(Jij, kT, rkT, kTp2) < -709.0) Jij = -700.0 * kT;
          const double mJkT = div_kT(-Jij, kT, rkT, kTp2);
          double bij, aij;
          ucg_exp_expm1(mJkT, &bij, &aij);
          const double Qij = (pi1 + pj1) * aij + 1.;
          double Dij = Qij * Qij - 4. * aij * bij * pi1 * pj1;
          Dij = (Dij > 0.0) ? Dij : 0.0;
          double pij11 = pi1 * pj1;
          if (P.method_flag == 1) {
            const double sD = sqrt(Dij);
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
            // full-SCE scores exactly as shipped (:583-601): the "i" bead's and the "j" bead's
            const double pj0i0 = pij00 / pi0, pj0i1 = pij01 / pi0, pj1i0 = pij10 / pi1, pj1i1 = pij11 / pi1;
            const double si0 = -((pj0i0 * q.u00 + pj1i0 * cu01) / kT), si1 = -((pj0i1 * cu10 + pj1i1 * q.u11) / kT);
            const double pi0j0 = pij00 / pj0, pi0j1 = pij10 / pj0, pi1j0 = pij01 / pj1, pi1j1 = pij11 / pj1;
            const double sj0 = -((pi0j0 * q.u00 + pi0j1 * cu01) / kT), sj1 = -((pi1j0 * cu10 + pi1j1 * q.u11) / kT);
            t0 = k_is_i ? si0 : sj0;
            t1 = k_is_i ? si1 : sj1;
            if (partner) {
              vr_partner_add(ap + (NACC - 2) * VR_BEADS, k_is_i ? sj0 : si0, err);
              vr_partner_add(ap + (NACC - 1) * VR_BEADS, k_is_i ? sj1 : si1, err);
            }
          }
          fpair = pij00 * q.f00 + pij01 * cf01 + pij10 * cf10 + pij11 * q.f11;
          if (EV) evdwl = pij00 * q.u00 + pij01 * cu01 + pij10 * cu10 + pij11 * q.u11;
        }
        const double tx = dx * fpair, ty = dy * fpair, tz = dz * fpair;
        const long long ix = vr_image(tx), iy = vr_image(ty), iz = vr_image(tz);
        ax += ix;
        ay += iy;
        az += iz;
        a0 += vr_image(t0);
        a1 += vr_image(t1);
        nterm++;
        if (partner) {  // the partner's force terms are the negatives: image(-v) = -image(v) (round to nearest even is symmetric)
          atomicAdd(ap, (unsigned long long) (VR_MAGIC_BITS - ix));
          atomicAdd(ap + VR_BEADS, (unsigned long long) (VR_MAGIC_BITS - iy));
          atomicAdd(ap + 2 * VR_BEADS, (unsigned long long) (VR_MAGIC_BITS - iz));
        }
        if (EV) {
          const double h = partner ? 1.0 : 0.5;  // the pair is seen once here, twice (two halves) otherwise
          ev[0] += h * evdwl;
          ev[1] += h * (dx * dx * fpair);
          ev[2] += h * (dy * dy * fpair);
          ev[3] += h * (dz * dz * fpair);
          ev[4] += h * (dx * dy * fpair);
          ev[5] += h * (dx * dz * fpair);
          ev[6] += h * (dy * dz * fpair);
        }
      }
      ent = ent_n;
      ent_n = ent_nn;
      pm = pm_n;
      mm = mm_n;
    }
    flush();
  };

  const int2 mtA = V.lanemeta[(size_t) vlane * 2], mtB = V.lanemeta[(size_t) vlane * 2 + 1];
  sweep(V.entA, mtA);
  sweep(V.entB, mtB);
  __syncthreads();  // every lane of the workgroup has made its adds

  const int k = k0 + (int) threadIdx.x;
  if (threadIdx.x < VR_BEADS && k < nlocal) {
    const unsigned long long *ap = s_acc + threadIdx.x;
    const double4 pk = s_ownpos[threadIdx.x];
    const int mk = s_ownmeta[threadIdx.x];
    const int tk = UCG_META_TYPE(mk);
    const double mu0 = P.mu[tk * 2 + 0], mu1 = P.mu[tk * 2 + 1];
    // the prologue values (:170-180 / bethe :155-162) plus the exact integer sums
    double fx = vr_decode(ap[0]), fy = vr_decode(ap[VR_BEADS]), fz = vr_decode(ap[2 * VR_BEADS]);
    double uf = 0.0, s0, s1;
    if (STYLE == 0) {
      const double mui = mu1 - mu0;
      uf = -mui + vr_decode(ap[3 * VR_BEADS]);
      s0 = vr_decode(ap[4 * VR_BEADS]);
      s1 = -(mui / kT) + vr_decode(ap[5 * VR_BEADS]);
    } else {
      s0 = -mu0 / kT + vr_decode(ap[3 * VR_BEADS]);
      s1 = -mu1 / kT + vr_decode(ap[4 * VR_BEADS]);
    }
    const PostDev &Q = Lst.post;
    if (!EV && Q.enabled) {
      pair_epilogue<STYLE>(A, Q, k, mk, pk, fx, fy, fz, uf, s0, s1);
    } else {
      A.frc4[k] = make_double4(fx, fy, fz, STYLE == 0 ? uf : 0.0);
      A.scores[k] = make_double2(s0, s1);
      A.num_ucgstates[k] = 2;
    }
  }
  err |= range_flags(s_par[0], P.tlm1, rtrack);
  if (rtrack.rsq_min < P.fixed_rsq_safe) err |= 4;  // a pair so close that a term could leave the accumulators' range
  if (err) atomicOr(errflag, err);
  if (EV) block_sum_store<8>(ev, s_red, evpart);
}

template <int STYLE, int TS>
hipError_t launch_vrow_ts(const PairDev &P, const AtomsDev &A, const ListDev &L, const VrowDev &V, bool ev, double *evpart,
                          int *errflag, hipStream_t st, int nblocks, size_t lds)
{
#define UCG_VLAUNCH(EVF, ONE, SC)                                                                          \
  do {                                                                                                     \
    auto kern = k_pair_vrow<STYLE, TS, EVF, ONE, SC>;                                                      \
    hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(PAIR_BLOCK), lds, st, P, A, L, V, evpart, errflag);       \
  } while (0)
  const bool one = P.n_actual == 1;
  if (STYLE == 0) {
    if (ev) {
      if (one) UCG_VLAUNCH(true, true, -1);
      else UCG_VLAUNCH(true, false, -1);
    } else {
      if (one) UCG_VLAUNCH(false, true, -1);
      else UCG_VLAUNCH(false, false, -1);
    }
  } else {
    if (ev) {
      UCG_VLAUNCH(true, false, -1);
    } else if (one) {
      if (P.pseudo_flag) UCG_VLAUNCH(false, true, 1);
      else UCG_VLAUNCH(false, true, 0);
    } else {
      UCG_VLAUNCH(false, false, -1);
    }
  }
#undef UCG_VLAUNCH
  return hipGetLastError();
}

}  // namespace

int vrow_blocks(int nlocal) { return (nlocal + VR_BEADS - 1) / VR_BEADS; }
int vrow_beads() { return VR_BEADS; }

size_t vrow_lds_bytes(const PairDev &P)
{
  const size_t tab = ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4);
  const int nacc = P.style == 0 ? 6 : 5;
  return tab + (size_t) VR_BEADS * (sizeof(double4) + sizeof(int) + (P.style == 1 ? sizeof(double) : 0) + nacc * sizeof(unsigned long long));
}

// counts (fill = false: beadoff / blockinfo / maxlen are written) or fills (fill = true) the virtual rows of every block
hipError_t launch_vrow_build(bool fill, const PairDev &P, const AtomsDev &A, const ListDev &L, int2 *beadoff, int4 *blockinfo,
                             int *maxlen, int *entA, int *entB, int vpitch, int2 *lanemeta, hipStream_t st)
{
  const int nb = vrow_blocks(A.nlocal);
  if (nb == 0) return hipSuccess;
  if (fill)
    hipLaunchKernelGGL(k_vrow_build<true>, dim3(nb), dim3(VR_BEADS), 0, st, A.nlocal, P.n_actual + 1, A.pos4, A.meta, P.cutsq,
                       L.numneigh, L.neigh, L.pitch, beadoff, blockinfo, maxlen, entA, entB, vpitch, lanemeta);
  else
    hipLaunchKernelGGL(k_vrow_build<false>, dim3(nb), dim3(VR_BEADS), 0, st, A.nlocal, P.n_actual + 1, A.pos4, A.meta, P.cutsq,
                       L.numneigh, L.neigh, L.pitch, beadoff, blockinfo, maxlen, entA, entB, vpitch, lanemeta);
  return hipGetLastError();
}

hipError_t launch_pair_vrow(const PairDev &P, const AtomsDev &A, const ListDev &L, const int *entA, const int *entB,
                            const int2 *lanemeta, int vpitch, bool ev, double *evpart, double *evout, int *errflag,
                            hipStream_t st)
{
  const int nblocks = vrow_blocks(A.nlocal);
  if (nblocks == 0) return hipSuccess;
  if (!P.fast || !P.tab_in_lds || P.tabstyle == 3) return hipErrorInvalidValue;
  const size_t lds = vrow_lds_bytes(P);
  if (lds + 4608 > 160 * 1024) return hipErrorInvalidValue;
  VrowDev V{entA, entB, lanemeta, vpitch};
  hipError_t e;
#define UCG_VTS(ST)                                                                                               \
  switch (P.tabstyle) {                                                                                           \
    case 0: e = launch_vrow_ts<ST, 0>(P, A, L, V, ev, evpart, errflag, st, nblocks, lds); break;                  \
    case 1: e = launch_vrow_ts<ST, 1>(P, A, L, V, ev, evpart, errflag, st, nblocks, lds); break;                  \
    default: e = launch_vrow_ts<ST, 2>(P, A, L, V, ev, evpart, errflag, st, nblocks, lds); break;                 \
  }
  if (P.style == 0) {
    UCG_VTS(0)
  } else {
    UCG_VTS(1)
  }
#undef UCG_VTS
  if (e != hipSuccess) return e;
  if (ev) {
    e = launch_ev_final(evpart, nblocks, evout, st);
  }
  return e;
}

}  // namespace ucg
