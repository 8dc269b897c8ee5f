// ucg_pair_vrow.hip -- the neighbour loop of table_ucgld / table_ucg_bethe with every pair of two beads of one
// workgroup evaluated ONCE, on balanced "virtual rows" (gfx950).
//
// What is computed: Scenario 4 of PairTable_UCGLD::compute (UCG/pair_table_ucgld.cpp:424-533) and of
// PairTable_UCG_Bethe::compute (UCG/pair_table_ucg_bethe.cpp:457-620) -- the same per-pair arithmetic as
// k_pair_gather (ucg_pair.hip), in the reference's (i, j) orientation.
//
// How the work is laid out.  k_pair_gather gives every owned bead one lane that walks the bead's row of a FULL list:
// every pair is evaluated twice, and a wavefront runs as long as its longest row.  Here a workgroup owns a block of
// VR_BEADS = 512 consecutive beads (tables + the block's beads + their accumulators fill the 160 KB of LDS) and its
// 1024 lanes walk 1024 VIRTUAL rows of equal length, made from the full rows at every re-neighbouring (k_vrow_build):
//   * an entry whose neighbour is a bead of the same block is kept only in the row of the lower-indexed bead: the pair
//     is then evaluated once, and what the reference's half-list sweep adds to the partner
//     (UCG/pair_table_ucgld.cpp:500-502, 514-517, 523-530) is formed by the same lane;
//   * the block's kept entries form five lists -- P: own-block pairs inside the force cutoff at build time, Q: the
//     other entries inside the cutoff, S1..S3: the thirds of the skin, nearest first -- so that the lanes of a
//     wavefront run the same code (P: the partner's terms always, Q: never) and run the heavy body together (a skin
//     entry that has drifted inside the cutoff is most likely one of S1: the other skin lists stay light);
//   * in each list a bead's entries are padded to a multiple of VR_ALIGN, the beads' padded lists are concatenated in
//     bead order and cut into 1024 pieces of equal length: piece l is virtual row l, stored transposed
//     ([slot][lane], 64 consecutive ints per wavefront and slot).  The first entry of a bead carries VR_FLAG;
//   * a lane keeps the running sums of the bead it is working on in registers and adds them to that bead's LDS
//     accumulators when its row moves on to the next bead (possible at slots that are multiples of VR_ALIGN only:
//     one uniform test per VR_ALIGN slots) and at the end of the row.
// Sums: fixed sums (ucg_pair_dev.h) -- every term is added as a 64-bit INTEGER image; integer addition is associative
// and commutative, so a bead's sums do not depend on the order or the lane in which its terms arrive: bit-reproducible,
// and independent of how the rows were cut.  The oracle's statement: orc_pair_set_sum_fixed (oracle/orc_compute.c).
//
// Compiled with -ffp-contract=off like ucg_pair.hip: every product and sum of the pair arithmetic rounds where the
// reference's scalar x86-64 code rounds (the images' fma is written explicitly: its product is exact).
#include "ucg_pair_dev.h"

namespace ucg {

namespace {

constexpr int VR_BEADS = 512;
constexpr int VR_LANES = PAIR_BLOCK;
constexpr int VR_ALIGN = 4;
constexpr int VR_FLAG = 1 << 30;
constexpr int VR_DUMMY = (int) 0x80000000u;  // bit 31: a padding slot (its index field holds the block's first bead)
constexpr int VR_MAXROW = 128;               // entries of a full row the builder can classify (4 bits each in registers)
constexpr int VR_NLIST = 5;                  // P, Q, S1, S2, S3
constexpr int VR_STAGE_WORDS = 36 * 1024;    // LDS staging of one list of one block: 1024 lanes x (T + 1) ints

// ------------------------------------------------------------------------------------------------ building the rows

__device__ __forceinline__ int roundup_align(int n) { return ((n + VR_ALIGN - 1) / VR_ALIGN) * VR_ALIGN; }

// One workgroup per block of VR_BEADS beads, one thread per bead.
//  1. every thread walks its bead's full row (coalesced: the rows are stored [entry][bead]) and classifies each entry
//     (dropped, or one of the five lists: four bits, kept in registers; the distances need the neighbours' positions:
//     gathers through L2, which is what this kernel's time goes to -- 1.16 ms at 1 M beads);
//  2. the padded per-bead counts of each list are scanned over the block: the bead's offset in the list, the list's
//     length, the row length T = the length cut into 1024 pieces;
//  3. list by list: the threads walk their rows again (L2) and put the kept entries at their places in an LDS image of
//     the transposed list, then the whole workgroup writes that image out with coalesced stores, together with every
//     lane's first bead and entry count.
struct VrowLists {
  int *ent[VR_NLIST];
};

__global__ __launch_bounds__(VR_BEADS) void k_vrow_build(const int nlocal, const int na1, const double4 *pos4, const int *meta,
                                                        const double *cutsq, const double skin, const int *numneigh,
                                                        const int *neigh, const int pitch, const VrowLists lists, const int cap,
                                                        const int vpitch, int2 *lanemeta, int *errflag)
{
  extern __shared__ int s_stage[];
  __shared__ int s_scan[VR_NLIST][VR_BEADS];
  const int blk = blockIdx.x, i = threadIdx.x, k0 = blk * VR_BEADS, k = k0 + i;
  const bool live = k < nlocal;
  unsigned long long cls[VR_MAXROW / 16];  // 4 bits per entry: 0 dropped, 1 + list
  int cnt[VR_NLIST] = {0, 0, 0, 0, 0};
  int n = 0;
  if (live) {
    n = numneigh[k];
    if (n > VR_MAXROW) {
      atomicOr(errflag, 8);
      n = VR_MAXROW;
    }
    const double4 pk = pos4[k];
    const int tk = UCG_META_TYPE(meta[k]);
#pragma unroll
    for (int c = 0; c < VR_MAXROW / 16; c++) {
      unsigned long long w = 0ull;
      const int e1 = min(n, 16 * c + 16);
      for (int e = 16 * c; e < e1; e++) {
        const int m = neigh[(size_t) e * pitch + k] & 0x1FFFFFFF;
        const bool own = m >= k0 && m < k0 + VR_BEADS && m < nlocal;
        if (own && m <= k) continue;  // the pair lives in the lower bead's row
        const double4 pm = pos4[m];
        const int tm = UCG_META_TYPE(meta[m]);
        const double dx = pk.x - pm.x, dy = pk.y - pm.y, dz = pk.z - pm.z;
        const double rsq = dx * dx + dy * dy + dz * dz;
        const double csq = cutsq[tk * na1 + tm];
        int li;
        if (rsq < csq) {
          li = own ? 0 : 1;
        } else {  // thirds of the skin (a heuristic only: any assignment gives the same sums)
          const double rc = sqrt(csq), r1 = rc + skin * (1.0 / 3.0), r2 = rc + skin * (2.0 / 3.0);
          li = rsq < r1 * r1 ? 2 : (rsq < r2 * r2 ? 3 : 4);
        }
#pragma unroll
        for (int x = 0; x < VR_NLIST; x++) cnt[x] += li == x;
        w |= (unsigned long long) (li + 1) << (4 * (e - 16 * c));
      }
      cls[c] = w;
    }
  }
  // padded counts, inclusive scans (Hillis-Steele in LDS; 512 values per list)
  int pad[VR_NLIST];
#pragma unroll
  for (int x = 0; x < VR_NLIST; x++) {
    pad[x] = live ? roundup_align(cnt[x] > 0 ? cnt[x] : 1) : 0;  // a bead without entries still announces itself
    s_scan[x][i] = pad[x];
  }
  __syncthreads();
  for (int d = 1; d < VR_BEADS; d <<= 1) {
    int a[VR_NLIST];
#pragma unroll
    for (int x = 0; x < VR_NLIST; x++) a[x] = i >= d ? s_scan[x][i - d] : 0;
    __syncthreads();
#pragma unroll
    for (int x = 0; x < VR_NLIST; x++) s_scan[x][i] += a[x];
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < VR_NLIST; x++) {
    int *out = lists.ent[x];
    const int total = s_scan[x][VR_BEADS - 1];
    int T = roundup_align((total + VR_LANES - 1) / VR_LANES);
    T = T > 0 ? T : VR_ALIGN;
    const bool fits = T <= cap && (T + 1) * VR_LANES <= VR_STAGE_WORDS;
    if (!fits) {
      if (i == 0) atomicOr(errflag, 8);
      T = VR_ALIGN;  // (the lists of this block are not usable; the error is reported before any result is)
    }
    const int Tp = T + 1;  // LDS row pitch: odd multiples of the bank width apart, the transposed reads do not conflict
    if (live && fits) {
      int q = s_scan[x][i] - pad[x];
      int lane = q / T, slot = q - lane * T;
      int j = 0;
#pragma unroll
      for (int c = 0; c < VR_MAXROW / 16; c++) {
        const unsigned long long w = cls[c];
        const int e1 = min(n, 16 * c + 16);
        for (int e = 16 * c; e < e1; e++) {
          if ((int) ((w >> (4 * (e - 16 * c))) & 15ull) != x + 1) continue;
          const int ent = neigh[(size_t) e * pitch + k];
          s_stage[lane * Tp + slot] = (ent & 0x3FFFFFFF) | (j == 0 ? VR_FLAG : 0);
          j++;
          if (++slot == T) {
            slot = 0;
            lane++;
          }
        }
      }
      for (; j < pad[x]; j++) {
        s_stage[lane * Tp + slot] = VR_DUMMY | k0 | (j == 0 ? VR_FLAG : 0);
        if (++slot == T) {
          slot = 0;
          lane++;
        }
      }
    }
    __syncthreads();
    for (int l = i; l < VR_LANES; l += VR_BEADS) {
      int c = fits ? total - l * T : 0;
      c = c < 0 ? 0 : (c > T ? T : c);
      int *op = out + (size_t) blk * VR_LANES + l;
      for (int s = 0; s < c; s++) op[(size_t) s * vpitch] = s_stage[l * Tp + s];
      // the bead the lane's first slot belongs to: the first bead whose inclusive sum exceeds l * T
      const int q0 = l * T;
      int lo = 0, hi = VR_BEADS - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_scan[x][mid] > q0) hi = mid;
        else lo = mid + 1;
      }
      lanemeta[((size_t) blk * VR_LANES + l) * VR_NLIST + x] = make_int2(lo, c);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------ the sweep

struct VrowDev {
  const int *ent[VR_NLIST];
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
  const SumDev SU = sum_dev(P);
  double ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int err = 0;
  RangeTrack rtrack = range_track_init();
  const size_t vpitch = (size_t) V.vpitch;
  const int vlane = blk * VR_LANES + (int) threadIdx.x;

  // one pass over one of the block's lists.  MODE 0: list P (every entry's partner is a bead of the block: its data
  // come from LDS, its terms are formed), 1: list Q (no partner terms; gathers through L2), 2: the skin lists (either kind)
  auto sweep = [&](auto mode_c, const int *ve, const int2 mt) {
    constexpr int MODE = decltype(mode_c)::value;
    const int cnt = mt.y;
    if (cnt <= 0) return;
    int ib = mt.x;  // the bead (block-local) whose entries the row is at
    // ---- state of the current bead
    double4 pk;
    int mk, tk, sk;
    double lk;
    bool lam_ok;
    double pk_as_i1 = 0.0, pk_as_j1 = 0.0, pk_as_i0 = 0.0, pk_as_j0 = 0.0;
    bool k_first_chempot = false;
    auto load_bead = [&]() {
      pk = s_ownpos[ib];
      mk = s_ownmeta[ib];
      tk = UCG_META_TYPE(mk);
      sk = UCG_META_STATE(mk);
      lk = pk.w;
      lam_ok = fabs(lk - 0.5) <= 2.3;  // the bound behind sum_rsq_safe assumes mixing weights up to 2.8^2
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
    unsigned long long ax = 0, ay = 0, az = 0, au = 0, a0 = 0, a1 = 0;  // raw images of the running sums, modulo 2^64
    unsigned nterm = 0;
    auto flush = [&]() {
      if (nterm) {
        unsigned long long *ap = s_acc + ib;
        atomicAdd(ap, (unsigned long long) sum_finish(ax, nterm));
        atomicAdd(ap + VR_BEADS, (unsigned long long) sum_finish(ay, nterm));
        atomicAdd(ap + 2 * VR_BEADS, (unsigned long long) sum_finish(az, nterm));
        if (STYLE == 0) atomicAdd(ap + 3 * VR_BEADS, (unsigned long long) sum_finish(au, nterm));
        atomicAdd(ap + (NACC - 2) * VR_BEADS, (unsigned long long) sum_finish(a0, nterm));
        atomicAdd(ap + (NACC - 1) * VR_BEADS, (unsigned long long) sum_finish(a1, nterm));
        ax = ay = az = au = a0 = a1 = 0;
        nterm = 0;
      }
    };

    const int *rp = ve + vlane;
    int ent = rp[0];
    int ent_n = (1 < cnt) ? rp[vpitch] : ent;
    auto gather = [&](const int e, double4 &pm, int &mm) {
      const int m = e & 0x1FFFFFFF;  // (a padding slot gathers the block's first bead: never used)
      if (MODE == 0) {
        pm = s_ownpos[m - k0];
        mm = s_ownmeta[m - k0];
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
        // the pair is evaluated here only: the partner's terms too
        const bool partner = MODE == 0 ? true : (MODE == 1 ? false : ml < nown);
        unsigned long long *ap = s_acc + (partner ? ml : 0u);
        // the images' fast form holds for every term of this pair (else the checked conversion)
        const bool fast_sum = lam_ok && rsq >= SU.rsq_safe && fabs(lm - 0.5) <= 2.3;
        double evdwl = 0.0, fpair;
        double t0 = 0.0, t1 = 0.0, tu = 0.0;  // this bead's score terms and ucgforce term
        double p0 = 0.0, p1 = 0.0, pu = 0.0;  // the partner's
        if (STYLE == 0 || pseudo_flag == 0) {
          // pseudo-likelihood scores (:492-502): S[k][a] -= u[a][state of the neighbour] / kT
          t0 = -div_kT(sm ? q.u01 : q.u00, kT, rkT, kTp2);
          t1 = -div_kT(sm ? q.u11 : q.u10, kT, rkT, kTp2);
          if (MODE != 1) {
            // ... and S[m][b] -= u[state of this bead][b] / kT (:500-502)
            p0 = -div_kT(sk ? q.u10 : q.u00, kT, rkT, kTp2);
            p1 = -div_kT(sk ? q.u11 : q.u01, kT, rkT, kTp2);
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
          tu = -(lm * (q.u11 - q.u01) + (1. - lm) * (q.u10 - q.u00));
          if (MODE != 1) pu = -(lk * (q.u11 - q.u10) + (1. - lk) * (q.u01 - q.u00));
        } else {
          // Bethe closure in the reference's orientation (UCG/pair_table_ucg_bethe.cpp:544-604), as in k_pair_gather
          const double cu01 = k_is_i ? q.u01 : q.u10, cu10 = k_is_i ? q.u10 : q.u01;
          const double cf01 = k_is_i ? q.f01 : q.f10, cf10 = k_is_i ? q.f10 : q.f01;
          double pm_as_i1, pm_as_i0, pm_as_j1, pm_as_j0;
          {
            const double upm = MODE == 0 ? s_ownp[ml] : A.ucgp[m];
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
            double si0 = 0.0, si1 = 0.0, sj0 = 0.0, sj1 = 0.0;
            if (MODE != 1 || k_is_i) {
              const double pj0i0 = pij00 / pi0, pj0i1 = pij01 / pi0, pj1i0 = pij10 / pi1, pj1i1 = pij11 / pi1;
              si0 = -((pj0i0 * q.u00 + pj1i0 * cu01) / kT);
              si1 = -((pj0i1 * cu10 + pj1i1 * q.u11) / kT);
            }
            if (MODE != 1 || !k_is_i) {
              const double pi0j0 = pij00 / pj0, pi0j1 = pij10 / pj0, pi1j0 = pij01 / pj1, pi1j1 = pij11 / pj1;
              sj0 = -((pi0j0 * q.u00 + pi0j1 * cu01) / kT);
              sj1 = -((pi1j0 * cu10 + pi1j1 * q.u11) / kT);
            }
            t0 = k_is_i ? si0 : sj0;
            t1 = k_is_i ? si1 : sj1;
            p0 = k_is_i ? sj0 : si0;
            p1 = k_is_i ? sj1 : si1;
          }
          fpair = pij00 * q.f00 + pij01 * cf01 + pij10 * cf10 + pij11 * q.f11;
          if (EV) evdwl = pij00 * q.u00 + pij01 * cu01 + pij10 * cu10 + pij11 * q.u11;
        }
        const double tx = dx * fpair, ty = dy * fpair, tz = dz * fpair;
        unsigned long long ix, iy, iz;
        if (fast_sum) {
          ix = sum_raw_fast(tx, SU.sc_f);
          iy = sum_raw_fast(ty, SU.sc_f);
          iz = sum_raw_fast(tz, SU.sc_f);
          if (STYLE == 0) au += sum_raw_fast(tu, SU.sc_u);
          a0 += sum_raw_fast(t0, SU.sc_s);
          a1 += sum_raw_fast(t1, SU.sc_s);
        } else {
          ix = sum_raw_slow(tx, SU.sc_f, err);
          iy = sum_raw_slow(ty, SU.sc_f, err);
          iz = sum_raw_slow(tz, SU.sc_f, err);
          if (STYLE == 0) au += sum_raw_slow(tu, SU.sc_u, err);
          a0 += sum_raw_slow(t0, SU.sc_s, err);
          a1 += sum_raw_slow(t1, SU.sc_s, err);
        }
        ax += ix;
        ay += iy;
        az += iz;
        nterm++;
        if (MODE != 1 && partner) {
          // the partner's force terms are the negatives: image(-v) = -image(v) (round to nearest even is symmetric)
          atomicAdd(ap, (unsigned long long) SUM_MAGIC_BITS - ix);
          atomicAdd(ap + VR_BEADS, (unsigned long long) SUM_MAGIC_BITS - iy);
          atomicAdd(ap + 2 * VR_BEADS, (unsigned long long) SUM_MAGIC_BITS - iz);
          if (fast_sum) {
            if (STYLE == 0) atomicAdd(ap + 3 * VR_BEADS, sum_raw_fast(pu, SU.sc_u) - (unsigned long long) SUM_MAGIC_BITS);
            atomicAdd(ap + (NACC - 2) * VR_BEADS, sum_raw_fast(p0, SU.sc_s) - (unsigned long long) SUM_MAGIC_BITS);
            atomicAdd(ap + (NACC - 1) * VR_BEADS, sum_raw_fast(p1, SU.sc_s) - (unsigned long long) SUM_MAGIC_BITS);
          } else {
            if (STYLE == 0) atomicAdd(ap + 3 * VR_BEADS, sum_raw_slow(pu, SU.sc_u, err) - (unsigned long long) SUM_MAGIC_BITS);
            atomicAdd(ap + (NACC - 2) * VR_BEADS, sum_raw_slow(p0, SU.sc_s, err) - (unsigned long long) SUM_MAGIC_BITS);
            atomicAdd(ap + (NACC - 1) * VR_BEADS, sum_raw_slow(p1, SU.sc_s, err) - (unsigned long long) SUM_MAGIC_BITS);
          }
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

  const int2 *lmv = V.lanemeta + (size_t) vlane * VR_NLIST;
  sweep(std::integral_constant<int, 0>{}, V.ent[0], lmv[0]);
  sweep(std::integral_constant<int, 1>{}, V.ent[1], lmv[1]);
  for (int x = 2; x < VR_NLIST; x++) sweep(std::integral_constant<int, 2>{}, V.ent[x], lmv[x]);
  __syncthreads();  // every lane of the workgroup has made its adds

  const int k = k0 + (int) threadIdx.x;
  if (threadIdx.x < VR_BEADS && k < nlocal) {
    const unsigned long long *ap = s_acc + threadIdx.x;
    const double4 pk = s_ownpos[threadIdx.x];
    const int mk = s_ownmeta[threadIdx.x];
    const int tk = UCG_META_TYPE(mk);
    const double mu0 = P.mu[tk * 2 + 0], mu1 = P.mu[tk * 2 + 1];
    // the prologue values (:170-180 / bethe :155-162) plus the exact integer sums
    double fx = sum_decode((long long) ap[0], SU.dec_f), fy = sum_decode((long long) ap[VR_BEADS], SU.dec_f);
    double fz = sum_decode((long long) ap[2 * VR_BEADS], SU.dec_f);
    double uf = 0.0, s0, s1;
    if (STYLE == 0) {
      const double mui = mu1 - mu0;
      uf = -mui + sum_decode((long long) ap[3 * VR_BEADS], SU.dec_u);
      s0 = 0.0 + sum_decode((long long) ap[4 * VR_BEADS], SU.dec_s);
      s1 = -(mui / kT) + sum_decode((long long) ap[5 * VR_BEADS], SU.dec_s);
    } else {
      s0 = -mu0 / kT + sum_decode((long long) ap[3 * VR_BEADS], SU.dec_s);
      s1 = -mu1 / kT + sum_decode((long long) ap[4 * VR_BEADS], SU.dec_s);
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
int vrow_maxrow() { return VR_MAXROW; }
int vrow_lists() { return VR_NLIST; }

// slots per lane the lists of a full list with rows of at most `maxrow` entries can need: a block's kept entries are at
// most VR_BEADS * (maxrow + padding), cut into VR_LANES pieces
int vrow_capacity(int maxrow) { return ((VR_BEADS * (maxrow + VR_ALIGN) + VR_LANES - 1) / VR_LANES / VR_ALIGN + 1) * VR_ALIGN; }

size_t vrow_lds_bytes(const PairDev &P)
{
  const size_t tab = ((size_t) (P.tablength * P.fast_stride + 1) / 2) * sizeof(double4);
  const int nacc = P.style == 0 ? 6 : 5;
  return tab + (size_t) VR_BEADS * (sizeof(double4) + sizeof(int) + (P.style == 1 ? sizeof(double) : 0) + nacc * sizeof(unsigned long long));
}

// the virtual rows of every block, from the resident full rows
hipError_t launch_vrow_build(const PairDev &P, const AtomsDev &A, const ListDev &L, double skin, int *ent, size_t list_stride,
                             int cap, int vpitch, int2 *lanemeta, int *errflag, hipStream_t st)
{
  const int nb = vrow_blocks(A.nlocal);
  if (nb == 0) return hipSuccess;
  const size_t lds = (size_t) VR_STAGE_WORDS * sizeof(int);
  hipError_t e = hipFuncSetAttribute((const void *) k_vrow_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  if (e != hipSuccess) return e;
  VrowLists lists;
  for (int x = 0; x < VR_NLIST; x++) lists.ent[x] = ent + (size_t) x * list_stride;
  hipLaunchKernelGGL(k_vrow_build, dim3(nb), dim3(VR_BEADS), lds, st, A.nlocal, P.n_actual + 1, A.pos4, A.meta, P.cutsq, skin,
                     L.numneigh, L.neigh, L.pitch, lists, cap, vpitch, lanemeta, errflag);
  return hipGetLastError();
}

hipError_t launch_pair_vrow(const PairDev &P, const AtomsDev &A, const ListDev &L, const int *ent, size_t list_stride,
                            const int2 *lanemeta, int vpitch, bool ev, double *evpart, double *evout, int *errflag,
                            hipStream_t st)
{
  const int nblocks = vrow_blocks(A.nlocal);
  if (nblocks == 0) return hipSuccess;
  if (!P.fast || !P.tab_in_lds || P.tabstyle == 3) return hipErrorInvalidValue;
  const size_t lds = vrow_lds_bytes(P);
  if (lds + 4608 > 160 * 1024) return hipErrorInvalidValue;
  VrowDev V;
  for (int x = 0; x < VR_NLIST; x++) V.ent[x] = ent + (size_t) x * list_stride;
  V.lanemeta = lanemeta;
  V.vpitch = vpitch;
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
