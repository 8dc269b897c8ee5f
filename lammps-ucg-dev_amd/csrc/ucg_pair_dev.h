// ucg_pair_dev.h -- device helpers shared by the neighbour-loop kernels (ucg_pair.hip,
// ucg_density.hip): table interpolation, the exact division, deterministic block sums.
// Compiled with -ffp-contract=off; see ucg_pair.hip for the arithmetic contract.
#pragma once

#include "ucg_dev.h"
#include "ucg_launch.h"
#include "ucg_math.h"

namespace ucg {
namespace {

constexpr int PAIR_BLOCK = 1024;
// LDS bytes per staged own bead of the gather kernels: {x, y, z, lambda} + meta word (+ ucgp for table_ucg_bethe)
constexpr size_t pair_own_bytes(int style) { return sizeof(double4) + sizeof(int) + (style == 1 ? sizeof(double) : 0); }

struct Quad {
  double u00, u01, u10, u11;
  double f00, f01, f10, f11;
};
// ---- fixed sums (the kernels of ucg_pair_vrow.hip).  A bead's force components, ucgforce and scores are sums of
// per-pair terms.  Every term is rounded to nearest-even at 2^-38 of a per-field power-of-two unit (PairDev::sum_sc,
// fixed by the tables at ucg_pair_init so that it follows the unit system) and the images are added as 64-bit
// integers: associative and commutative, so the sums do not depend on the order of the additions, on how entries are
// dealt to lanes, on which lane evaluates a pair, or on the decomposition (the oracle states the same:
// oracle/orc_compute.c, sum_fixed).
// Fast path: bits(v * unit + 1.5 * 2^14) - bits(1.5 * 2^14) is that image while |v * unit| < 8192 (the sum's ulp is
// 2^-38 there); callers take it only where the tables guarantee the bound (PairDev::sum_rsq_safe, moderate lambda),
// the slow path converts in software (same rounding) up to 2^24 units and reports anything beyond as error bit 4.
constexpr double SUM_MAGIC = 24576.0;
constexpr long long SUM_MAGIC_BITS = 0x40D8000000000000ll;

struct SumDev {
  double sc_f, sc_u, sc_s;     // 2^e of the three fields
  double dec_f, dec_u, dec_s;  // 2^(-38 - e)
  double rsq_safe;
};
__device__ __forceinline__ SumDev sum_dev(const PairDev &P)
{
  return SumDev{P.sum_sc[0], P.sum_sc[1], P.sum_sc[2], P.sum_dec[0], P.sum_dec[1], P.sum_dec[2], P.sum_rsq_safe};
}
// RAW images: the image plus bits(1.5 * 2^14).  Accumulators add raw images modulo 2^64 and subtract (number of terms)
// x bits(1.5 * 2^14) once at the end (sum_finish): one fused multiply-add (v * unit is exact, so the fusion changes nothing)
// and one 64-bit integer addition per term.
__device__ __forceinline__ unsigned long long sum_raw_fast(const double v, const double sc)
{
  return (unsigned long long) __double_as_longlong(fma(v, sc, SUM_MAGIC));
}
__device__ __forceinline__ unsigned long long sum_raw_slow(const double v, const double sc, int &err)
{
  const double u = v * sc;
  long long im = 0;
  if (!(fabs(u) < 16777216.0)) err |= 4;
  else im = __double2ll_rn(u * 274877906944.0);  // 2^38: exact product, rounded to nearest even
  return (unsigned long long) im + (unsigned long long) SUM_MAGIC_BITS;
}
__device__ __forceinline__ unsigned long long sum_raw(const double v, const double sc, const bool fast, int &err)
{
  return fast ? sum_raw_fast(v, sc) : sum_raw_slow(v, sc, err);
}
// the integer sum of `nterms` raw images
__device__ __forceinline__ long long sum_finish(const unsigned long long raw, const unsigned nterms)
{
  // bits(1.5 * 2^14) has a zero low word: n x it is one 32-bit product in the high word (modulo 2^64 like the sum)
  return (long long) (raw - ((unsigned long long) (nterms * 0x40D80000u) << 32));
}
__device__ __forceinline__ double sum_decode(const long long s, const double dec) { return (double) s * dec; }


// correctly rounded a / b from y = RN(1/b): q0 = RN(a*y), then two FMA residual steps.
// Exact when b's significand is not all ones and neither the quotient nor the residuals leave the
// normal range: b within 2^+-200 (recip_ok / the host check of kT) and |a| within 1e+-200; anything
// else takes the hardware division.
__device__ __forceinline__ double div_by_const(const double a, const double b, const double y)
{
  if (!(fabs(a) > 1.0e-200 && fabs(a) < 1.0e200)) return a / b;
  double q = a * y;
  double r = fma(-b, q, a);
  q = fma(r, y, q);
  r = fma(-b, q, a);
  return fma(r, y, q);
}

// a / kT in the FAST kernels.  When kT is a power of two (kT = 1 in reduced units at T* = 1) its reciprocal
// is exact and a * (1/kT) IS the correctly rounded quotient; otherwise div_by_const.  `pow2` is uniform.
__device__ __forceinline__ double div_kT(const double a, const double kT, const double rkT, const int pow2)
{
  return pow2 ? a * rkT : div_by_const(a, kT, rkT);
}

// is b a denominator div_by_const is exact for?  (normal, mid-range exponent, significand not all ones)
__device__ __forceinline__ bool recip_ok(const double b)
{
  const unsigned long long kb = (unsigned long long) __double_as_longlong(b);
  const unsigned long long mant = kb & 0xFFFFFFFFFFFFFull, ex = (kb >> 52) & 0x7FF;
  return mant != 0xFFFFFFFFFFFFFull && ex >= 1023 - 200 && ex <= 1023 + 200;
}

// knot index of rsq on one table's r^2 grid (the shared part of UCG/pair_table_ucgld.cpp:436-459)
__device__ __forceinline__ int grid_locate(const double4 par, const int tlm1, const double rsq, int &err)
{
  if (rsq < par.x) err |= 1;
  int it = static_cast<int>((rsq - par.x) * par.z);
  if (it >= tlm1) {
    err |= 2;
    it = tlm1 - 1;
  }
  if (it < 0) it = 0;
  return it;
}

// the same for the FAST kernels (one shared grid): instead of two compares and flag updates per pair they keep the
// smallest r^2 and the largest raw knot index seen, and the two range tests are made once per row (range_flags)
// (rsq < innersq  <=>  rsq - innersq < 0, the subtraction the knot index needs anyway: the inner test is the OR of the
// differences' sign words -- one 32-bit instruction per pair where a running fmin cost three fp64 ones)
struct RangeTrack {
  int neg_or;
  int it_max;
};
__device__ __forceinline__ RangeTrack range_track_init() { return RangeTrack{0, -1}; }
__device__ __forceinline__ int grid_locate_track(const double4 par, const int tlm1, const double rsq, RangeTrack &rt)
{
  const double d = rsq - par.x;
  rt.neg_or |= __double2hiint(d);
  const int raw = static_cast<int>(d * par.z);
  rt.it_max = max(rt.it_max, raw);
  int it = min(raw, tlm1 - 1);
  if (it < 0) it = 0;
  return it;
}
__device__ __forceinline__ int range_flags(const double4 /*par*/, const int tlm1, const RangeTrack &rt)
{
  return (rt.neg_or < 0 ? 1 : 0) | (rt.it_max >= tlm1 ? 2 : 0);
}

// a wavefront-uniform double, moved to scalar registers
__device__ __forceinline__ double uniform_f64(const double v)
{
  const unsigned long long b = (unsigned long long) __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) b), hi = __builtin_amdgcn_readfirstlane((unsigned) (b >> 32));
  return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

struct Basis {
  double a, b, a3, b3;  // SPLINE: a, b, a^3-a, b^3-b ; LINEAR: b = fraction
};

template <int TS>
__device__ __forceinline__ Basis grid_basis(const double4 par, const int it, const double rsq)
{
  Basis B;
  B.a = B.b = B.a3 = B.b3 = 0.0;
  if (TS != 0) {
    const double rsq_it = par.x + it * par.y;  // == tb->rsq[it] bit for bit (:1159,:1196)
    B.b = (rsq - rsq_it) * par.z;
    if (TS == 2) {
      B.a = 1.0 - B.b;
      B.a3 = B.a * B.a * B.a - B.a;
      B.b3 = B.b * B.b * B.b - B.b;
    }
  }
  return B;
}

// (f/r, e) of one table at a located knot; UCG/pair_table_ucgld.cpp:440-481
template <int TS, typename TabPtr>
__device__ __forceinline__ void knot_eval(TabPtr tab, const double deltasq6, const int it, const Basis &B,
                                          double &fval, double &eval)
{
  if (TS == 0) {  // LOOKUP {e, f, -, -}
    const double4 k = tab[it];
    eval = k.x;
    fval = k.y;
  } else if (TS == 1) {  // LINEAR {e, de, f, df}
    const double4 k = tab[it];
    fval = k.z + B.b * k.w;
    eval = k.x + B.b * k.y;
  } else {  // SPLINE {e, f, e2, f2}
    const double4 k0 = tab[it];
    const double4 k1 = tab[it + 1];
    fval = B.a * k0.y + B.b * k1.y + (B.a3 * k0.w + B.b3 * k1.w) * deltasq6;
    eval = B.a * k0.x + B.b * k1.x + (B.a3 * k0.z + B.b3 * k1.z) * deltasq6;
  }
}

// FAST layout: knot-major, the tables of one knot side by side, each {e,f | e2,f2} (or the
// LINEAR / LOOKUP quartet) as two 16-byte slots, plus ONE padding slot per knot so that the
// knot stride (2*ntab+1 slots) is odd: a random knot index then lands on any of the 16
// ds_read_b128 bank slots, instead of only 8 of them with a 32-byte stride.
template <int TS>
__device__ __forceinline__ void knot_eval_fast(const double2 *rec, const int stride, const double deltasq6,
                                               const Basis &B, double &fval, double &eval)
{
  if (TS == 0) {
    const double2 k = rec[0];
    eval = k.x;
    fval = k.y;
  } else if (TS == 1) {
    const double2 ka = rec[0], kb = rec[1];  // {e, de}, {f, df}
    fval = kb.x + B.b * kb.y;
    eval = ka.x + B.b * ka.y;
  } else {
    const double2 k0a = rec[0], k0b = rec[1];            // {e, f}, {e2, f2} at knot it
    const double2 k1a = rec[stride], k1b = rec[stride + 1];  // ... at knot it+1
    fval = B.a * k0a.y + B.b * k1a.y + (B.a3 * k0b.y + B.b3 * k1b.y) * deltasq6;
    eval = B.a * k0a.x + B.b * k1a.x + (B.a3 * k0b.x + B.b3 * k1b.x) * deltasq6;
  }
}

// one table of the generic (per-table parameters, tables in HBM / L2) path
template <int TS, typename TabPtr>
__device__ __forceinline__ void table_eval(TabPtr tab, const double4 par, const int tlm1, const double rsq, int &err,
                                           double &fval, double &eval)
{
  if (TS == 3) {
    // BITMAP (UCG/pair_table_ucgld.cpp:466-476): the bin is cut out of the bits of (float) rsq; par = {innersq,
    // nmask, nshiftbits}; records {e, de, f, df}, {rsq, drsq}.  Only the inner cutoff is checked in this branch.
    if (rsq < par.x) err |= 1;
    const float fl = (float) rsq;
    const int it = (__float_as_int(fl) & (int) par.y) >> (int) par.z;
    const double4 k = tab[2 * it];
    const double4 g = tab[2 * it + 1];
    const double fraction = ((double) fl - g.x) * g.y;
    fval = k.z + fraction * k.w;
    eval = k.x + fraction * k.y;
  } else {
    const int it = grid_locate(par, tlm1, rsq, err);
    knot_eval<TS>(tab, par.w, it, grid_basis<TS>(par, it, rsq), fval, eval);
  }
}

// own-frame quad: u[a][b] = table(F(tk,a), F(tm,b)); equals the reference's u[b][a] when the
// row owner is the pair's "j" (tabindex is symmetric after init_one)
// SAME10: the caller knows t10 == t01 at compile time (one actual type: tabindex is symmetric after init_one, SURVEY.md
// App. B #26), so the mixed-state values are one evaluation and one set of registers.  parF: the shared grid's parameters
// held by the caller (scalar registers), instead of a read of s_par[0] per pair.
template <int TS, bool FAST, bool SAME10 = false, typename TabPtr>
__device__ __forceinline__ void eval_quad(TabPtr tab, const double4 *s_par, const int *pt, const int tablength,
                                          const int tlm1, const int fast_stride, const double rsq,
                                          const double factor_lj, Quad &q, int &err, RangeTrack &rt,
                                          const double2 *lds_hot = nullptr, const bool hot = false, const int hot_k0 = -1,
                                          const double4 *parF = nullptr)
{
  const int t00 = pt[0], t01 = pt[1], t10 = pt[2], t11 = pt[3];
  if (FAST) {
    const double4 par = parF ? *parF : s_par[0];
    const int it = grid_locate_track(par, tlm1, rsq, rt);
    const Basis B = grid_basis<TS>(par, it, rsq);
    // hot lanes (PairDev::hot_type: both beads of the actual type whose three tables are staged in LDS) read the LDS
    // copy {t00, t01 = t10, t11} at stride 7, the others the full layout through L1 / L2: one instruction stream, the
    // two pointers selected per lane (generic-address loads).  Same values either way.
    // (hot_k0 >= 0, one actual type with long tables: the LDS holds the knots from hot_k0 on in the same layout, and the
    // lanes whose knot lies there are the hot ones)
    const bool window = hot_k0 >= 0;  // uniform
    const bool h = lds_hot != nullptr && (window ? it >= hot_k0 : hot);
    const bool hb = h && !window;
    const double2 *rec = h ? (window ? lds_hot + (it - hot_k0) * fast_stride : lds_hot + it * 7)
                           : reinterpret_cast<const double2 *>(tab) + it * fast_stride;
    const int st = hb ? 7 : fast_stride;
    const int o00 = hb ? 0 : 2 * t00, o01 = hb ? 2 : 2 * t01, o10 = hb ? 2 : 2 * t10, o11 = hb ? 4 : 2 * t11;
    knot_eval_fast<TS>(rec + o00, st, par.w, B, q.f00, q.u00);
    knot_eval_fast<TS>(rec + o01, st, par.w, B, q.f01, q.u01);
    if (SAME10 || t10 == t01) {
      q.f10 = q.f01;
      q.u10 = q.u01;
    } else {
      knot_eval_fast<TS>(rec + o10, st, par.w, B, q.f10, q.u10);
    }
    knot_eval_fast<TS>(rec + o11, st, par.w, B, q.f11, q.u11);
  } else {
    table_eval<TS>(tab + t00 * tablength, s_par[t00], tlm1, rsq, err, q.f00, q.u00);
    table_eval<TS>(tab + t01 * tablength, s_par[t01], tlm1, rsq, err, q.f01, q.u01);
    if (t10 == t01) {
      q.f10 = q.f01;
      q.u10 = q.u01;
    } else {
      table_eval<TS>(tab + t10 * tablength, s_par[t10], tlm1, rsq, err, q.f10, q.u10);
    }
    table_eval<TS>(tab + t11 * tablength, s_par[t11], tlm1, rsq, err, q.f11, q.u11);
    q.f00 = factor_lj * q.f00; q.u00 *= factor_lj;
    q.f01 = factor_lj * q.f01; q.u01 *= factor_lj;
    q.f10 = factor_lj * q.f10; q.u10 *= factor_lj;
    q.f11 = factor_lj * q.f11; q.u11 *= factor_lj;
  }
}

// The quad of one entry from a compact block: knot-major records {t00, t01, [t10,] t11} x two 16-byte slots + one padding
// slot (a kind block, KindsDev, or the LDS hot block); o10 == 2: the mixed-state tables are one table.  The same values
// as eval_quad's FAST branch.
template <int TS>
__device__ __forceinline__ void eval_quad_kind(const double2 *blk, const int stride, const int o10, const int o11,
                                               const double4 par, const int tlm1, const double rsq, Quad &q, RangeTrack &rt)
{
  const int it = grid_locate_track(par, tlm1, rsq, rt);
  const Basis B = grid_basis<TS>(par, it, rsq);
  const double2 *rec = blk + it * stride;
  knot_eval_fast<TS>(rec, stride, par.w, B, q.f00, q.u00);
  knot_eval_fast<TS>(rec + 2, stride, par.w, B, q.f01, q.u01);
  if (o10 == 2) {
    q.f10 = q.f01;
    q.u10 = q.u01;
  } else {
    knot_eval_fast<TS>(rec + o10, stride, par.w, B, q.f10, q.u10);
  }
  knot_eval_fast<TS>(rec + o11, stride, par.w, B, q.f11, q.u11);
}

// one word of a neighbour row: a plain load, or a non-temporal one when the rows are streamed (ListDev::stream_rows; uniform)
__device__ __forceinline__ int row_load(const int *p, const int stream)
{
  return stream ? __builtin_nontemporal_load(p) : *p;
}

// bias_force of fix nve/ucgld/wall/hard (UCG/fix_nve_ucgld_wall_hard.cpp:216-221), as in csrc/ucg_fix.hip
__device__ __forceinline__ double post_wall_bias(const double lmd, const double H)
{
  const double x = lmd - 0.5;
  return (-7980.0 * x * x * x * x * x * x * x * x * x + 2.0 * x) * 10.0 * H;
}

// The per-bead hooks that follow the pair force on a step whose next initial_integrate is fused in -- the statements of
// k_post_fused<.., NEXT = true> (csrc/ucg_fix.hip), in the same order -- applied by the lane that holds a bead's summed
// force, ucgforce and scores (the gather kernels' epilogue).  x / lambda / state / ucgp of the next step go to the second
// buffers (other workgroups still gather from the current ones).
template <int STYLE>
__device__ __forceinline__ void pair_epilogue(const AtomsDev &A, const PostDev &Q, const int k, const int mk, const double4 pk,
                                              const double fx, const double fy, const double fz, const double uf,
                                              const double s0, const double s1)
{
  const int tk = UCG_META_TYPE(mk);
  {
    // Epilogue: the statements of k_post_fused<.., NEXT = true> (csrc/ucg_fix.hip), in the same order, on
    // the sums this lane holds: [wall/hard bias ->] ucgld/langevin -> ucgstate -> final_integrate -> the
    // next step's initial_integrate.  f, ucgforce and the scores never reach HBM; x / lambda / state of the
    // next step go to the second buffers (other workgroups still gather from the current ones).
    const bool ingroup = (A.mask[k] & Q.groupbit) != 0;
    int meta = mk;
    double4 f = make_double4(fx, fy, fz, STYLE == 0 ? uf : 0.0);
    double4 v = A.vel4[k];
    double4 x = pk;
    if (Q.nve == 3 && ingroup) f.w += post_wall_bias(x.w, Q.barrier);
    if (Q.lang && ingroup) {
      const double gamma1 = Q.gfactor1[tk];
      const double gamma2 = Q.gfactor2[tk] * Q.tsqrt;
      const double uni = (double) Q.lang_draws[k] * 5.9604644775390625e-08;
      double fran = gamma2 * (uni - 0.5);
      if (Q.lang_bias && v.w == 0.0) fran = 0.0;
      const double fdrag = gamma1 * v.w;
      f.w += fdrag + fran;
    }
    if (Q.ucgst) {
      // num_ucgstates is 2 for every bead these kernels handle (set right here in the plain path)
      const double e0 = ucg_exp_nb((700.0 < s0) ? 700.0 : s0);
      const double e1 = ucg_exp_nb((700.0 < s1) ? 700.0 : s1);
      double softmax_denom = 0.0;
      softmax_denom += e0;
      softmax_denom += e1;
      const double r = e1 / softmax_denom;
      const double lo = (1e-6 < r) ? r : 1e-6;
      const double ucgp = (lo < 1.0 - 1e-6) ? lo : 1.0 - 1e-6;
      if (!Q.ld_flag) {
        int state;
        if (Q.mc_flag) {
          const int cur = UCG_META_STATE(meta);
          double mc_factor;
          if (cur == 0) mc_factor = ucgp / (1.0 - ucgp);
          else mc_factor = (1.0 - ucgp) / ucgp;
          mc_factor = ((1.0 < mc_factor) ? 1.0 : mc_factor) * Q.mc_rate;
          const double mc_rand = (double) Q.mc_draws[k] * 5.9604644775390625e-08;
          state = (mc_rand < mc_factor) ? 0 : 1;
        } else {
          state = (int) round(ucgp);
        }
        meta = (meta & 0xFFFF) | (state << 16);
        x.w = ucgp;
      }
      Q.ucgp_out[k] = ucgp;  // second buffer: the Bethe variant gathers its neighbours' ucgp in this very launch
    }
    if (Q.nve && ingroup) {
      const double dtfm = Q.dtf / A.mass[UCG_META_TYPE(meta)];
      const double dtflm = Q.dtf / A.ucgml[k];
      v.x += dtfm * f.x;
      v.y += dtfm * f.y;
      v.z += dtfm * f.z;
      v.w += dtflm * f.w;
      if (Q.nve >= 2) {
        if (x.w < 0.0) {
          x.w = -x.w;
          v.w = -v.w;
        } else if (x.w > 1.0) {
          x.w = 2.0 - x.w;
          v.w = -v.w;
        }
      }
      v.x += dtfm * f.x;
      v.y += dtfm * f.y;
      v.z += dtfm * f.z;
      x.x += Q.dtv * v.x;
      x.y += Q.dtv * v.y;
      x.z += Q.dtv * v.z;
      v.w += dtflm * f.w;
      x.w += Q.dtv * v.w;
      if (Q.nve >= 2) meta = (meta & 0xFFFF) | ((x.w < 0.5 ? 0 : 1) << 16);
      A.vel4[k] = v;
    }
    Q.pos_out[k] = x;
    Q.meta_out[k] = meta;
    A.num_ucgstates[k] = 2;
  }
}

// deterministic block sum of NV doubles per lane -> out[blockIdx.x*NV + c]
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double *red, double *out)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int c = 0; c < NV; c++) {
    double s = v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave * NV + c] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < nw; w++) s += red[w * NV + threadIdx.x];
    out[(size_t) blockIdx.x * NV + threadIdx.x] = s;
  }
}

// XCD-aware chunk order: blocks b and b+8 share an XCD (observed round-robin), so give
// each XCD one contiguous range of bead chunks; beads are bin-sorted, so a range is a
// spatial slab whose neighbour gathers stay in that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_chunk(int b, int nb)
{
  const int per = nb >> 3;
  if (per == 0 || b >= per * 8) return b;
  return (b & 7) * per + (b >> 3);
}


}  // namespace
}  // namespace ucg
