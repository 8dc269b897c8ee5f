// ucg_fix.hip -- per-bead kernels: fix nve/ucgld, fix ucgld/langevin, fix ucgstate,
// force_clear and the small thermo reductions (gfx950).
//
//   k_nve_initial / k_nve_final   FixNVE_UCGLD::initial_integrate / final_integrate
//                                 (UCG/fix_nve_ucgld.cpp:44-101, 104-153, per-type mass branch)
//                                 wall != 0: FixNVE_UCGLD_Wall_Hard (UCG/fix_nve_ucgld_wall_hard.cpp:61-199),
//                                 i.e. ucgstate from lambda after the drift, reflection of lambda at 0 / 1
//   k_wall_bias                   FixNVE_UCGLD_Wall_Hard::post_force + bias_force (:216-241)
//   k_langevin                    Fix_UCGLD_Langevin::post_force_templated<0>
//                                 (UCG/fix_ucgld_langevin.cpp:226-297)
//   k_lambda_ke                   Fix_UCGLD_Langevin::end_of_step (:303-312)
//   k_ucgstate                    FixUCGState::post_force (UCG/fix_ucgstate.cpp:88-132)
//   k_force_clear                 AtomVecUCG::force_clear (UCG/atom_vec_ucg.cpp:131-135)
//
// lambda rides as the 4th component of pos4 / vel4 / frc4, so the (x,v) and the
// (lambda, v_lambda) velocity-Verlet updates are the same three double4 streams.
// All of these are HBM-streaming kernels: one bead per lane, 32-byte accesses.
// Compiled with -ffp-contract=off (bit parity with the scalar reference).
#include "ucg_dev.h"
#include "ucg_launch.h"
#include "ucg_math.h"

namespace ucg {

namespace {

constexpr int FIX_BLOCK = 256;

// bias_force (UCG/fix_nve_ucgld_wall_hard.cpp:216-221), products left to right as written there
__device__ __forceinline__ double wall_bias_force(const double lmd, const double H)
{
  const double x = lmd - 0.5;
  return (-7980.0 * x * x * x * x * x * x * x * x * x + 2.0 * x) * 10.0 * H;
}

__global__ __launch_bounds__(FIX_BLOCK) void k_nve_initial(const AtomsDev A, const double dtv, const double dtf,
                                                          const int groupbit, const int wall)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  if (!(A.mask[i] & groupbit)) return;
  double4 x = A.pos4[i];
  double4 v = A.vel4[i];
  const double4 f = A.frc4[i];
  const double dtfm = dtf / A.mass[UCG_META_TYPE(A.meta[i])];
  v.x += dtfm * f.x;
  v.y += dtfm * f.y;
  v.z += dtfm * f.z;
  x.x += dtv * v.x;
  x.y += dtv * v.y;
  x.z += dtv * v.z;
  const double dtflm = dtf / A.ucgml[i];
  v.w += dtflm * f.w;
  x.w += dtv * v.w;
  A.vel4[i] = v;
  A.pos4[i] = x;
  if (wall) A.meta[i] = (A.meta[i] & 0xFFFF) | ((x.w < 0.5 ? 0 : 1) << 16);
}

__global__ __launch_bounds__(FIX_BLOCK) void k_nve_final(const AtomsDev A, const double dtf, const int groupbit,
                                                        const int wall)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  if (!(A.mask[i] & groupbit)) return;
  double4 v = A.vel4[i];
  const double4 f = A.frc4[i];
  const double dtfm = dtf / A.mass[UCG_META_TYPE(A.meta[i])];
  v.x += dtfm * f.x;
  v.y += dtfm * f.y;
  v.z += dtfm * f.z;
  const double dtflm = dtf / A.ucgml[i];
  v.w += dtflm * f.w;
  if (wall) {
    double4 x = A.pos4[i];
    if (x.w < 0.0) {
      x.w = -x.w;
      v.w = -v.w;
      A.pos4[i] = x;
    } else if (x.w > 1.0) {
      x.w = 2.0 - x.w;
      v.w = -v.w;
      A.pos4[i] = x;
    }
  }
  A.vel4[i] = v;
}

__global__ __launch_bounds__(FIX_BLOCK) void k_wall_bias(const AtomsDev A, const double barrier, const int groupbit)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  if (!(A.mask[i] & groupbit)) return;
  double4 f = A.frc4[i];
  f.w += wall_bias_force(A.pos4[i].w, barrier);
  A.frc4[i] = f;
}

__global__ __launch_bounds__(FIX_BLOCK) void k_langevin(const AtomsDev A, const LangevinDev Lg, const int groupbit)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  if (!(A.mask[i] & groupbit)) return;
  const int t = UCG_META_TYPE(A.meta[i]);
  const double gamma1 = Lg.gfactor1[t];
  const double gamma2 = Lg.gfactor2[t] * Lg.tsqrt;
  // RanMars::uniform() is an exact multiple of 2^-24
  const double uni = (double) Lg.draws[i] * 5.9604644775390625e-08;
  double fran = gamma2 * (uni - 0.5);
  const double vl = A.vel4[i].w;
  if (Lg.bias && vl == 0.0) fran = 0.0;  // post_force_templated<1> (UCG/fix_ucgld_langevin.cpp:283-291)
  const double fdrag = gamma1 * vl;
  double4 f = A.frc4[i];
  f.w += fdrag + fran;
  A.frc4[i] = f;
}

__global__ __launch_bounds__(FIX_BLOCK) void k_ucgstate(const AtomsDev A, const int ld_flag, const int mc_flag,
                                                       const double mc_rate, const unsigned int *draws)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  int meta = A.meta[i];
  double ucgp;
  if (A.num_ucgstates[i] == 1) {
    if (!ld_flag) meta &= 0xFFFF;
    ucgp = 1.0;
  } else {
    const double2 s = A.scores[i];
    const double e0 = ucg_exp_nb((700.0 < s.x) ? 700.0 : s.x);
    const double e1 = ucg_exp_nb((700.0 < s.y) ? 700.0 : s.y);
    double softmax_denom = 0.0;
    softmax_denom += e0;
    softmax_denom += e1;
    const double r = e1 / softmax_denom;
    const double lo = (1e-6 < r) ? r : 1e-6;
    ucgp = (lo < 1.0 - 1e-6) ? lo : 1.0 - 1e-6;
    if (!ld_flag) {
      int state;
      if (mc_flag) {
        const int cur = UCG_META_STATE(meta);
        double mc_factor;
        if (cur == 0) mc_factor = ucgp / (1.0 - ucgp);
        else mc_factor = (1.0 - ucgp) / ucgp;
        mc_factor = ((1.0 < mc_factor) ? 1.0 : mc_factor) * mc_rate;
        const double mc_rand = (double) draws[i] * 5.9604644775390625e-08;
        state = (mc_rand < mc_factor) ? 0 : 1;
      } else {
        state = (int) round(ucgp);
      }
      meta = (meta & 0xFFFF) | (state << 16);
    }
  }
  A.ucgp[i] = ucgp;
  if (!ld_flag) {
    A.meta[i] = meta;
    double4 x = A.pos4[i];
    x.w = ucgp;
    A.pos4[i] = x;
  }
}

// Resident-loop fusion of the per-bead hooks that follow the pair kernel, in the reference's
// order: [wall/hard bias post_force ->] fix ucgld/langevin post_force -> fix ucgstate post_force ->
// fix nve/ucgld[/wall/hard] final_integrate
// [-> the NEXT step's initial_integrate when nothing has to look at the state in between].
// Each bead is independent and every statement is the one of the stand-alone kernels above,
// in the same order, so the results are bit-identical; only the HBM passes are merged.
template <bool LANG, bool UCGST, bool NVE, bool NEXT>
__global__ __launch_bounds__(FIX_BLOCK) void k_post_fused(const AtomsDev A, const LangevinDev Lg, const int ld_flag,
                                                         const int mc_flag, const double mc_rate,
                                                         const unsigned int *mc_draws, const double dtv, const double dtf,
                                                         const int groupbit, const int wall, const double barrier)
{
  // wall: 0 fix nve/ucgld, 2 fix nve/ucgld/wall/hard, 3 the same with bias_potential (uniform branches)
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  const bool ingroup = (A.mask[i] & groupbit) != 0;
  int meta = A.meta[i];
  double4 f = A.frc4[i];
  double4 v = A.vel4[i];
  double4 x;
  bool x_loaded = false, x_dirty = false, f_dirty = false, meta_dirty = false;
  if (NVE && wall == 3 && ingroup) {
    x = A.pos4[i];
    x_loaded = true;
    f.w += wall_bias_force(x.w, barrier);
    f_dirty = true;
  }
  if (LANG && ingroup) {
    const int t = UCG_META_TYPE(meta);
    const double gamma1 = Lg.gfactor1[t];
    const double gamma2 = Lg.gfactor2[t] * Lg.tsqrt;
    const double uni = (double) Lg.draws[i] * 5.9604644775390625e-08;
    double fran = gamma2 * (uni - 0.5);
    if (Lg.bias && v.w == 0.0) fran = 0.0;
    const double fdrag = gamma1 * v.w;
    f.w += fdrag + fran;
    f_dirty = true;
  }
  // with the next initial_integrate fused in, nothing reads ucgforce before the next pair kernel overwrites it
  if (f_dirty && !(NVE && NEXT)) A.frc4[i] = f;
  if (UCGST) {
    double ucgp;
    if (A.num_ucgstates[i] == 1) {
      if (!ld_flag) { meta &= 0xFFFF; meta_dirty = true; }
      ucgp = 1.0;
    } else {
      const double2 s = A.scores[i];
      const double e0 = ucg_exp_nb((700.0 < s.x) ? 700.0 : s.x);
      const double e1 = ucg_exp_nb((700.0 < s.y) ? 700.0 : s.y);
      double softmax_denom = 0.0;
      softmax_denom += e0;
      softmax_denom += e1;
      const double r = e1 / softmax_denom;
      const double lo = (1e-6 < r) ? r : 1e-6;
      ucgp = (lo < 1.0 - 1e-6) ? lo : 1.0 - 1e-6;
      if (!ld_flag) {
        int state;
        if (mc_flag) {
          const int cur = UCG_META_STATE(meta);
          double mc_factor;
          if (cur == 0) mc_factor = ucgp / (1.0 - ucgp);
          else mc_factor = (1.0 - ucgp) / ucgp;
          mc_factor = ((1.0 < mc_factor) ? 1.0 : mc_factor) * mc_rate;
          const double mc_rand = (double) mc_draws[i] * 5.9604644775390625e-08;
          state = (mc_rand < mc_factor) ? 0 : 1;
        } else {
          state = (int) round(ucgp);
        }
        meta = (meta & 0xFFFF) | (state << 16);
        meta_dirty = true;
      }
    }
    A.ucgp[i] = ucgp;
    if (!ld_flag) {
      if (!x_loaded) x = A.pos4[i];
      x_loaded = true;
      x.w = ucgp;
      x_dirty = true;
    }
  }
  if (NVE && ingroup) {
    const double dtfm = dtf / A.mass[UCG_META_TYPE(meta)];
    const double dtflm = dtf / A.ucgml[i];
    // final_integrate of this step
    v.x += dtfm * f.x;
    v.y += dtfm * f.y;
    v.z += dtfm * f.z;
    v.w += dtflm * f.w;
    if (wall) {
      if (!x_loaded) x = A.pos4[i];
      x_loaded = true;
      if (x.w < 0.0) {
        x.w = -x.w;
        v.w = -v.w;
        x_dirty = true;
      } else if (x.w > 1.0) {
        x.w = 2.0 - x.w;
        v.w = -v.w;
        x_dirty = true;
      }
    }
    if (NEXT) {
      // initial_integrate of the next step (same forces)
      if (!x_loaded) x = A.pos4[i];
      v.x += dtfm * f.x;
      v.y += dtfm * f.y;
      v.z += dtfm * f.z;
      x.x += dtv * v.x;
      x.y += dtv * v.y;
      x.z += dtv * v.z;
      v.w += dtflm * f.w;
      x.w += dtv * v.w;
      x_dirty = true;
      if (wall) {
        meta = (meta & 0xFFFF) | ((x.w < 0.5 ? 0 : 1) << 16);
        meta_dirty = true;
      }
    }
    A.vel4[i] = v;
  }
  if (meta_dirty) A.meta[i] = meta;
  if (x_dirty) A.pos4[i] = x;
}

__global__ __launch_bounds__(FIX_BLOCK) void k_force_clear(const AtomsDev A)
{
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  if (i >= A.nlocal) return;
  A.frc4[i] = make_double4(0.0, 0.0, 0.0, 0.0);
  A.scores[i] = make_double2(0.0, 0.0);
}

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

// partial sums: [0] lambda kinetic energy, [1] beads in state 1, [2] sum of lambda, [3] translational KE*2
__global__ __launch_bounds__(FIX_BLOCK) void k_thermo_part(const AtomsDev A, const int groupbit, const double mvv2e,
                                                          double *part)
{
  __shared__ double red[(FIX_BLOCK / 64) * 4];
  const int i = blockIdx.x * FIX_BLOCK + threadIdx.x;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < A.nlocal) {
    const int meta = A.meta[i];
    v[1] = (double) UCG_META_STATE(meta);
    v[2] = A.pos4[i].w;
    if (A.mask[i] & groupbit) {
      const double4 vel = A.vel4[i];
      v[0] = 0.5 * A.ucgml[i] * vel.w * vel.w * mvv2e;
      v[3] = A.mass[UCG_META_TYPE(meta)] * (vel.x * vel.x + vel.y * vel.y + vel.z * vel.z) * mvv2e;
    }
  }
  block_sum_store<4>(v, red, part);
}

__global__ void k_thermo_final(const double *part, int nblocks, double *out)
{
  const int c = threadIdx.x;
  if (c < 4) {
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += part[(size_t) b * 4 + c];
    out[c] = s;
  }
}

// PMC calibration: stream `n` elements of width 4 B (WIDE=false) or 16 B (WIDE=true) per lane
template <bool WIDE>
__global__ __launch_bounds__(FIX_BLOCK) void k_stream(const int4 *p, size_t n, int *sink)
{
  size_t i = (size_t) blockIdx.x * FIX_BLOCK + threadIdx.x;
  const size_t stride = (size_t) gridDim.x * FIX_BLOCK;
  int acc = 0;
  if (WIDE) {
    for (; i < n; i += stride) {
      const int4 v = p[i];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  } else {
    const int *q = reinterpret_cast<const int *>(p);
    for (; i < n; i += stride) acc += q[i];
  }
  if (acc == 0x7fffffff) *sink = acc;
}

inline int nblk(int n) { return (n + FIX_BLOCK - 1) / FIX_BLOCK; }

}  // namespace

hipError_t launch_stream(const void *buf, size_t nbytes, int wide, int *sink, hipStream_t st)
{
  if (wide) hipLaunchKernelGGL(k_stream<true>, dim3(2048), dim3(FIX_BLOCK), 0, st, (const int4 *) buf, nbytes / 16, sink);
  else hipLaunchKernelGGL(k_stream<false>, dim3(2048), dim3(FIX_BLOCK), 0, st, (const int4 *) buf, nbytes / 4, sink);
  return hipGetLastError();
}

hipError_t launch_nve_initial(const AtomsDev &A, double dtv, double dtf, int groupbit, int wall, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_nve_initial, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A, dtv, dtf, groupbit, wall);
  return hipGetLastError();
}

hipError_t launch_nve_final(const AtomsDev &A, double dtf, int groupbit, int wall, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_nve_final, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A, dtf, groupbit, wall);
  return hipGetLastError();
}

hipError_t launch_wall_bias(const AtomsDev &A, double barrier, int groupbit, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wall_bias, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A, barrier, groupbit);
  return hipGetLastError();
}

hipError_t launch_langevin(const AtomsDev &A, const LangevinDev &Lg, int groupbit, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_langevin, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A, Lg, groupbit);
  return hipGetLastError();
}

hipError_t launch_ucgstate(const AtomsDev &A, int ld_flag, int mc_flag, double mc_rate,
                           const unsigned int *draws, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_ucgstate, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A, ld_flag, mc_flag, mc_rate, draws);
  return hipGetLastError();
}

hipError_t launch_post_fused(const AtomsDev &A, bool lang, const LangevinDev &Lg, bool ucgst, int ld_flag, int mc_flag,
                             double mc_rate, const unsigned int *mc_draws, bool nve, bool next, double dtv, double dtf,
                             int groupbit, int wall, double barrier, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  const dim3 g(nblk(A.nlocal)), b(FIX_BLOCK);
#define UCG_PF(L, U, N, X) \
  hipLaunchKernelGGL((k_post_fused<L, U, N, X>), g, b, 0, st, A, Lg, ld_flag, mc_flag, mc_rate, mc_draws, dtv, dtf, groupbit, \
                     wall, barrier)
  const int sel = (lang ? 8 : 0) | (ucgst ? 4 : 0) | (nve ? 2 : 0) | ((nve && next) ? 1 : 0);
  switch (sel) {
    case 0: break;
    case 2: UCG_PF(false, false, true, false); break;
    case 3: UCG_PF(false, false, true, true); break;
    case 4: UCG_PF(false, true, false, false); break;
    case 6: UCG_PF(false, true, true, false); break;
    case 7: UCG_PF(false, true, true, true); break;
    case 8: UCG_PF(true, false, false, false); break;
    case 10: UCG_PF(true, false, true, false); break;
    case 11: UCG_PF(true, false, true, true); break;
    case 12: UCG_PF(true, true, false, false); break;
    case 14: UCG_PF(true, true, true, false); break;
    default: UCG_PF(true, true, true, true); break;
  }
#undef UCG_PF
  return hipGetLastError();
}

hipError_t launch_force_clear(const AtomsDev &A, hipStream_t st)
{
  if (A.nlocal == 0) return hipSuccess;
  hipLaunchKernelGGL(k_force_clear, dim3(nblk(A.nlocal)), dim3(FIX_BLOCK), 0, st, A);
  return hipGetLastError();
}

// out[0..3] = {lambda KE, state-1 count, sum lambda, 2*KE}; part must hold 4*nblk doubles
hipError_t launch_lambda_ke(const AtomsDev &A, int groupbit, double mvv2e, double *part, double *out,
                            hipStream_t st)
{
  const int nb = nblk(A.nlocal > 0 ? A.nlocal : 1);
  hipLaunchKernelGGL(k_thermo_part, dim3(nb), dim3(FIX_BLOCK), 0, st, A, groupbit, mvv2e, part);
  hipLaunchKernelGGL(k_thermo_final, dim3(1), dim3(64), 0, st, part, nb, out);
  return hipGetLastError();
}

}  // namespace ucg
