// ucg_ranmars.hip -- exact, block-parallel RANMAR for the per-bead draws of
// fix ucgld/langevin and fix ucgstate mc (gfx950).
//
// The reference draws ONE RanMars::uniform() per owned bead per step from a
// sequential stream (UCG/fix_ucgld_langevin.cpp:85,280; UCG/fix_ucgstate.cpp:62,117):
// a million dependent draws per step on the host would dwarf the whole GPU step.
// RanMars is upstream LAMMPS (absent from the reference tree); its published
// algorithm (Marsaglia, Zaman & Tsang 1990) is
//     X_n = X_{n-97} - X_{n-33}  (mod 2^24)         lagged Fibonacci, 24-bit fractions
//     c_n = c_{n-1} - 7654321    (mod 16777213)      arithmetic sequence
//     uniform_n = (X_n - c_n) mod 2^24, scaled by 2^-24
// Every value is an exact multiple of 2^-24, so integer arithmetic mod 2^24
// reproduces the double-precision original bit for bit.
//
// Parallel form: the lag recurrence has characteristic polynomial
// P(z) = z^97 + z^64 - 1 over Z/2^24, so X_{t+J} = sum_k g_k X_{t+k} with
// g = z^J mod P.  The host precomputes g for J = p*CHUNK; block p jumps the
// 97-value history to its chunk start with one 97x97 multiply-accumulate, then
// generates CHUNK values 33 at a time (X_n only needs values >= 33 back);
// c_n has the closed form (c_0 - n*cd) mod cm.
#include <vector>

#include "ucg_launch.h"

namespace ucg {

namespace {

constexpr unsigned int M24 = 0xFFFFFFu;
constexpr unsigned long long CM = 16777213ull, CD = 7654321ull, C0 = 362436ull;

__host__ __device__ inline unsigned int c_of_call(unsigned long long n)
{
  // c after the n-th uniform() call: (C0 - n*CD) mod CM, kept in [0, CM)
  const unsigned long long r = (n % CM) * (CM - CD) % CM;
  return (unsigned int) ((C0 + r) % CM);
}

constexpr int RM_BLOCK = 64;

__global__ __launch_bounds__(RM_BLOCK) void k_ranmars(const unsigned int *hist_in, unsigned int *hist_out,
                                                     const unsigned int *jump, const long long count,
                                                     const int n, unsigned int *out)
{
  __shared__ unsigned int W[193];
  __shared__ unsigned int G[97];
  __shared__ unsigned int Y[97 + RANMARS_CHUNK];
  const int p = blockIdx.x;
  const int tid = threadIdx.x;
  const int base = p * RANMARS_CHUNK;
  const int nthis = min(RANMARS_CHUNK, n - base);

  for (int t = tid; t < 97; t += RM_BLOCK) {
    W[t] = hist_in[t];
    G[t] = jump[(size_t) p * 97 + t];
  }
  __syncthreads();
  // extend the window by 96 values, 33 per round
  for (int i0 = 0; i0 < 96; i0 += 33) {
    const int i = i0 + tid;
    if (tid < 33 && i < 96) W[97 + i] = (W[i] - W[i + 64]) & M24;
    __syncthreads();
  }
  // jump: history at this block's chunk start
  for (int t = tid; t < 97; t += RM_BLOCK) {
    unsigned int acc = 0;
    for (int k = 0; k < 97; k++) acc += G[k] * W[t + k];
    Y[t] = acc & M24;
  }
  __syncthreads();
  // generate
  for (int i0 = 0; i0 < nthis; i0 += 33) {
    const int i = i0 + tid;
    if (tid < 33 && i < nthis) Y[97 + i] = (Y[i] - Y[i + 64]) & M24;
    __syncthreads();
  }
  const unsigned long long call0 = (unsigned long long) count + (unsigned long long) base;
  for (int i = tid; i < nthis; i += RM_BLOCK) {
    const unsigned int c = c_of_call(call0 + (unsigned long long) i + 1ull);
    out[base + i] = (Y[97 + i] - c) & M24;
  }
  if (p == gridDim.x - 1)
    for (int t = tid; t < 97; t += RM_BLOCK) hist_out[t] = Y[nthis + t];
}

void poly_mulmod(const unsigned int *a, const unsigned int *b, unsigned int *c)
{
  unsigned int tmp[193];
  for (int i = 0; i < 193; i++) tmp[i] = 0;
  for (int i = 0; i < 97; i++) {
    if (!a[i]) continue;
    for (int j = 0; j < 97; j++) tmp[i + j] += a[i] * b[j];
  }
  for (int d = 192; d >= 97; d--) {
    const unsigned int coef = tmp[d];
    tmp[d] = 0;
    tmp[d - 97] += coef;       // z^97 = 1 - z^64
    tmp[d - 97 + 64] -= coef;
  }
  for (int i = 0; i < 97; i++) c[i] = tmp[i] & M24;
}

}  // namespace

void ranmars_seed_host(int seed, unsigned int *hist97, long long *count)
{
  // LAMMPS RanMars constructor (upstream, absent): seeds -> 97 24-bit fractions
  int ij = (seed - 1) / 30082;
  int kl = (seed - 1) - 30082 * ij;
  int i = (ij / 177) % 177 + 2;
  int j = ij % 177 + 2;
  int k = (kl / 169) % 178 + 1;
  int l = kl % 169;
  unsigned int U[98];
  for (int ii = 1; ii <= 97; ii++) {
    unsigned int s = 0;
    for (int jj = 1; jj <= 24; jj++) {
      const int m = ((i * j) % 179) * k % 179;
      i = j;
      j = k;
      k = m;
      l = (53 * l + 1) % 169;
      if ((l * m) % 64 >= 32) s |= 1u << (24 - jj);
    }
    U[ii] = s;
  }
  // oldest first: X_{-96+t} = U[97-t]
  unsigned int h[98];
  for (int t = 0; t < 97; t++) h[t] = U[97 - t];
  // the constructor's warm-up uniform(): X_1 = X_{-96} - X_{-32}
  h[97] = (h[0] - h[64]) & M24;
  for (int t = 0; t < 97; t++) hist97[t] = h[t + 1];
  *count = 1;
}

void ranmars_jump_host(int nchunks, unsigned int *out)
{
  unsigned int zc[97], acc[97], base[97];
  // z^CHUNK by square-and-multiply
  for (int i = 0; i < 97; i++) { acc[i] = 0; base[i] = 0; }
  acc[0] = 1;
  base[1] = 1;
  int e = RANMARS_CHUNK;
  while (e) {
    if (e & 1) poly_mulmod(acc, base, acc);
    poly_mulmod(base, base, base);
    e >>= 1;
  }
  for (int i = 0; i < 97; i++) zc[i] = acc[i];
  unsigned int cur[97];
  for (int i = 0; i < 97; i++) cur[i] = 0;
  cur[0] = 1;
  for (int p = 0; p < nchunks; p++) {
    for (int i = 0; i < 97; i++) out[(size_t) p * 97 + i] = cur[i];
    poly_mulmod(cur, zc, cur);
  }
}

hipError_t launch_ranmars(RanMarsDev &R, int n, unsigned int *out, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const int nchunks = (n + RANMARS_CHUNK - 1) / RANMARS_CHUNK;
  if (nchunks > R.nchunks_max) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ranmars, dim3(nchunks), dim3(RM_BLOCK), 0, st, R.hist[R.cur], R.hist[R.cur ^ 1], R.jump,
                     R.count, n, out);
  R.cur ^= 1;
  R.count += n;
  return hipGetLastError();
}

}  // namespace ucg
