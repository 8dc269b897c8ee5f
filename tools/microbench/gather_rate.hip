// microbenchmark: cost of per-lane gathers on gfx950 (what bounds k_pair_gather?)
// build: hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: one dwordx4 per entry (16 B of a 32-B record)
// MODE 1: two dwordx4 per entry (whole 32-B record)
// MODE 2: two dwordx4 + one dword from a second array (what the pair kernel does)
// MODE 3: one dword only
template <int MODE>
__global__ __launch_bounds__(1024) void k(const int *idx, const double4 *rec, const int *meta, int nent, int iters, double *out)
{
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int it = 0; it < iters; it++) {
    const int e = (int) (((long long) it * total + lane) % nent);
    const int m = idx[e];
    if (MODE == 0) { const double2 a = reinterpret_cast<const double2 *>(rec)[2 * (size_t) m]; acc += a.x + a.y; }
    if (MODE == 1 || MODE == 2) { const double4 a = rec[m]; acc += a.x + a.y + a.z + a.w; }
    if (MODE == 2 || MODE == 3) acc += meta[m];
  }
  if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char **argv)
{
  const int nrec = 1200000;             // beads + ghosts
  const int nent = 64 * 1024 * 1024;    // index stream
  const int pattern = argc > 1 ? atoi(argv[1]) : 0;   // 0: local-random (within +-2000), 1: fully random, 2: consecutive
  std::vector<int> h(nent);
  unsigned long long s = 88172645463325252ull;
  for (int e = 0; e < nent; e++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const int lane = e % (256 * 16 * 64);
    int base = (int) (((long long) lane * nrec) / (256 * 16 * 64));
    int m;
    if (pattern == 0) m = base + (int) (s % 4000) - 2000;
    else if (pattern == 1) m = (int) (s % nrec);
    else m = (base + (e / (256 * 16 * 64))) % nrec;
    if (m < 0) m += nrec;
    if (m >= nrec) m -= nrec;
    h[e] = m;
  }
  int *d_idx, *d_meta; double4 *d_rec; double *d_out;
  CK(hipMalloc(&d_idx, (size_t) nent * 4)); CK(hipMalloc(&d_meta, (size_t) nrec * 4));
  CK(hipMalloc(&d_rec, (size_t) nrec * 32)); CK(hipMalloc(&d_out, 64));
  CK(hipMemcpy(d_idx, h.data(), (size_t) nent * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_rec, 0, (size_t) nrec * 32)); CK(hipMemset(d_meta, 0, (size_t) nrec * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256, block = 1024, iters = 256;
  auto run = [&](int mode) {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) k<0><<<grid, block>>>(d_idx, d_rec, d_meta, nent, iters, d_out);
      if (mode == 1) k<1><<<grid, block>>>(d_idx, d_rec, d_meta, nent, iters, d_out);
      if (mode == 2) k<2><<<grid, block>>>(d_idx, d_rec, d_meta, nent, iters, d_out);
      if (mode == 3) k<3><<<grid, block>>>(d_idx, d_rec, d_meta, nent, iters, d_out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) {
        const double waveiters = (double) grid * block / 64 * iters;
        printf("pattern %d mode %d: %.3f ms, %.1f ns per wave-iteration per CU-slot => %.0f cycles@2.1GHz per wave-iter per CU\n",
               pattern, mode, ms, ms * 1e6 / (waveiters / 256), ms * 1e-3 * 2.1e9 / (waveiters / 256));
      }
    }
  };
  for (int mode = 0; mode < 4; mode++) run(mode);
  return 0;
}
