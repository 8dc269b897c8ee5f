// lds_knot_read.hip -- what do the spline-knot reads of the UCG gather kernel cost on the LDS of MI355X, and does another
// read width or record stride lower it?  One 1024-lane workgroup per CU (the gather kernel's shape); every lane picks a
// pseudo-random knot per iteration and reads the 12 x 16 bytes the three tables need at knots it and it+1
// (lammps-ucg-dev_amd/csrc/ucg_pair_dev.h: knot_eval_fast), as
//   MODE 0: 12 ds_read_b128, record stride 7 slots of 16 B (the kernel's layout: odd slot stride)
//   MODE 1: 12 ds_read_b128, record stride 6 slots (no padding)
//   MODE 2: 24 ds_read_b64, record stride 13 doubles (odd 8-byte stride)
//   MODE 3: 24 ds_read_b64, record stride 14 doubles (the kernel's layout read as doubles)
//   MODE 4: MODE 0 with lanes 2l, 2l+1 reading the SAME knot (two lanes sharing a pair: broadcast)
//   MODE 5: MODE 0 with knots drawn from a window of 64 around a per-wave centre (lanes of a wave at similar r)
// Reports LDS time per 16 bytes read per lane (one b128 or two b64), in ns per wave-instruction-equivalent per CU.
// build: hipcc -O3 --offload-arch=gfx950 lds_knot_read.hip -o lds_knot_read
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hash(unsigned x)
{
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

constexpr int NKNOT = 1024;

template <int MODE>
__global__ __launch_bounds__(1024) void k_read(int iters, double *out)
{
  extern __shared__ double tab[];
  const int stride_d = (MODE == 1) ? 12 : (MODE == 2) ? 13 : 14;  // doubles per knot record
  for (int t = threadIdx.x; t < NKNOT * stride_d + 32; t += blockDim.x) tab[t] = (double) (t & 1023) * 1e-3;
  __syncthreads();
  unsigned h = hash(blockIdx.x * 1024u + threadIdx.x);
  const unsigned wave_centre = hash(blockIdx.x * 16u + (threadIdx.x >> 6)) % (NKNOT - 130) + 65;
  double sink = 0.0;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    unsigned r = h >> 8;
    if (MODE == 4) r = __shfl(r, (threadIdx.x & 63) & ~1, 64);
    int k = (int) (r % (unsigned) (NKNOT - 1));
    if (MODE == 5) k = (int) wave_centre - 32 + (int) (r & 63);
    if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
      const double2 *rec = reinterpret_cast<const double2 *>(tab) + k * (stride_d / 2);
      const int st = stride_d / 2;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const double2 a = rec[2 * t], b = rec[2 * t + 1], c = rec[st + 2 * t], d = rec[st + 2 * t + 1];
        sink += a.x * b.y + c.x * d.y + a.y * c.y + b.x * d.x;
      }
    } else {
      const double *rec = tab + k * stride_d;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const double a0 = rec[4 * t], a1 = rec[4 * t + 1], b0 = rec[4 * t + 2], b1 = rec[4 * t + 3];
        const double c0 = rec[stride_d + 4 * t], c1 = rec[stride_d + 4 * t + 1], d0 = rec[stride_d + 4 * t + 2], d1 = rec[stride_d + 4 * t + 3];
        sink += a0 * b1 + c0 * d1 + a1 * c1 + b0 * d0;
      }
    }
  }
  if (sink == 0.123456789) out[0] = sink;
}

template <int MODE>
void run(const char *name, int iters, double *out)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t lds = (size_t) (NKNOT * 14 + 32) * 8;
  CHECK(hipFuncSetAttribute((const void *) k_read<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k_read<MODE>), dim3(256), dim3(1024), lds, 0, iters, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_read<MODE>), dim3(256), dim3(1024), lds, 0, iters, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double units = 16.0 * iters * 12;  // 16-byte reads per lane, as wave-instructions per CU
  printf("%-58s %8.1f us  %6.2f ns per 16 B wave-read per CU  (%5.1f ns per pair)\n", name, ms * 1e3, ms * 1e6 / units,
         ms * 1e6 / (16.0 * iters));
}

int main()
{
  double *out;
  CHECK(hipMalloc(&out, 64));
  const int iters = 2000;
  run<0>("12 x ds_read_b128, stride 7 slots (kernel layout)", iters, out);
  run<1>("12 x ds_read_b128, stride 6 slots (no padding)", iters, out);
  run<2>("24 x ds_read_b64, stride 13 doubles", iters, out);
  run<3>("24 x ds_read_b64, stride 14 doubles", iters, out);
  run<4>("12 x ds_read_b128, stride 7, lane pairs share a knot", iters, out);
  run<5>("12 x ds_read_b128, stride 7, wave's knots within 64", iters, out);
  return 0;
}
