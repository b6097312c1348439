// tools/mall_stream.hip -- what a coalesced stream sustains on MI355X as a function of its footprint and its read : write mix.
//
// The step kernels are judged against "what a pure stream sustains on this part".  Two things about that yardstick were not
// measured before: (1) a step of melbourne x 1024 touches ~170 MB, which FITS the 256 MB Infinity Cache (MALL), so the
// yardstick must be a stream over a footprint of that size, launched repeatedly (warm), not one over 1.5 GiB; (2) node_kernel
// reads 1.4 bytes per byte written, link_kernel 1 : 1 -- a 2 : 1 triad flatters both.
//
//   for footprint in 24 MB ... 3 GB:  for mix in  2R:1W (out = a + b), 1R:1W (out = a), 1R (sum), 1W (fill):
//       launch 12 times over the same buffers, report the median of the last 8 launches in TB/s (bytes moved / duration)
// 16-byte accesses per lane (the shape of link_kernel), grid = elements / 2 / 256 workgroups of 256 threads.
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/mall_stream tools/mall_stream.hip && /tmp/mall_stream
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void triad(const double2* a, const double2* b, double2* out, size_t n2) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) { double2 x = a[i], y = b[i]; out[i] = make_double2(x.x + y.x, x.y + y.y); }
}
// 8-byte accesses per lane (the shape of node_kernel's f64 rows): contiguous, and in 512-byte segments (one wave instruction) visited
// in a scrambled order (n a power of two)
__global__ __launch_bounds__(256) void triad8(const double* a, const double* b, double* out, size_t n, int scramble) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (scramble) {
    const size_t nch = n >> 6, c = i >> 6;
    const size_t pc = (c * 2654435761ull + 12345ull) & (nch - 1);   // odd multiplier: a permutation of the segments
    i = (pc << 6) | (i & 63);
  }
  out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void copy(const double2* a, double2* out, size_t n2) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) out[i] = a[i];
}
__global__ __launch_bounds__(256) void rsum(const double2* a, double* sink, size_t n2) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) { double2 x = a[i]; if (x.x + x.y == 123456.789) *sink = x.x; }   // never true: the load must still happen
}
__global__ __launch_bounds__(256) void fill(double2* out, size_t n2) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) out[i] = make_double2(1.0, 2.0);
}

int main() {
  CK(hipSetDevice(0));
  const size_t max_n = (size_t)1 << 27;   // doubles per array: 1 GiB
  double *a, *b, *o, *sink;
  CK(hipMalloc(&a, max_n * 8));
  CK(hipMalloc(&b, max_n * 8));
  CK(hipMalloc(&o, max_n * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 0, max_n * 8));
  CK(hipMemset(b, 0, max_n * 8));
  CK(hipMemset(o, 0, max_n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%12s | %28s %28s %22s %22s\n", "doubles/arr", "2R:1W  footprint MB   TB/s", "1R:1W  footprint MB   TB/s", "1R  MB   TB/s", "1W  MB   TB/s");
  for (size_t n = (size_t)1 << 20; n <= max_n; n <<= 1) {
    const size_t n2 = n / 2;
    const unsigned grid = (unsigned)((n2 + 255) / 256);
    double res[4], mb[4] = {3.0 * n * 8 / 1e6, 2.0 * n * 8 / 1e6, 1.0 * n * 8 / 1e6, 1.0 * n * 8 / 1e6};
    for (int mix = 0; mix < 4; ++mix) {
      std::vector<float> ms;
      for (int rep = 0; rep < 12; ++rep) {
        // start / stop events of the dispatch itself (what rocprofv3 reports), not an enqueue-to-completion bracket
        if (mix == 0) hipExtLaunchKernelGGL(triad, dim3(grid), dim3(256), 0, 0, e0, e1, 0, (const double2*)a, (const double2*)b, (double2*)o, n2);
        if (mix == 1) hipExtLaunchKernelGGL(copy, dim3(grid), dim3(256), 0, 0, e0, e1, 0, (const double2*)a, (double2*)o, n2);
        if (mix == 2) hipExtLaunchKernelGGL(rsum, dim3(grid), dim3(256), 0, 0, e0, e1, 0, (const double2*)a, sink, n2);
        if (mix == 3) hipExtLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, e0, e1, 0, (double2*)o, n2);
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 4) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      res[mix] = mb[mix] * 1e6 / (ms[ms.size() / 2] * 1e-3) / 1e12;
    }
    // the 2R:1W mix once more with 8-byte lanes: contiguous, and as scattered 512-byte segments
    double r8[2];
    for (int sc = 0; sc < 2; ++sc) {
      std::vector<float> ms;
      for (int rep = 0; rep < 12; ++rep) {
        hipExtLaunchKernelGGL(triad8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, e0, e1, 0, (const double*)a, (const double*)b, o, n, sc);
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 4) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      r8[sc] = mb[0] * 1e6 / (ms[ms.size() / 2] * 1e-3) / 1e12;
    }
    printf("%12zu | %20.0f %7.2f %20.0f %7.2f %14.0f %7.2f %14.0f %7.2f | 2R:1W 8-byte lanes %5.2f, in scattered 512 B segments %5.2f\n", n, mb[0], res[0], mb[1],
           res[1], mb[2], res[2], mb[3], res[3], r8[0], r8[1]);
  }
  printf("(durations are the dispatches' own start / stop timestamps)\n");
  return 0;
}
