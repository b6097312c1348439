// tools/graph_gap.hip -- does replaying a chain of dependent launches as a hipGraph shorten the gap between them?
//
// A step of the engine is two dependent launches (node_kernel, then the link update); at 1024 replicas the gaps between launches are
// ~1.2 us of a 38 us step, at 256 replicas of a small network a larger share.  The chain here: 2 x STEPS kernels, each reading what the one
// before wrote (so they cannot overlap), of a duration set by the grid (a stream over n doubles), launched (a) one by one on a
// stream, (b) captured once into a graph and replayed.  Reported: wall time per launch with the device kept busy, and the kernels'
// own durations from their dispatch timestamps (direct launches only), so that gap = wall per launch - kernel duration.
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/graph_gap tools/graph_gap.hip && /tmp/graph_gap
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void hop(const double* in, double* out, size_t n, int t) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i] + (double)t;
}

int main() {
  CK(hipSetDevice(0));
  const int STEPS = 200;   // 400 launches per chain
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  printf("%12s %14s | %22s %22s | %18s\n", "doubles", "kernel us", "stream: us per launch", "graph: us per launch", "graph build ms");
  for (size_t n = (size_t)1 << 14; n <= (size_t)1 << 24; n <<= 2) {
    double *a, *b;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8));
    CK(hipMemset(b, 0, n * 8));
    const unsigned grid = (unsigned)((n + 255) / 256);
    // kernel duration from dispatch timestamps
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float kern = 0;
    for (int rep = 0; rep < 20; ++rep) {
      hipExtLaunchKernelGGL(hop, dim3(grid), dim3(256), 0, st, e0, e1, 0, (const double*)a, b, n, rep);
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 10) kern += ms / 10;
    }
    auto chain = [&](hipStream_t s) {
      for (int t = 0; t < STEPS; ++t) {
        hipLaunchKernelGGL(hop, dim3(grid), dim3(256), 0, s, (const double*)a, b, n, t);
        hipLaunchKernelGGL(hop, dim3(grid), dim3(256), 0, s, (const double*)b, a, n, t);
      }
    };
    double best_stream = 1e9, best_graph = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::steady_clock::now();
      chain(st);
      CK(hipStreamSynchronize(st));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * STEPS);
      if (us < best_stream) best_stream = us;
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    auto b0 = std::chrono::steady_clock::now();
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (2 * STEPS);
      if (us < best_graph) best_graph = us;
    }
    printf("%12zu %14.2f | %22.2f %22.2f | %18.2f\n", n, kern * 1e3, best_stream, best_graph, build_ms);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipFree(a));
    CK(hipFree(b));
  }
  printf("(400 dependent launches per chain; best of 5 chains; kernel us = dispatch timestamps of single launches)\n");
  return 0;
}
