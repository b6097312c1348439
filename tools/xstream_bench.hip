// Micro-benchmark behind DESIGN.md's "no second stream" decision: what one step costs when the link update and the turning
// fractions run on two streams tied together with events (A on s0; then B on s0 || C on s1; the next A waits for both),
// against the same three kernels back to back on one stream.  Build: hipcc -O2 --offload-arch=gfx950 -o xstream tools/xstream_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(long long cycles, int* sink) {
  long long t0 = wall_clock64();  // constant 100 MHz
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}
int main() {
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t eA, eC;
  hipEventCreateWithFlags(&eA, hipEventDisableTiming);
  hipEventCreateWithFlags(&eC, hipEventDisableTiming);
  int* sink; hipMalloc(&sink, 4);
  const int N = 2000;
  const long long A = 100 * 25, B = 100 * 12, C = 100 * 10;  // wall_clock64 ticks at 100 MHz: 25 / 12 / 10 us
  auto run = [&](int mode) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < N; ++i) {
      if (mode == 0) {  // one stream, three launches
        spin<<<64, 64, 0, s0>>>(A, sink); spin<<<64, 64, 0, s0>>>(B, sink); spin<<<64, 64, 0, s0>>>(C, sink);
      } else {          // A; then B on s0 and C on s1; next A waits for C
        spin<<<64, 64, 0, s0>>>(A, sink);
        hipEventRecord(eA, s0);
        hipStreamWaitEvent(s1, eA, 0);
        spin<<<64, 64, 0, s1>>>(C, sink);
        hipEventRecord(eC, s1);
        spin<<<64, 64, 0, s0>>>(B, sink);
        hipStreamWaitEvent(s0, eC, 0);
      }
    }
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
    printf("mode %d: %.2f us per step (kernels A=25 B=12 C=10 us; serial sum 47, overlapped ideal 37)\n", mode, us);
  };
  run(0); run(0); run(1); run(1);
  return 0;
}
