// tools/group_barrier.hip -- what does a barrier among the workgroups of ONE 64-replica group cost on MI355X?
//
// Background (DESIGN.md "persistent step"): replicas never interact, so a persistent step kernel would
// not need a grid-wide barrier between its phases (node -> link/turn -> node ...): only the B workgroups that work on the same
// 64 replicas have to meet.  This microbenchmark prices that meeting point, with the data hand-off it exists for:
//
//   grid = G groups x B workgroups of 256 threads, all resident; K rounds.  In every round each wave reads the 64 doubles its
//   ring neighbour (another workgroup of the group) wrote in the previous round, adds one, writes its own 64 doubles, then the
//   group meets: thread 0 of every workgroup adds to the group's counter and spins until it reads B * round.
//   After K rounds every value must equal K -- a barrier or a hand-off that lets a stale value through is a wrong answer.
//
// Variants (argv[1], default all):
//   fence   plain loads / stores; release fence (agent scope) before the add, acquire fence after the spin: what the memory model
//           asks for between workgroups (on gfx950: L2 write-back + L1/L2 invalidate, the L2s of the 8 XCDs are not coherent)
//   sc1     hand-off data stored write-through and loaded past the caches (relaxed agent-scope atomics = sc1 accesses), the
//           wave drains its stores (s_waitcnt vmcnt(0)) before the add; no cache maintenance
//   nosync  no meeting point at all (the floor: K rounds of load + store)
// Placement: "xcd" -- linear workgroup id = b * Gpad + g with Gpad a multiple of 8, so the B workgroups of a group run on ONE
// XCD (workgroups go round-robin over the 8 XCDs by linear id); "spread" -- id = g * B + b: a group's workgroups on all XCDs.
// A spin is bounded (2^22 polls): a lost arrival sets an error flag and the workgroup leaves; nothing can hang.
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/group_barrier tools/group_barrier.hip && /tmp/group_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { FENCE = 0, SC1 = 1, NOSYNC = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(double* data, int* counters, int* error, int G, int Gpad, int B, int K, int xcd_placed) {
  int g, b;
  if (xcd_placed) { g = (int)(blockIdx.x % (unsigned)Gpad); b = (int)(blockIdx.x / (unsigned)Gpad); }
  else { g = (int)(blockIdx.x / (unsigned)B); b = (int)(blockIdx.x % (unsigned)B); }
  if (g >= G) return;   // padding workgroups of the xcd placement
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // slot (g, b, wave): 64 doubles, double-buffered by round parity
  auto slot = [&](int bb, int par) { return data + ((((size_t)par * G + g) * B + bb) * 4 + wave) * 64 + lane; };
  const int nb = (b + 1) % B;
  double acc = 0.0;
  for (int k = 1; k <= K; ++k) {
    double* src = slot(nb, (k - 1) & 1);
    double* dst = slot(b, k & 1);
    double x;
    if (MODE == SC1) x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else x = *src;
    x += 1.0;
    acc = x;
    if (MODE == SC1) {
      __hip_atomic_store(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      *dst = x;
    }
    if (MODE == NOSYNC) continue;
    if (MODE == FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&counters[g * 64], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 256-byte line per group
      const int want = B * k;
      int polls = 0;
      while (__hip_atomic_load(&counters[g * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(8);    // ~0.2 us between polls: the pollers must not saturate the path the arrivals take
        if (++polls > (1 << 22)) { atomicExch(error, 1); break; }
      }
    }
    __syncthreads();
    if (MODE == FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (MODE != NOSYNC && acc != (double)K) atomicExch(error, 2);
}

template <int MODE>
static double run(double* data, int* counters, int* error, int G, int B, int K, int xcd, size_t data_bytes, int* err_out) {
  const int Gpad = (G + 7) / 8 * 8;
  const unsigned grid = xcd ? (unsigned)(Gpad * B) : (unsigned)(G * B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(data, 0, data_bytes));
    CK(hipMemset(counters, 0, sizeof(int) * (size_t)G * 64));
    CK(hipMemset(error, 0, sizeof(int)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rounds_kernel<MODE>, dim3(grid), dim3(256), 0, 0, data, counters, error, G, Gpad, B, K, xcd);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipMemcpy(err_out, error, sizeof(int), hipMemcpyDeviceToHost));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best * 1e3 / K;   // us per round
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "all";
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rounds_kernel<FENCE>, 256, 0));
  const int capacity = per_cu * prop.multiProcessorCount;
  printf("%s: %d CUs, %d resident workgroups of 256 threads per CU -> %d co-resident workgroups\n", prop.name, prop.multiProcessorCount, per_cu, capacity);
  const int K = 400;
  // (B workgroups per group, G groups): the shapes of the step -- nine_intersections x 256 (4 groups x ~8), 45_intersections x
  // 2048 (32 x ~32), melbourne / delft x 1024 (16 x ~190, more than half the machine per XCD pair), and a few in between
  const int shapes[][2] = {{2, 4}, {4, 4}, {8, 4}, {8, 32}, {16, 32}, {32, 32}, {64, 16}, {120, 16}};
  printf("%-8s %-7s %4s %4s %8s | us per round (K = %d rounds; every wave: 512 B in, 512 B out)\n", "variant", "placed", "B", "G", "wgs", K);
  for (auto& sh : shapes) {
    const int B = sh[0], G = sh[1], Gpad = (G + 7) / 8 * 8;
    if (Gpad * B > capacity / 2) { printf("shape B=%d G=%d skipped: %d workgroups are more than half of what is resident\n", B, G, Gpad * B); continue; }
    const size_t data_bytes = (size_t)2 * G * B * 4 * 64 * sizeof(double);
    double* data;
    int *counters, *error;
    CK(hipMalloc(&data, data_bytes));
    CK(hipMalloc(&counters, sizeof(int) * (size_t)G * 64));
    CK(hipMalloc(&error, sizeof(int)));
    for (int xcd = 1; xcd >= 0; --xcd) {
      struct { const char* name; double us; int err; } r[3] = {{"fence", 0, 0}, {"sc1", 0, 0}, {"nosync", 0, 0}};
      if (!strcmp(only, "all") || !strcmp(only, "fence")) r[0].us = run<FENCE>(data, counters, error, G, B, K, xcd, data_bytes, &r[0].err);
      if (!strcmp(only, "all") || !strcmp(only, "sc1")) r[1].us = run<SC1>(data, counters, error, G, B, K, xcd, data_bytes, &r[1].err);
      if (!strcmp(only, "all") || !strcmp(only, "nosync")) r[2].us = run<NOSYNC>(data, counters, error, G, B, K, xcd, data_bytes, &r[2].err);
      for (auto& x : r)
        if (x.us > 0)
          printf("%-8s %-7s %4d %4d %8d | %7.2f%s\n", x.name, xcd ? "xcd" : "spread", B, G, (xcd ? Gpad : G) * B, x.us,
                 x.err == 1 ? "   SPIN LIMIT HIT" : x.err == 2 ? "   STALE VALUE SEEN" : "");
    }
    hipFree(data);
    hipFree(counters);
    hipFree(error);
  }
  return 0;
}
