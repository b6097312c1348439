// Does `buffer_inv sc0` (workgroup scope) drop a CU's vector L1 on gfx950, and what do the scopes cost?  Two workgroups on ONE XCD
// (indices 0 and 8): A caches a word in its L1 with a plain load, B overwrites it, A is told through an L2 atomic, then A re-reads
// (1) plainly, (2) after buffer_inv sc0, (3) after buffer_inv sc1.  Prints what A saw and the ticks of each invalidate.
//   hipcc --offload-arch=gfx950 -O3 -o l1_inv_probe tools/l1_inv_probe.hip && ./l1_inv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned plain_load(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__global__ void probe(unsigned* x, unsigned* flag, unsigned* out, unsigned long long* ticks, int mode) {
  if (blockIdx.x != 0 && blockIdx.x != 8) return;
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[4] = xcc & 15u;
    out[0] = plain_load(x);                                   // 0, now in A's L1
    __hip_atomic_store(&flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int n = 0;
    while (__hip_atomic_load(&flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++n < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    out[1] = plain_load(x);                                   // stale if the L1 kept the line
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 1) asm volatile("buffer_inv sc0" ::: "memory");
    if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[2] = plain_load(x);
    ticks[0] = t1 - t0;
  } else {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[5] = xcc & 15u;
    int n = 0;
    while (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++n < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    *x = 7u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // acknowledged by the L2
    __hip_atomic_store(&flag[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  unsigned *x, *flag, *out;
  unsigned long long* ticks;
  hipMalloc(&x, 256); hipMalloc(&flag, 256); hipMalloc(&out, 256); hipMalloc(&ticks, 64);
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(x, 0, 256); hipMemset(flag, 0, 256); hipMemset(out, 0xff, 256);
      hipLaunchKernelGGL(probe, dim3(16), dim3(64), 0, 0, x, flag, out, ticks, mode);
      hipDeviceSynchronize();
      unsigned h[8]; unsigned long long t;
      hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("%s: first %u, after B's store (plain) %u, after the invalidate %u; %llu ticks of s_memtime (100 MHz); XCC of A %u, of B %u\n",
             mode == 0 ? "no invalidate " : mode == 1 ? "buffer_inv sc0" : "buffer_inv sc1", h[0], h[1], h[2], t, h[4], h[5]);
    }
  return 0;
}
