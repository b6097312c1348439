// Issue cost of the vector instructions the PedN kernels lean on, on this part: N dependent-free copies of one instruction per
// wave, 8 waves per SIMD on every CU, cycles per wave-instruction = elapsed SIMD cycles / (waves per SIMD * instructions).
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 512, UNROLL = 8;

template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(double* out, uint32_t seed) {
  uint32_t a[UNROLL];
  uint64_t q[UNROLL];
  float f[UNROLL];
  double d[UNROLL];
  for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * 7 + i; q[i] = a[i]; f[i] = 1.0f + 1e-3f * a[i]; d[i] = 1.0 + 1e-3 * a[i]; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(0xD2511F53u) : "vcc");
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(1.0001f));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(1.0001));
      if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      if (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xCD9E8D57u));
      if (OP == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0x9E3779B9u));
      if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0001));
      if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0001));
      if (OP == 10) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[i]) : "v"(1.0001f) : "vcc");
      if (OP == 11) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
      if (OP == 12) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));
      if (OP == 13) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(1.0001));
      if (OP == 15) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x8D57u));
    }
  }
  double s = 0;
  for (int i = 0; i < UNROLL; ++i) s += (double)a[i] + (double)q[i] + f[i] + d[i];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
int run(const char* name, double* out, double clock_mhz, int cus) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = cus * 4;   // 4 blocks of 512 threads per CU = 8 waves per SIMD
  rate_kernel<OP><<<blocks, 512>>>(out, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  rate_kernel<OP><<<blocks, 512>>>(out, 2u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double cycles = ms * 1e-3 * clock_mhz * 1e6, per = cycles / (8.0 * ITER * UNROLL);
  printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction (8 waves per SIMD)\n", name, ms, per);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const double mhz = p.clockRate / 1000.0;
  printf("%s, %d CUs, %.0f MHz (nominal; cycles below assume it)\n", p.name, p.multiProcessorCount, mhz);
  double* out;
  CHECK(hipMalloc(&out, 8));
  run<1>("v_fma_f32", out, mhz, p.multiProcessorCount);
  run<14>("v_pk_fma_f32", out, mhz, p.multiProcessorCount);
  run<7>("v_xor_b32", out, mhz, p.multiProcessorCount);
  run<15>("v_mul_u32_u24", out, mhz, p.multiProcessorCount);
  run<0>("v_mad_u64_u32", out, mhz, p.multiProcessorCount);
  run<6>("v_mul_lo_u32", out, mhz, p.multiProcessorCount);
  run<2>("v_fma_f64", out, mhz, p.multiProcessorCount);
  run<8>("v_mul_f64", out, mhz, p.multiProcessorCount);
  run<9>("v_add_f64", out, mhz, p.multiProcessorCount);
  run<12>("v_floor_f64", out, mhz, p.multiProcessorCount);
  run<4>("v_cvt_f32_f64", out, mhz, p.multiProcessorCount);
  run<13>("v_cvt_f64_u32", out, mhz, p.multiProcessorCount);
  run<10>("v_div_scale_f32", out, mhz, p.multiProcessorCount);
  run<3>("v_rcp_f32", out, mhz, p.multiProcessorCount);
  run<5>("v_rcp_f64", out, mhz, p.multiProcessorCount);
  run<11>("v_sqrt_f64", out, mhz, p.multiProcessorCount);
  return 0;
}
