// tools/node_pattern.hip -- is node_kernel's duration set by its memory access pattern alone?
//
// A stripped copy of node_kernel's traffic for a melbourne-sized batch (980 slots x 1024 replicas, grid (16 replica groups, 123 bins)
// x 512 threads, wave = (slot, 64 replicas), 8 waves per SIMD): per wave one 312-byte scalar record, a batch of 8 independent row
// loads (3 x f32, 5 x f64) at t-1 / t-2 / t+1-tau_sw, ONE data-dependent look-back load (cumulative_inflow[t+1-tau], tau per link +- 1),
// two block barriers with an LDS exchange, six f64 row stores -- 60 B read + 48 B written per lane like the real kernel -- and
// almost no arithmetic.  Rows live in [32][columns][RS] rings so that the working set (~110 MB) stays in the Infinity Cache as in
// real stepping.  Variants isolate what costs time:
//   full        everything
//   nodep       the look-back address does not depend on loaded data (issued with the batch)
//   nobar       no barriers / LDS
//   norec       no scalar record: link ids computed from the block and wave index
//   loads1      ONE 8-byte load + one store per lane (how fast is a trivially short wave on this grid?)
//   wide        two replicas per lane (16-byte f64 / 8-byte f32 accesses), half the replica groups: 8 x 123 blocks
//   binfast     grid (bins, replica groups): consecutive workgroups are different bins of one replica group (the real kernel: the
//               replica group is the fastest index)
//   pipe2       every block handles TWO bins: the batches of both are issued up front, then the first bin is finished (look-back,
//               barriers, stores) while the second one's loads are in flight -- half the blocks, one generation of waves
//   packed      what if everything a wave reads sat next to each other?  its 9 loads come from ONE contiguous 4.6 KB chunk per
//               (slot, replica group) and its 6 stores go to another: same bytes, same waves, perfect locality
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/node_pattern tools/node_pattern.hip && /tmp/node_pattern
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { int lin, lout, tau_sw, pad; double params[38]; };   // 312 bytes like SlotRec

struct View {
  double *CI, *CO, *S, *R, *IN, *OUT;
  float *N, *ATT;
  const Rec* rec;
  int cols, RS, rows;
};

enum { FULL = 0, NODEP, NOBAR, NOREC, LOADS1, WIDE, BINFAST, PACKED, PIPE2 };

template <typename T>
__device__ __forceinline__ T* rowp(T* base, int row, int col, int cols, int RS, int r0) {
  return base + (((size_t)row * (size_t)cols + (size_t)col) * (size_t)RS + (size_t)r0);
}

template <int MODE>
__global__ __launch_bounds__(512, 8) void pattern(View v, int t) {
  __shared__ double lds[2 * 8 * 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const int RS = v.RS, cols = v.cols, M = v.rows - 1;
  const int r0 = (MODE == BINFAST ? (int)blockIdx.y : (int)blockIdx.x) * 64;
  const int slot = (MODE == BINFAST ? (int)blockIdx.x : (int)blockIdx.y) * 8 + wave;
  int lin, lout, tau_sw;
  double p0 = 1.0;
  if (MODE == NOREC) { lin = slot; lout = (slot * 7 + 3) % cols; tau_sw = 1 + slot % 24; }
  else { const Rec& W = v.rec[slot]; lin = W.lin; lout = W.lout; tau_sw = W.tau_sw; p0 = W.params[5]; }
  if (MODE == LOADS1) {
    const double x = rowp(v.CO, (t - 1) & M, lin, cols, RS, r0)[lane];
    rowp(v.CO, t & M, lin, cols, RS, r0)[lane] = x + p0;
    return;
  }
  const int tp = t - 1;
  if (MODE == PACKED) {
    // chunk of (slot, replica group): [9 pieces of 64 doubles] read from ring row (t-1), [6 pieces] written to ring row t
    const size_t chunk = ((size_t)slot * (size_t)(RS / 64) + blockIdx.x) * 9 * 64;
    const double* src = v.CI + ((size_t)(tp & 1) * (size_t)cols * RS * 9) + chunk;
    double* dst = v.CO + ((size_t)(t & 1) * (size_t)cols * RS * 9) + chunk;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += src[k * 64 + lane];
    acc += (double)reinterpret_cast<const float*>(src + 7 * 64)[lane] + (double)reinterpret_cast<const float*>(src + 7 * 64)[64 + lane];
    const int pick = ((int)acc) & 1;
    acc += src[(8 - pick) * 64 + lane] * p0;     // the dependent piece
    lds[wave * 64 + lane] = acc;
    __syncthreads();
    const double q = lds[((wave + 1) & 7) * 64 + lane];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k * 64 + lane] = acc + q + k;
    return;
  }
  // batch
  const float n_in = rowp(v.N, tp & M, lin, cols, RS, r0)[lane], n_out = rowp(v.N, tp & M, lout, cols, RS, r0)[lane];
  const float att = rowp(v.ATT, tp & M, lin, cols, RS, r0)[lane];
  const double co_in = rowp(v.CO, tp & M, lin, cols, RS, r0)[lane], s_prev = rowp(v.S, (tp - 1) & M, lin, cols, RS, r0)[lane];
  const double co_sw = rowp(v.CO, (tp + 1 - tau_sw) & M, lout, cols, RS, r0)[lane], ci_out = rowp(v.CI, tp & M, lout, cols, RS, r0)[lane];
  const double r_prev = rowp(v.R, (tp - 1) & M, lout, cols, RS, r0)[lane];
  // look-back: tau = a per-link value (2..19) + 0 / 1 from the loaded travel time (an idle link's tau is its free-flow value, the
  // speed noise flips it by one in some replicas), or the same without the data dependence (NODEP)
  const int tau = 2 + slot % 18 + (MODE == NODEP ? (lane * 7 + slot) & 1 : ((int)att) & 1);
  const double ci_look = rowp(v.CI, (tp + 1 - tau) & M, lin, cols, RS, r0)[lane];
  double s_i = (double)(n_in + n_out) + co_in + s_prev + ci_look * p0, r_i = co_sw + ci_out + r_prev;
  rowp(v.S, tp & M, lin, cols, RS, r0)[lane] = s_i;
  rowp(v.R, tp & M, lout, cols, RS, r0)[lane] = r_i;
  double qo = s_i, qi = r_i;
  if (MODE != NOBAR) {
    lds[wave * 64 + lane] = s_i;
    lds[(8 + wave) * 64 + lane] = r_i;
    __syncthreads();
    qo = lds[((wave + 1) & 7) * 64 + lane] + r_i;
    __syncthreads();
    lds[wave * 64 + lane] = qo;
    __syncthreads();
    qi = lds[((wave + 7) & 7) * 64 + lane] + s_i;
  }
  rowp(v.OUT, t & M, lin, cols, RS, r0)[lane] = qo;
  rowp(v.IN, t & M, lout, cols, RS, r0)[lane] = qi;
  rowp(v.CO, t & M, lin, cols, RS, r0)[lane] = co_in + qo;
  rowp(v.CI, t & M, lout, cols, RS, r0)[lane] = ci_out + qi;
}

struct Batch { float n_in, n_out, att; double co_in, s_prev, co_sw, ci_out, r_prev; int lin, lout; double p0; };

__global__ __launch_bounds__(512, 8) void pattern_pipe2(View v, int t, int half_bins) {
  __shared__ double lds[2 * 8 * 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const int RS = v.RS, cols = v.cols, M = v.rows - 1, r0 = (int)blockIdx.x * 64, tp = t - 1;
  Batch b[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = ((int)blockIdx.y + k * half_bins) * 8 + wave;
    const Rec& W = v.rec[slot];
    const int lin = W.lin, lout = W.lout, tau_sw = W.tau_sw;
    b[k].lin = lin; b[k].lout = lout; b[k].p0 = W.params[5];
    b[k].n_in = rowp(v.N, tp & M, lin, cols, RS, r0)[lane]; b[k].n_out = rowp(v.N, tp & M, lout, cols, RS, r0)[lane];
    b[k].att = rowp(v.ATT, tp & M, lin, cols, RS, r0)[lane];
    b[k].co_in = rowp(v.CO, tp & M, lin, cols, RS, r0)[lane]; b[k].s_prev = rowp(v.S, (tp - 1) & M, lin, cols, RS, r0)[lane];
    b[k].co_sw = rowp(v.CO, (tp + 1 - tau_sw) & M, lout, cols, RS, r0)[lane]; b[k].ci_out = rowp(v.CI, tp & M, lout, cols, RS, r0)[lane];
    b[k].r_prev = rowp(v.R, (tp - 1) & M, lout, cols, RS, r0)[lane];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = ((int)blockIdx.y + k * half_bins) * 8 + wave, lin = b[k].lin, lout = b[k].lout;
    const int tau = 2 + slot % 18 + (((int)b[k].att) & 1);
    const double ci_look = rowp(v.CI, (tp + 1 - tau) & M, lin, cols, RS, r0)[lane];
    const double s_i = (double)(b[k].n_in + b[k].n_out) + b[k].co_in + b[k].s_prev + ci_look * b[k].p0, r_i = b[k].co_sw + b[k].ci_out + b[k].r_prev;
    rowp(v.S, tp & M, lin, cols, RS, r0)[lane] = s_i;
    rowp(v.R, tp & M, lout, cols, RS, r0)[lane] = r_i;
    lds[wave * 64 + lane] = s_i;
    lds[(8 + wave) * 64 + lane] = r_i;
    __syncthreads();
    const double qo = lds[((wave + 1) & 7) * 64 + lane] + r_i;
    __syncthreads();
    lds[wave * 64 + lane] = qo;
    __syncthreads();
    const double qi = lds[((wave + 7) & 7) * 64 + lane] + s_i;
    __syncthreads();
    rowp(v.OUT, t & M, lin, cols, RS, r0)[lane] = qo;
    rowp(v.IN, t & M, lout, cols, RS, r0)[lane] = qi;
    rowp(v.CO, t & M, lin, cols, RS, r0)[lane] = b[k].co_in + qo;
    rowp(v.CI, t & M, lout, cols, RS, r0)[lane] = b[k].ci_out + qi;
  }
}

// two replicas per lane: 16-byte f64 and 8-byte f32 accesses, a wave covers 128 replicas
__global__ __launch_bounds__(512, 4) void pattern_wide(View v, int t) {
  __shared__ double2 lds[2 * 8 * 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const int r0 = (int)blockIdx.x * 128, RS = v.RS, cols = v.cols, M = v.rows - 1;
  const int slot = (int)blockIdx.y * 8 + wave;
  const Rec& W = v.rec[slot];
  const int lin = W.lin, lout = W.lout, tau_sw = W.tau_sw;
  const double p0 = W.params[5];
  const int tp = t - 1;
  auto d2 = [&](double* b, int row, int col) { return reinterpret_cast<double2*>(rowp(b, row & M, col, cols, RS, r0))[lane]; };
  auto f2 = [&](float* b, int row, int col) { return reinterpret_cast<float2*>(rowp(b, row & M, col, cols, RS, r0))[lane]; };
  auto st = [&](double* b, int row, int col, double2 x) { reinterpret_cast<double2*>(rowp(b, row & M, col, cols, RS, r0))[lane] = x; };
  const float2 n_in = f2(v.N, tp, lin), n_out = f2(v.N, tp, lout), att = f2(v.ATT, tp, lin);
  const double2 co_in = d2(v.CO, tp, lin), s_prev = d2(v.S, tp - 1, lin), co_sw = d2(v.CO, tp + 1 - tau_sw, lout), ci_out = d2(v.CI, tp, lout),
                r_prev = d2(v.R, tp - 1, lout);
  const int tau0 = 2 + slot % 18 + (((int)att.x) & 1), tau1 = 2 + slot % 18 + (((int)att.y) & 1);
  const double l0 = rowp(v.CI, (tp + 1 - tau0) & M, lin, cols, RS, r0)[2 * lane], l1 = rowp(v.CI, (tp + 1 - tau1) & M, lin, cols, RS, r0)[2 * lane + 1];
  double2 s_i = make_double2((double)(n_in.x + n_out.x) + co_in.x + s_prev.x + l0 * p0, (double)(n_in.y + n_out.y) + co_in.y + s_prev.y + l1 * p0);
  double2 r_i = make_double2(co_sw.x + ci_out.x + r_prev.x, co_sw.y + ci_out.y + r_prev.y);
  st(v.S, tp, lin, s_i);
  st(v.R, tp, lout, r_i);
  lds[wave * 64 + lane] = s_i;
  lds[(8 + wave) * 64 + lane] = r_i;
  __syncthreads();
  double2 a = lds[((wave + 1) & 7) * 64 + lane];
  double2 qo = make_double2(a.x + r_i.x, a.y + r_i.y);
  __syncthreads();
  lds[wave * 64 + lane] = qo;
  __syncthreads();
  double2 b = lds[((wave + 7) & 7) * 64 + lane];
  double2 qi = make_double2(b.x + s_i.x, b.y + s_i.y);
  st(v.OUT, t, lin, qo);
  st(v.IN, t, lout, qi);
  st(v.CO, t, lin, make_double2(co_in.x + qo.x, co_in.y + qo.y));
  st(v.CI, t, lout, make_double2(ci_out.x + qi.x, ci_out.y + qi.y));
}

template <typename F>
static double time_launches(F launch, int n) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int i = 0; i < n; ++i) {
    launch(40 + i, e0, e1);
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (i >= n / 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms[ms.size() / 2] * 1e3;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 1024;
  CK(hipSetDevice(0));
  const int cols = 984, RS = R, rows = 32, bins = cols / 8;
  View v{};
  v.cols = cols; v.RS = RS; v.rows = rows;
  const size_t n = (size_t)rows * cols * RS;
  double** f64[6] = {&v.CI, &v.CO, &v.S, &v.R, &v.IN, &v.OUT};
  for (auto p : f64) { CK(hipMalloc(p, n * 8)); CK(hipMemset(*p, 0, n * 8)); }
  CK(hipMalloc(&v.N, n * 4)); CK(hipMemset(v.N, 0, n * 4));
  std::vector<float> att(n);
  for (size_t i = 0; i < n; ++i) att[i] = (float)((i * 2654435761ull >> 7) % 97);
  CK(hipMalloc(&v.ATT, n * 4)); CK(hipMemcpy(v.ATT, att.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<Rec> rec(cols);
  for (int s = 0; s < cols; ++s) { rec[s].lin = s; rec[s].lout = (s * 7 + 3) % cols; rec[s].tau_sw = 1 + s % 24; for (double& x : rec[s].params) x = 1.0; }
  Rec* drec;
  CK(hipMalloc(&drec, rec.size() * sizeof(Rec))); CK(hipMemcpy(drec, rec.data(), rec.size() * sizeof(Rec), hipMemcpyHostToDevice));
  v.rec = drec;
  const double lanes = (double)cols * RS;
  printf("node_kernel's access pattern without its arithmetic: %d slots x %d replicas, grid (%d, %d) x 512; 60 B read + 48 B written per lane = %.1f MB\n", cols, RS,
         RS / 64, bins, lanes * 108 / 1e6);
  auto report = [&](const char* name, double us, double bytes_per_lane) {
    printf("  %-8s %7.2f us per launch   %5.2f TB/s\n", name, us, lanes * bytes_per_lane / us / 1e6);
  };
  const dim3 grid(RS / 64, bins), block(512);
#define RUN(MODE_, name, bpl) report(name, time_launches([&](int t, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pattern<MODE_>, grid, block, 0, 0, a, b, 0, v, t); }, 60), bpl)
  RUN(FULL, "full", 108);
  RUN(NODEP, "nodep", 108);
  RUN(NOBAR, "nobar", 108);
  RUN(NOREC, "norec", 108);
  RUN(LOADS1, "loads1", 16);
  RUN(PACKED, "packed", 9 * 8 + 6 * 8);
  report("binfast", time_launches([&](int t, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pattern<BINFAST>, dim3(bins, RS / 64), block, 0, 0, a, b, 0, v, t); }, 60), 108);
  report("pipe2", time_launches([&](int t, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pattern_pipe2, dim3(RS / 64, (bins + 1) / 2), block, 0, 0, a, b, 0, v, t, bins / 2); }, 60), 108);
  report("wide", time_launches([&](int t, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(pattern_wide, dim3(RS / 128, bins), block, 0, 0, a, b, 0, v, t); }, 60), 108);
  RUN(FULL, "full", 108);
  return 0;
}
