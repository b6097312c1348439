// pedn_hip.hip -- host side of the MI355X (gfx950) engine for PedNStream's network_loading hot path: the C-ABI of
// include/pedn.h (handle, device buffers, launches, host<->device transfers).  The kernels live in pedn_kernels.hpp, the
// device data structures in pedn_types.hpp, the bit-exact arithmetic primitives in pedn_math.hpp.
//
// Data layout in HBM: every history field is one array [T+1][columns][RS] with the replica index
// fastest (RS = replicas rounded up to a multiple of 128).  A wavefront owns 64 consecutive replicas of ONE link or node:
// topology and link parameters are wave-uniform (scalar loads), every history access of a wave is one coalesced row
// segment, and the data-dependent look-backs gather between rows of the same column.
//
// Compile with -ffp-contract=off: results must match the reference bit for bit.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/pedn.h"
#include "pedn_kernels.hpp"

// ------------------------------------------------------------------------------------------------- host side
static thread_local std::string g_last_error;

struct pedn_sim {
  DevView v{};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int device = 0;
  int n_nodes = 0, n_turns = 0, n_demand = 0, n_od = 0, n_blocks = 0, n_ent = 0;
  int link_owner = 0;  // pedn_run: node_kernel(t + 1)'s slot waves perform the link update of t (one launch per step), PEDN_LINK_OWNER
  int packed_by = 1;   // how nodes were binned into node_kernel's blocks: 0 by degree, 1 by the static load estimate, 2 by measured cost
  // Single-launch plan of small batches with dynamic turning fractions (inline_tf): every device-computed row is short enough for ONE
  // wave and its probabilities fit PEDN_TF_INL_ROWS LDS rows (inline_tf_ok), and the whole node_kernel grid is one generation at 4 waves
  // per SIMD: the slot waves of node_kernel<LU, TF> compute their own rows, a step is one launch.
  bool inline_tf_ok = false;
  int inline_tf = 0;
  int inline_help = 0;      // ... with helper waves: node_kernel_h, sixteen waves per workgroup (PEDN_INLINE_TF=2)
  // A caller that looks at the state after EVERY step (a controller reading densities, an output handler) makes every pending link
  // update a launch of its own and every next step start from stand-alone turning fractions: three launches per step where the plain plan
  // has two.  pedn_step notices (touched: something settled the pending state since the last step) and steps such a caller under the
  // plain plan until two steps in a row go untouched (nine_intersections, step + two reads: 76.0 -> 71 us per step).
  int touched = 0, touch_streak = 0;
  int step_streak = 0;   // consecutive pedn_step(t), pedn_step(t + 1), ... calls with nothing looking at the state in between (see pedn_step)
  size_t node_lds_tf = 0;   // dynamic LDS of node_kernel<.., TF>
  std::vector<int32_t> h_slot_trow;
  int rl_chains = 0;   // pedn_rl_step steps the two halves of the envs as two chains that stay forked ACROSS calls (PEDN_RL_CHAINS)
  int forked = 0;      // stream2 holds work of such a chain that the engine's stream does not order yet (join_forked)
  // Device-resident step clock (DevView.clock; pedn_rl_clock_begin .. pedn_rl_clock_end): while `clocked`, env steps are enqueued with
  // constant arguments (pedn_rl_step_clocked) and the host does not know the step the device is at -- every other entry point that
  // steps, reads or changes state first ends the clocked section (clock_end: synchronises and takes the bookkeeping back).
  int32_t* d_clock = nullptr;
  bool clocked = false;
  int clock_t0 = 0;   // step the clock was set to by pedn_rl_clock_begin
  int valid_hi = 0x7fffffff;   // lazy reset: history rows above this index are neither written nor cleared (DevView.valid_hi)
  int link_pending = -1;  // owner-wave plan: step whose link update has not been performed yet, -1 none
  int fuse_obs = 1;    // pedn_rl_step: observations / rewards ride in the link update's launch (PEDN_FUSE_OBS=0: own launch)
  // (The link update as a launch of its own runs one replica per lane -- link_kernel_1r: 42-47 VGPRs, 8 waves per SIMD; melbourne x 1024
  // 12.3-12.6 against 12.7-13.1 us with two replicas per lane, profiles/r03_link_kernel_variants.txt; inside link_turn_kernel, whose
  // budget is set by the turning fractions, it keeps two replicas per lane: half the workgroups.)
  int max_degree = 0;     // largest number of incident corridors of a node
  size_t node_lds = 0;    // dynamic LDS bytes of node_kernel
  hipStream_t stream2 = nullptr;   // second half of the replicas in pedn_run (two_streams)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int warmed_chains = 1;  // chains whose streams exist and were probed to overlap
  int chains = 1;         // plan of pedn_run for long ranges: 1 or 2 chains of launches (two_streams = chains > 1)
  int run_chains = 1;     // chains of the range being launched (launch_step / flush_links: the last chain does the bookkeeping)
  int stream_probe_attempts = 0;   // warm_chain_streams: probes run until the chains' streams were seen to overlap
  float stream_probe_ms = 0.0f;
  int two_streams = 0;    // pedn_run launches the two halves of the batch on two streams (replicas are independent)
  int second_launch = 0;  // launch_step: a launch followed node_kernel
  int fuse_tp = 0;     // the link update and the next step's turn probabilities share one launch (launch_step)
  int tp_ran = 0;      // launch_step launched the stand-alone turn_frac_kernel (pedn_profile_step)
  int tp_ready = -1;   // step whose turning fractions are in tfd[step & 1] (written by link_turn_kernel of the step before), -1: none
  std::vector<int32_t> node_turn_ptr, node_demand_row;
  std::vector<int32_t> h_up_od_ptr, h_upod_od, h_pair_upod;  // route-choice tables needed to re-tabulate P(od | up)
  std::vector<double> h_od_w;
  std::vector<int32_t> h_turn_pair_ptr, h_pair_const, h_turn_mode;
  double* d_pair_pod = nullptr;
  double* d_turn_tab = nullptr;
  RlView rl{};
  bool rl_ready = false;
  bool node_lp = false;   // PEDN_NODE_OPTIMAL: the node LP instead of the classic rule
  bool rl_fold = false;   // gater-only agent set: pedn_rl_step lets node_kernel apply the actions (no launch of rl_apply_kernel)
  std::vector<SlotRec> h_slot_rec;
  SlotRec* d_slot_rec = nullptr;
  std::vector<double> h_front_u, h_back_u, h_tf_u;
  double *d_front_u = nullptr, *d_back_u = nullptr, *d_tf_u = nullptr;
  std::vector<int32_t> h_node_dyn, h_slot_dyn;  // per node: dynamic; per slot: SlotRec.dyn (0 static, 1 turn_frac_kernel, 2 tabulated)
  std::vector<int32_t> h_node_slot_ptr;
  std::vector<double> h_ttab, h_ttab_r;         // host copies of turn_tab [T+1][n_turns] / turn_tab_r [n_turns][R] (tabulated rows: final values)
  std::vector<char> h_rl_link;
  LinkPR* d_prm = nullptr;           // per-replica link parameters [L][RS] (pedn_set_link_params, pedn_randomize_scenarios)
  LinkPR* d_prm_draw = nullptr;      // recent-history mode: where pedn_randomize_scenarios draws before the result is accepted
  double *d_pair_pod_r = nullptr, *d_turn_tab_r = nullptr;
  // per-replica OD weights and the tables derived from them on the device (scenario_pod_tables)
  double *d_od_w_r = nullptr, *d_pod_tot = nullptr;     // [n_od][RS], [n_up][RS]
  const int32_t *d_up_od_ptr = nullptr, *d_upod_od = nullptr, *d_upod_up = nullptr, *d_pair_upod = nullptr, *d_turn_pair_ptr = nullptr,
                *d_turn_mode = nullptr, *d_tab_rows = nullptr;
  int n_tab_rows = 0, n_upod = 0;
  bool pod_tables_uploaded = false, ttab_r_stale = false;   // h_ttab_r is older than turn_tab_r on the device
  int* d_max_tau = nullptr;
  int n_pair = 0, n_up = 0, n_over = 0;
  int n_tf_heavy_quads = 0;  // leading workgroups of turn_frac_body with long chains (more than PEDN_TF_HEAVY_GROUPS softmax groups in a row)
  long step_epoch = 1;  // counts launched steps; h_tf_set_epoch[node] == step_epoch: fractions imposed since the last step
  std::vector<long> h_tf_set_epoch;
  int rows64[7], rows32[6];  // history rows of every field (T+1, or the size of its ring in recent-history mode)
  int last_t = -1;     // last step launched (pedn_get_turning_fractions: which buffer holds a dynamic node's fractions)
  std::vector<void*> allocs;
  // host <-> device staging: two slots used in turn, each a pinned host buffer + a device buffer + the event recorded behind
  // the slot's last consumer, so that an upload neither waits for the stream nor borrows caller memory beyond the call
  struct Stage { void* pin = nullptr; void* dev = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; };
  Stage stage[2];
  int stage_next = 0;
  void* rl_pin = nullptr;      // pinned landing buffer of the RL step's observations + rewards (rl_fetch)
  size_t rl_pin_bytes = 0;
  std::string err;
};

static int fail(pedn_sim* s, int code, const std::string& msg) {
  g_last_error = msg;
  if (s) s->err = msg;
  return code;
}

#define HIP_TRY(sim, expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(sim, PEDN_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

template <typename T>
static int upload(pedn_sim* s, const T* src, size_t n, const T** dst) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIP_TRY(s, hipMalloc(&p, bytes));
  s->allocs.push_back(p);
  if (n) HIP_TRY(s, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  *dst = (const T*)p;
  return PEDN_OK;
}

template <typename T>
static int dalloc(pedn_sim* s, size_t n, T** dst) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) return fail(s, PEDN_E_NOMEM, std::string("hipMalloc of ") + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
  s->allocs.push_back(p);
  *dst = (T*)p;
  return PEDN_OK;
}

// A staging slot of at least `bytes` whose previous use has completed (the other slot may still be in flight).
#define PEDN_DIRECT_READ_BYTES 65536     // pedn_read_block: up to this size the gather kernel writes into the pinned buffer itself
#define PEDN_IN_PLACE_BYTES (4u << 20)   // host rows up to this size are read in place by their consuming kernel (stage_in_place; measured up to 256 KB)
static int stage_acquire(pedn_sim* s, size_t bytes, pedn_sim::Stage** out) {
  pedn_sim::Stage& st = s->stage[s->stage_next];
  s->stage_next ^= 1;
  if (!st.done) HIP_TRY(s, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
  else HIP_TRY(s, hipEventSynchronize(st.done));
  if (bytes > st.bytes) {
    if (st.dev) HIP_TRY(s, hipFree(st.dev));
    if (st.pin) HIP_TRY(s, hipHostFree(st.pin));
    st.dev = st.pin = nullptr;
    st.bytes = 0;
    const size_t want = std::max<size_t>(bytes, 1 << 20);
    HIP_TRY(s, hipMalloc(&st.dev, want));
    HIP_TRY(s, hipHostMalloc(&st.pin, want, hipHostMallocDefault));
    st.bytes = want;
  }
  *out = &st;
  return PEDN_OK;
}

// host values -> the slot's device buffer (through its pinned buffer: the caller's memory is not touched after the return)
static int stage_upload(pedn_sim* s, pedn_sim::Stage* st, const void* src, size_t bytes, size_t offset = 0) {
  memcpy((char*)st->pin + offset, src, bytes);
  HIP_TRY(s, hipMemcpyAsync((char*)st->dev + offset, (char*)st->pin + offset, bytes, hipMemcpyHostToDevice, s->stream));
  return PEDN_OK;
}

// call after the last launch that reads or writes the slot
static int stage_commit(pedn_sim* s, pedn_sim::Stage* st) {
  HIP_TRY(s, hipEventRecord(st->done, s->stream));
  return PEDN_OK;
}

// Host rows that ONE kernel reads once (the action rows of a host-driven env step): copied into a pinned slot and read by the kernel IN
// PLACE over the bus -- no copy command in front of the launch (a DMA costs ~20 us of stream latency, a copy from pageable memory waits
// for the stream; the consuming wave's bus read costs it ~2 us).  The caller records the slot's event behind the consuming launch
// (stage_commit).  2048 envs: 93-96 -> 81 us per host-driven step, 48 -> 29 without a fetch.
static int stage_in_place(pedn_sim* s, const void* src, size_t bytes, pedn_sim::Stage** out) {
  int rc = stage_acquire(s, bytes, out);
  if (rc != PEDN_OK) return rc;
  memcpy((*out)->pin, src, bytes);
  return PEDN_OK;
}

// host bytes -> a device buffer of the engine; the caller's memory is borrowed for the call only, so the copy is waited for (rows beyond
// PEDN_IN_PLACE_BYTES; what was measured instead for smaller ones -- a copy COMMAND from a pinned slot, a copy KERNEL from it -- lost to
// reading them in place, profiles/r05_host_step_time.txt)
static int upload_through_stage(pedn_sim* s, void* dst, const void* src, size_t bytes) {
  HIP_TRY(s, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PEDN_OK;
}

// The first launch on a stream and the first cross-stream wait cost the runtime ~0.2 ms (queue creation, signal set-up): pay
// that when the two-chain plan is chosen, not inside the first pedn_run that uses it.
// The two chains only pay off when the two streams are served by DIFFERENT hardware queues.  The runtime maps streams onto a few
// queues (GPU_MAX_HW_QUEUES, 4 by default) and, in a process that holds other streams already -- torch + RCCL under the launcher --
// handed both of the engine's streams the same one: the chains then run one behind the other (melbourne x 1024 30.3 -> 39.0 us per
// step, delft 42.8 -> 64.8, profiles/r04_stream_queues.txt).  So the pairing is probed, not assumed: a 300 us spin on each stream,
// timed together; while they do not overlap another candidate for stream2 is created (the rejected ones stay alive until the
// search ends, so that the runtime moves on to its other queues).
static hipStream_t chain_stream(const pedn_sim* s, int c) { return c <= 0 ? s->stream : s->stream2; }

// a 300 us spin on the engine's stream and on stream2 at once; *ms = how long both took together
static int probe_overlap(pedn_sim* s, float* ms) {
  HIP_TRY(s, hipEventRecord(s->ev_fork, s->stream));
  HIP_TRY(s, hipStreamWaitEvent(s->stream2, s->ev_fork, 0));
  HIP_TRY(s, hipEventRecord(s->ev0, s->stream));
  for (int c = 0; c < 2; ++c) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, chain_stream(s, c), 30000ull);   // ticks of the constant 100 MHz clock
  HIP_TRY(s, hipEventRecord(s->ev_join, s->stream2));
  HIP_TRY(s, hipStreamWaitEvent(s->stream, s->ev_join, 0));
  HIP_TRY(s, hipEventRecord(s->ev1, s->stream));
  HIP_TRY(s, hipEventSynchronize(s->ev1));
  HIP_TRY(s, hipEventElapsedTime(ms, s->ev0, s->ev1));
  return PEDN_OK;
}
// The second chain's stream, probed to overlap with the engine's; *got = 2 when the two really run side by side, else 1 (the plan
// falls back to one chain).  (Four chains were built and measured no faster: profiles/r04_four_chains.txt.)
static int warm_chain_streams(pedn_sim* s, int* got) {
  *got = 1;
  bool probe = true;
  if (const char* f = getenv("PEDN_STREAM_PROBE")) probe = atoi(f) != 0;
  if (!probe) { *got = 2; return PEDN_OK; }
  std::vector<hipStream_t> rejected;
  int rc = PEDN_OK, attempts = 0;
  for (int attempt = 0; attempt < 10 && rc == PEDN_OK; ++attempt) {
    float ms = 0.0f;
    rc = probe_overlap(s, &ms);   // the first pass pays for whatever the runtime sets up lazily (queue creation: ~0.2 ms)
    if (rc == PEDN_OK) rc = probe_overlap(s, &ms);
    if (rc != PEDN_OK) break;
    ++attempts;
    s->stream_probe_ms = ms;
    if (ms < 0.45f) { *got = 2; break; }        // overlapped: 0.3 ms + overheads; one behind another: >= 0.6 ms
    hipStream_t next = nullptr;
    if (hipStreamCreateWithFlags(&next, hipStreamNonBlocking) != hipSuccess) break;   // keep what we have
    rejected.push_back(s->stream2);
    s->stream2 = next;
  }
  s->stream_probe_attempts = attempts;
  for (hipStream_t x : rejected) hipStreamDestroy(x);
  return rc;
}

// rows [row0, row1) of every history field back to their initial values (`what` as in init_state_kernel; 1 = everything)
static int clear_rows(pedn_sim* s, int row0, int row1, int what) {
  DevView& v = s->v;
  if (row1 <= row0) return PEDN_OK;
  if (what & 1)
    for (int f = 0; f < 4; ++f) {
      const int a = std::min(row0, s->rows64[f]), b = std::min(row1, s->rows64[f]);
      if (b > a) HIP_TRY(s, hipMemsetAsync(v.f64[f] + (size_t)a * v.Lall * v.RS, 0, (size_t)(b - a) * v.Lall * v.RS * sizeof(double), s->stream));
    }
  if (v.L > 0) {
    int max_rows = 0;  // over the fields init_state_kernel fills (all of them have L columns)
    for (int f = 4; f < 7; ++f) max_rows = std::max(max_rows, s->rows64[f]);
    for (int g = 0; g < 6; ++g) max_rows = std::max(max_rows, s->rows32[g]);
    const int a = std::min(row0, max_rows), b = std::min(row1, max_rows);
    if (b > a) {
      const size_t n_l = (size_t)(b - a) * v.L * v.RS;
      hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((n_l + 255) / 256)), dim3(256), 0, s->stream, v, a, b - a, what);
      HIP_TRY(s, hipGetLastError());
    }
  }
  return PEDN_OK;
}

static int reset_state(pedn_sim* s) {
  DevView& v = s->v;
  HIP_TRY(s, hipMemsetAsync(v.flags, 0, (size_t)v.RS * sizeof(uint32_t), s->stream));
  s->valid_hi = v.valid_hi = 0x7fffffff;
  return clear_rows(s, 0, v.T1, 1);
}

// Lazy reset (full-record mode): only what a fresh episode reads before it writes is restored -- row 0 of every field, the rows of
// avg_travel_time below the window (link.py:91: never written by the link update), the gate record (stored only where it differs from
// the width) -- and every other row is declared unwritten (valid_hi = 0): 19.5 GB -> 2.5 GB for 45_intersections x 2048 envs.
static int reset_state_lazy(pedn_sim* s) {
  DevView& v = s->v;
  if (v.hist) return reset_state(s);    // the rings are small: nothing to save
  HIP_TRY(s, hipMemsetAsync(v.flags, 0, (size_t)v.RS * sizeof(uint32_t), s->stream));
  int rc;
  if ((rc = clear_rows(s, 0, 1, 1)) || (rc = clear_rows(s, 1, std::min(v.W, v.T1), 2)) || (rc = clear_rows(s, 1, v.T1, 4))) return rc;
  s->valid_hi = v.valid_hi = 0;
  return PEDN_OK;
}

// Rows (valid_hi, upto] cleared now: something is about to read them from memory (an out-of-order step, an observation of a step that
// has not run, a zero-copy consumer).
static int catch_up(pedn_sim* s, int upto) {
  if (s->valid_hi >= upto) return PEDN_OK;
  upto = std::min(upto, s->v.T1 - 1);
  DevView view = s->v;
  int rc = PEDN_OK;
  if (upto > s->valid_hi) {
    // everything but the gate record and the low rows of avg_travel_time, which the lazy reset restored already
    DevView& v = s->v;
    const int a = s->valid_hi + 1, b = upto + 1;
    for (int f = 0; f < 4; ++f) HIP_TRY(s, hipMemsetAsync(v.f64[f] + (size_t)a * v.Lall * v.RS, 0, (size_t)(b - a) * v.Lall * v.RS * sizeof(double), s->stream));
    if (v.L > 0) {
      // sending / receiving flow of step t are entries t - 1: entry valid_hi belongs to the step that is being skipped
      const size_t n_1 = (size_t)v.L * v.RS;
      hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((n_1 + 255) / 256)), dim3(256), 0, s->stream, view, s->valid_hi, 1, 16);
      HIP_TRY(s, hipGetLastError());
      const size_t n_l = (size_t)(b - a) * v.L * v.RS;
      hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((n_l + 255) / 256)), dim3(256), 0, s->stream, view, a, b - a, 8);
      rc = hipGetLastError() == hipSuccess ? PEDN_OK : fail(s, PEDN_E_DEVICE, "init_state_kernel");
    }
    s->valid_hi = s->v.valid_hi = upto;
  }
  return rc;
}

static int push_uniform(pedn_sim* s) {
  for (size_t l = 0; l < s->h_rl_link.size(); ++l)
    if (s->h_rl_link[l]) s->h_front_u[l] = s->h_back_u[l] = __builtin_nan("");
  if (!s->h_front_u.empty()) {
    HIP_TRY(s, hipMemcpyAsync(s->d_front_u, s->h_front_u.data(), s->h_front_u.size() * 8, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(s, hipMemcpyAsync(s->d_back_u, s->h_back_u.data(), s->h_back_u.size() * 8, hipMemcpyHostToDevice, s->stream));
  }
  if (!s->h_tf_u.empty())
    HIP_TRY(s, hipMemcpyAsync(s->d_tf_u, s->h_tf_u.data(), s->h_tf_u.size() * 8, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));  // the host vectors may change right after
  return PEDN_OK;
}

// check_fractions (path_finder.py:691-715) for the rows whose fractions are tabulated on the host (SlotRec.dyn == 2): the
// same binary64 operations the device applies to the other rows.  tab[turn * stride] holds the raw sums.
static void finish_tabulated_rows(const pedn_sim* s, double* tab, size_t stride) {
  for (int n = 0; n < s->n_nodes; ++n) {
    const int s0 = s->h_node_slot_ptr[n], d = s->h_node_slot_ptr[n + 1] - s0;
    for (int i = 0; i < d; ++i) {
      if (s->h_slot_dyn[s0 + i] != 2) continue;
      const int ta = s->node_turn_ptr[n] + i * (d - 1);
      double rowsum = 0.0;
      for (int jj = 0; jj < d - 1; ++jj) rowsum = (jj == 0) ? tab[(size_t)(ta + jj) * stride] : rowsum + tab[(size_t)(ta + jj) * stride];
      if (fabs(rowsum - 1) > 1e-3)
        for (int jj = 0; jj < d - 1; ++jj) {
          double& f = tab[(size_t)(ta + jj) * stride];
          f = rowsum > 1e-6 ? f / rowsum : 1.0 / (double)(d - 1);
        }
    }
  }
}

// P(od | up)[t] = w_od[t] / sum over the upstream's ODs (uniform when the sum is 0), path_finder.py:599-615; the sum runs in
// table order.  Replica independent, so it is tabulated once per (step, product) on the host with the same binary64 operations.
static int tabulate_pair_pod(pedn_sim* s) {
  const int T1 = s->v.T1, np = s->n_pair;
  if (np == 0 || s->d_turn_tab == nullptr) return PEDN_OK;
  std::vector<double> upod((size_t)s->h_upod_od.size());
  std::vector<double> table((size_t)T1 * np);
  for (int t = 0; t < T1; ++t) {
    for (int u = 0; u < s->n_up; ++u) {
      const int a = s->h_up_od_ptr[u], b = s->h_up_od_ptr[u + 1];
      double tot = 0.0;
      for (int q = a; q < b; ++q) tot += s->h_od_w[(size_t)s->h_upod_od[q] * T1 + t];
      for (int q = a; q < b; ++q)
        upod[q] = tot > 0.0 ? s->h_od_w[(size_t)s->h_upod_od[q] * T1 + t] / tot : (b - a > 0 ? 1.0 / (double)(b - a) : 0.0);
    }
    for (int q = 0; q < np; ++q) table[(size_t)t * np + q] = upod[s->h_pair_upod[q]];
  }
  // turns whose products all have the constant probability 1: the fraction is the plain sequential sum of P(od | up)
  const int nt = s->n_turns;
  std::vector<double> ttab((size_t)T1 * std::max(nt, 1), 0.0);
  for (int t = 0; t < T1; ++t)
    for (int tn = 0; tn < nt; ++tn) {
      if (!s->h_turn_mode[tn]) continue;
      double acc = 0.0;
      for (int q = s->h_turn_pair_ptr[tn]; q < s->h_turn_pair_ptr[tn + 1]; ++q) acc += 1.0 * table[(size_t)t * np + q];
      ttab[(size_t)t * nt + tn] = acc;
    }
  for (int t = 0; t < T1; ++t) finish_tabulated_rows(s, &ttab[(size_t)t * nt], 1);
  s->h_ttab = ttab;
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy(s->d_pair_pod, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(s, hipMemcpy(s->d_turn_tab, ttab.data(), ttab.size() * sizeof(double), hipMemcpyHostToDevice));
  return PEDN_OK;
}

// rows[n_rows][R] (host) -> dst[n_rows][RS]
template <typename T>
static int push_matrix(pedn_sim* s, T* dst, const T* src, int n_rows) {
  DevView& v = s->v;
  if (n_rows == 0) return PEDN_OK;
  HIP_TRY(s, hipMemcpy2D(dst, (size_t)v.RS * sizeof(T), src, (size_t)v.R * sizeof(T), (size_t)v.R * sizeof(T), n_rows, hipMemcpyHostToDevice));
  return PEDN_OK;
}

// per-replica link-parameter records, allocated at the first use; padding lanes hold valid numbers and stay idle
static int ensure_link_records(pedn_sim* s) {
  if (s->d_prm) return PEDN_OK;
  const DevView& v = s->v;
  const size_t n = (size_t)v.L * v.RS;
  int rc = dalloc(s, n, &s->d_prm);
  if (rc != PEDN_OK) return rc;
  LinkPR fill;
  fill.kc = 1.0; fill.kj = 2.0; fill.vf = 1.0; fill.tt0 = 1.0f; fill.fft = 32767; fill.tau_sw = 1;
  std::vector<LinkPR> h(n, fill);
  HIP_TRY(s, hipMemcpy(s->d_prm, h.data(), n * sizeof(LinkPR), hipMemcpyHostToDevice));
  return PEDN_OK;
}

// device copies of the route-choice tables that P(od | up) per replica is derived from, and its buffers
static int ensure_pod_tables(pedn_sim* s) {
  if (s->pod_tables_uploaded) return PEDN_OK;
  DevView& v = s->v;
  const int np = s->n_pair, nt = s->n_turns;
  int rc;
  std::vector<int32_t> upod_up(std::max<size_t>(s->h_upod_od.size(), 1), 0), rows;
  for (int u = 0; u < s->n_up; ++u)
    for (int q = s->h_up_od_ptr[u]; q < s->h_up_od_ptr[u + 1]; ++q) upod_up[q] = u;
  for (int n = 0; n < s->n_nodes; ++n) {   // the rows finish_tabulated_rows visits
    const int s0 = s->h_node_slot_ptr[n], d = s->h_node_slot_ptr[n + 1] - s0;
    for (int i = 0; i < d; ++i)
      if (s->h_slot_dyn[s0 + i] == 2) { rows.push_back(s->node_turn_ptr[n] + i * (d - 1)); rows.push_back(d - 1); }
  }
  s->n_tab_rows = (int)rows.size() / 2;
  s->n_upod = (int)s->h_upod_od.size();
  if ((rc = upload(s, s->h_up_od_ptr.data(), s->h_up_od_ptr.size(), &s->d_up_od_ptr)) || (rc = upload(s, s->h_upod_od.data(), s->h_upod_od.size(), &s->d_upod_od)) ||
      (rc = upload(s, upod_up.data(), upod_up.size(), &s->d_upod_up)) || (rc = upload(s, s->h_pair_upod.data(), s->h_pair_upod.size(), &s->d_pair_upod)) ||
      (rc = upload(s, s->h_turn_pair_ptr.data(), s->h_turn_pair_ptr.size(), &s->d_turn_pair_ptr)) ||
      (rc = upload(s, s->h_turn_mode.data(), s->h_turn_mode.size(), &s->d_turn_mode)) || (rc = upload(s, rows.data(), rows.size(), &s->d_tab_rows)))
    return rc;
  if ((rc = dalloc(s, (size_t)std::max(s->n_od, 1) * v.RS, &s->d_od_w_r)) || (rc = dalloc(s, (size_t)std::max(s->n_up, 1) * v.RS, &s->d_pod_tot)) ||
      (rc = dalloc(s, (size_t)np * v.RS, &s->d_pair_pod_r)) || (rc = dalloc(s, (size_t)std::max(nt, 1) * v.RS, &s->d_turn_tab_r)))
    return rc;
  HIP_TRY(s, hipMemset(s->d_od_w_r, 0, (size_t)std::max(s->n_od, 1) * v.RS * 8));
  HIP_TRY(s, hipMemset(s->d_pair_pod_r, 0, (size_t)np * v.RS * 8));
  HIP_TRY(s, hipMemset(s->d_turn_tab_r, 0, (size_t)std::max(nt, 1) * v.RS * 8));
  s->pod_tables_uploaded = true;
  return PEDN_OK;
}

// P(od | up) and the tabulated rows of every replica from d_od_w_r (path_finder.py:599-615, :691-715), on the device
static int scenario_pod_tables(pedn_sim* s) {
  DevView& v = s->v;
  const int np = s->n_pair, nt = s->n_turns, RS = v.RS;
  auto blocks = [&](size_t rows) { return dim3((unsigned)((rows * (size_t)RS + 255) / 256)); };
  if (s->n_up > 0)
    hipLaunchKernelGGL(pod_tot_kernel, blocks(s->n_up), dim3(256), 0, s->stream, (const double*)s->d_od_w_r, s->d_up_od_ptr, s->d_upod_od, s->n_up, RS, s->d_pod_tot);
  hipLaunchKernelGGL(pod_pair_kernel, blocks(np), dim3(256), 0, s->stream, (const double*)s->d_od_w_r, s->d_up_od_ptr, s->d_upod_od, s->d_upod_up,
                     s->d_pair_upod, (const double*)s->d_pod_tot, np, RS, s->d_pair_pod_r);
  if (nt > 0)
    hipLaunchKernelGGL(pod_turn_kernel, blocks(nt), dim3(256), 0, s->stream, (const double*)s->d_pair_pod_r, s->d_turn_pair_ptr, s->d_turn_mode, nt, RS, s->d_turn_tab_r);
  if (s->n_tab_rows > 0)
    hipLaunchKernelGGL(pod_finish_kernel, blocks(s->n_tab_rows), dim3(256), 0, s->stream, s->d_turn_tab_r, s->d_tab_rows, s->n_tab_rows, RS);
  HIP_TRY(s, hipGetLastError());
  v.pair_pod_r = s->d_pair_pod_r; v.turn_tab_r = s->d_turn_tab_r;
  v.pod_pr = 1;
  s->ttab_r_stale = true;
  return PEDN_OK;
}

extern "C" {

static int push_rows(pedn_sim* s, double* dst, const double* values, int n_rows, size_t row0, size_t row_stride, int replica);
static void flush_links(pedn_sim* s, int half, hipEvent_t* ev);
static inline void pending_links_first(pedn_sim* s);
static inline void join_forked(pedn_sim* s);
typedef void (*node_kernel_fn)(DevView, int);
static node_kernel_fn node_kernel_for(const pedn_sim* s, bool lu, bool tf);
static int clock_end(pedn_sim* s);
static void prewarm_chains(pedn_sim* s);
static int fork_chains(pedn_sim* s, int n);

int pedn_abi_version(void) { return PEDN_ABI_VERSION; }

const char* pedn_last_error(const pedn_sim* sim) { return sim ? sim->err.c_str() : g_last_error.c_str(); }

int pedn_create(const pedn_model_desc* m, int32_t n_replicas, int32_t replica_offset, uint64_t seed, int32_t rng_mode,
                int32_t device, pedn_sim** out) {
  if (!m || !out) return fail(nullptr, PEDN_E_ARG, "null argument");
  *out = nullptr;
  if (m->abi_version != PEDN_ABI_VERSION) return fail(nullptr, PEDN_E_ARG, "pedn_model_desc.abi_version mismatch");
  if (n_replicas < 1 || m->n_nodes < 1 || m->n_links < 0 || m->T < 2 || m->window < 1)
    return fail(nullptr, PEDN_E_ARG, "bad sizes in model description");
  const int N = m->n_nodes, L = m->n_links;
  const int n_slots = m->node_slot_ptr[N];
  // ---- validate topology against what the kernels assume
  for (int n = 0; n < N; ++n) {
    int deg = m->node_slot_ptr[n + 1] - m->node_slot_ptr[n];
    if (deg < 2 || deg > PEDN_MAX_DEGREE)
      return fail(nullptr, PEDN_E_ARG, "node degree " + std::to_string(deg) + " outside 2.." + std::to_string(PEDN_MAX_DEGREE));
    if (m->node_kind[n] == 0 && deg != 2) return fail(nullptr, PEDN_E_ARG, "one-to-one node with degree != 2");
    if (m->node_turn_ptr[n + 1] - m->node_turn_ptr[n] != deg * (deg - 1)) return fail(nullptr, PEDN_E_ARG, "node_turn_ptr inconsistent");
    for (int k = m->node_slot_ptr[n]; k < m->node_slot_ptr[n + 1]; ++k) {
      int li = m->slot_in_link[k], lo = m->slot_out_link[k];
      if (li < 0 || lo < 0 || li >= L + m->n_vlinks || lo >= L + m->n_vlinks) return fail(nullptr, PEDN_E_ARG, "slot link index out of range");
      if ((li >= L) != (lo >= L)) return fail(nullptr, PEDN_E_ARG, "virtual/physical mismatch in a slot");
      if (li < L && m->link_rev[li] != lo) return fail(nullptr, PEDN_E_ARG, "slot does not hold a reverse pair");
      if (li >= L && (m->node_demand_row[n] < 0 || m->node_demand_row[n] >= m->n_demand)) return fail(nullptr, PEDN_E_ARG, "virtual slot without demand row");
    }
  }
  for (int l = 0; l < L; ++l) {
    int rv = m->link_rev[l];
    if (rv < 0 || rv >= L || rv == l || m->link_rev[rv] != l) return fail(nullptr, PEDN_E_ARG, "link_rev is not an involution");
    if (m->link_fd[l] < 0 || m->link_fd[l] > 2) return fail(nullptr, PEDN_E_ARG, "unknown fundamental diagram type");
  }
  for (int g = 0; g < m->n_grp; ++g)
    if (m->grp_ent_ptr[g + 1] - m->grp_ent_ptr[g] > PEDN_MAX_DEGREE - 1) return fail(nullptr, PEDN_E_ARG, "softmax group too large");

  if (m->history_mode != PEDN_HIST_FULL && m->history_mode != PEDN_HIST_RECENT) return fail(nullptr, PEDN_E_ARG, "unknown history_mode");
  if (m->node_model != PEDN_NODE_CLASSIC && m->node_model != PEDN_NODE_OPTIMAL) return fail(nullptr, PEDN_E_ARG, "unknown node_model");

  HIP_TRY(nullptr, hipSetDevice(device));
  pedn_sim* s = new pedn_sim();
  s->device = device;
  DevView& v = s->v;
  v.L = L;
  v.Lall = L + m->n_vlinks;
  v.T1 = m->T + 1;
  v.R = n_replicas;
  v.RS = (n_replicas + 127) / 128 * 128;  // a link_kernel wave covers 128 replicas (2 per lane), a node_kernel wave 64
  v.sub0 = 0;
  v.subRS = v.RS;
  v.W = m->window;
  v.dt = m->dt;
  {  // history rows (DevView.m64 / m32)
    const int FULL = 0x7fffffff;
    auto ring = [&](int need) {  // power-of-two ring of at least `need` rows, or all rows when that is no saving
      int p = 1;
      while (p < need) p <<= 1;
      return p >= v.T1 ? FULL : p - 1;
    };
    for (int f = 0; f < 7; ++f) v.m64[f] = FULL;
    for (int g = 0; g < 6; ++g) v.m32[g] = FULL;
    v.hist = m->history_mode == PEDN_HIST_RECENT;
    if (v.hist) {
      int max_sw = 0;
      for (int l = 0; l < L; ++l) max_sw = std::max(max_sw, m->link_tau_sw[l]);
      // cumulative_outflow[t' + 1 - tau_shockwave] (link.py:380-390) and [t'], with room for the per-replica scenarios of the
      // randomisers: free_flow_speed x U(0.6, 0.9) stretches the shock-wave look-back by up to 1 / 0.6 (env_loader.py:410-412; the
      // k_critical / k_jam factor cancels in the shock-wave speed unless a floor binds -- then pedn_set_link_params /
      // pedn_randomize_scenarios refuse the scenario)
      // (a look-back of x.5 before rounding divided by 0.6 can round one further up than max_sw / 0.6: sized for that)
      v.m64[F_CO] = ring((int)ceil(((double)max_sw + 0.5) / 0.6) + 2);
      v.m64[F_OUT] = v.m64[F_S] = v.m64[F_R] = v.m64[F_GATE] = ring(4);   // [t], [t-1], [t-2] at most
      v.m32[G_TT] = ring(v.W + 2);        // travel_time[t - W] leaves the moving average (link.py:183-186)
      v.m32[G_ATT] = v.m32[G_N] = v.m32[G_K] = v.m32[G_V] = v.m32[G_LF] = ring(4);
    }
    for (int f = 0; f < 7; ++f) s->rows64[f] = v.m64[f] == FULL ? v.T1 : v.m64[f] + 1;
    for (int g = 0; g < 6; ++g) s->rows32[g] = v.m32[g] == FULL ? v.T1 : v.m32[g] + 1;
  }
  v.pf_temp = m->pf_temp; v.pf_alpha = m->pf_alpha; v.pf_beta = m->pf_beta; v.pf_omega = m->pf_omega; v.pf_eps = m->pf_eps;
  v.k0 = (uint32_t)seed; v.k1 = (uint32_t)(seed >> 32);
  v.replica_offset = (uint32_t)replica_offset;
  v.meanfield = rng_mode == PEDN_RNG_MEANFIELD;
  v.n_grp = m->n_grp;
  if (const char* d = getenv("PEDN_TF_GENERAL")) v.tf_general = atoi(d);  // diagnostics: 1 general softmax path, 2 general row-sum path
  s->n_nodes = N; s->n_turns = m->n_turns; s->n_demand = m->n_demand; s->n_od = m->n_od; s->n_ent = m->n_ent;
  s->node_turn_ptr.assign(m->node_turn_ptr, m->node_turn_ptr + N + 1);
  s->h_node_slot_ptr.assign(m->node_slot_ptr, m->node_slot_ptr + N + 1);
  s->h_tf_set_epoch.assign(N, 0);
  s->node_demand_row.assign(m->node_demand_row, m->node_demand_row + N);

#define TRY(expr) do { int _rc = (expr); if (_rc != PEDN_OK) { std::string keep = g_last_error; pedn_destroy(s); g_last_error = keep; return _rc; } } while (0)
  {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return fail(nullptr, PEDN_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    // a null stream2 would be the legacy default stream (it synchronises with everything): every handle is checked
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) { pedn_destroy(s); return fail(nullptr, PEDN_E_DEVICE, std::string("second stream / events: ") + hipGetErrorString(e)); }
  }
  // ---- memory budget
  {
    size_t RS = v.RS, T1 = v.T1;
    size_t need = (size_t)(m->n_demand) * T1 * RS * 8 + (size_t)(3 * m->n_turns + 3 * L) * RS * 8;
    for (int f = 0; f < 7; ++f) need += (size_t)s->rows64[f] * (f < 4 ? v.Lall : L) * RS * 8;
    for (int g = 0; g < 6; ++g) need += (size_t)s->rows32[g] * L * RS * 4;
    size_t free_b = 0, total_b = 0;
    const hipError_t mi = hipMemGetInfo(&free_b, &total_b);
    if (mi != hipSuccess) {
      std::string msg = std::string("hipMemGetInfo: ") + hipGetErrorString(mi);
      pedn_destroy(s);
      return fail(nullptr, PEDN_E_DEVICE, msg);
    }
    if (need + (256u << 20) > free_b) {
      std::string msg = "not enough HBM: need " + std::to_string(need >> 20) + " MiB, free " + std::to_string(free_b >> 20) + " MiB";
      pedn_destroy(s);
      return fail(nullptr, PEDN_E_NOMEM, msg);
    }
  }
  // ---- static tables
  std::vector<LinkP> lp(std::max(L, 1));
  for (int l = 0; l < L; ++l) {
    LinkP& P = lp[l];
    P.length = m->link_length[l]; P.width = m->link_width[l]; P.vf = m->link_vf[l]; P.kc = m->link_kc[l]; P.kj = m->link_kj[l];
    P.gamma = m->link_gamma[l]; P.act = m->link_act[l]; P.bi = m->link_bi[l]; P.noise = m->link_noise[l];
    P.tt0 = m->link_tt0[l]; P.rev = m->link_rev[l]; P.sep = m->link_sep[l]; P.fd = m->link_fd[l];
    P.tau_sw = m->link_tau_sw[l]; P.fft = m->link_fft[l]; P.pad = 0;
    P.derive();
  }
  TRY(upload(s, lp.data(), lp.size(), &v.lp));
  TRY(upload(s, m->node_kind, N, &v.node_kind));
  TRY(upload(s, m->node_slot_ptr, N + 1, &v.node_slot_ptr));
  TRY(upload(s, m->node_turn_ptr, N + 1, &v.node_turn_ptr));
  TRY(upload(s, m->node_demand_row, N, &v.node_demand_row));
  TRY(upload(s, m->node_dyn, N, &v.node_dyn));
  TRY(upload(s, m->slot_in_link, n_slots, &v.slot_in));
  TRY(upload(s, m->slot_out_link, n_slots, &v.slot_out));
  s->h_up_od_ptr.assign(m->up_od_ptr, m->up_od_ptr + m->n_up + 1);
  s->h_upod_od.assign(m->upod_od, m->upod_od + m->n_upod);
  s->h_pair_upod.assign(m->pair_upod, m->pair_upod + m->n_pair);
  s->h_od_w.assign(m->od_w, m->od_w + (size_t)m->n_od * v.T1);
  s->n_pair = m->n_pair;
  s->n_up = m->n_up;
  v.n_pair = m->n_pair;
  {  // every softmax entry feeds exactly one (turn, od) product: store probabilities in product order
    std::vector<int32_t> ent_pair(std::max(m->n_ent, 1), -1);
    for (int q = 0; q < m->n_pair; ++q) {
      int e = m->pair_ent[q];
      if (e < 0 || e >= m->n_ent || ent_pair[e] != -1) { pedn_destroy(s); return fail(nullptr, PEDN_E_ARG, "pair_ent is not injective"); }
      ent_pair[e] = q;
    }
    // exp(x)/exp(x) == 1 exactly as long as exp(x) is finite and non-zero; |x| <= temp * (|alpha| + |beta|*k_max/8 + |omega| + |eps|)
    const double xmax = fabs(m->pf_temp) * (fabs(m->pf_alpha) + fabs(m->pf_beta) * 16.0 + fabs(m->pf_omega) + fabs(m->pf_eps));
    const bool shortcut = xmax < 600.0;
    std::vector<int32_t> pconst(std::max(m->n_pair, 1), 0);
    for (int g = 0; g < m->n_grp; ++g)
      if (m->grp_ent_ptr[g + 1] - m->grp_ent_ptr[g] == 1 && shortcut && ent_pair[m->grp_ent_ptr[g]] >= 0) pconst[ent_pair[m->grp_ent_ptr[g]]] = 1;
    s->h_pair_const = pconst;
    s->h_turn_pair_ptr.assign(m->turn_pair_ptr, m->turn_pair_ptr + m->n_turns + 1);
    s->h_turn_mode.assign(std::max(m->n_turns, 1), 0);
    for (int tn = 0; tn < m->n_turns; ++tn) {
      bool all_const = true;   // a turn without products is the constant 0
      for (int q = m->turn_pair_ptr[tn]; q < m->turn_pair_ptr[tn + 1]; ++q) all_const = all_const && pconst[q];
      s->h_turn_mode[tn] = all_const ? 1 : 0;
    }
    // row (incoming slot) of its node every softmax group belongs to, read off its products: the product (od, up, down) is
    // summed into the turn (up, down) (path_finder.py:668-686); a group none of whose entries feeds a product is never needed
    std::vector<int> grp_row(std::max(m->n_grp, 1), -1);
    for (int g = 0; g < m->n_grp; ++g) {
      const int n = m->grp_node[g];
      if (n < 0 || n >= N || g < m->node_grp_ptr[n] || g >= m->node_grp_ptr[n + 1]) {
        pedn_destroy(s);
        return fail(nullptr, PEDN_E_ARG, "grp_node / node_grp_ptr inconsistent");
      }
      const int d = m->node_slot_ptr[n + 1] - m->node_slot_ptr[n];
      for (int e = m->grp_ent_ptr[g]; e < m->grp_ent_ptr[g + 1]; ++e) {
        if (ent_pair[e] < 0) continue;
        const int tn = (int)(std::upper_bound(m->turn_pair_ptr, m->turn_pair_ptr + m->n_turns + 1, ent_pair[e]) - m->turn_pair_ptr) - 1;
        const int row = (tn - m->node_turn_ptr[n]) / (d - 1);
        if (tn < m->node_turn_ptr[n] || tn >= m->node_turn_ptr[n + 1] || (grp_row[g] >= 0 && grp_row[g] != row)) {
          pedn_destroy(s);
          return fail(nullptr, PEDN_E_ARG, "products of one softmax group lie in different rows");
        }
        grp_row[g] = row;
      }
    }
    // Rows (incoming slots) of dynamic nodes.  A row whose turns are all mode 1 (every product constant) has fractions that
    // depend on the OD weights only: tabulated and renormalised on the host (tabulate_pair_pod), SlotRec.dyn = 2.  The others
    // get one record each for turn_frac_kernel -- see pedn_types.hpp for the word layout -- and SlotRec.dyn = 1.
    struct Row { int node, i, cost, need, groups; };
    std::vector<Row> rows;
    s->h_slot_dyn.assign(n_slots, 0);
    for (int n = 0; n < N; ++n) {
      if (m->node_kind[n] != 1 || !m->node_dyn[n]) continue;
      const int d = m->node_slot_ptr[n + 1] - m->node_slot_ptr[n];
      for (int i = 0; i < d; ++i) {
        const int ta = m->node_turn_ptr[n] + i * (d - 1), tb = ta + d - 1;
        int need = 0;
        for (int q = m->turn_pair_ptr[ta]; q < m->turn_pair_ptr[tb]; ++q) need += pconst[q] ? 0 : 1;
        s->h_slot_dyn[m->node_slot_ptr[n] + i] = need ? 1 : 2;
        if (!need) continue;
        int cost = m->turn_pair_ptr[tb] - m->turn_pair_ptr[ta], groups = 0;
        for (int g = m->node_grp_ptr[n]; g < m->node_grp_ptr[n + 1]; ++g)
          if (grp_row[g] == i && (m->grp_ent_ptr[g + 1] - m->grp_ent_ptr[g] > 1 || !shortcut)) ++groups;
        cost += 16 * groups;  // a group costs two divisions and an exp per downstream
        rows.push_back(Row{n, i, cost, need, groups});
      }
    }
    // workgroups of four rows, heaviest first (they start first); the rows of a workgroup share its PEDN_TF_LDS_ROWS LDS rows
    std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.cost > b.cost; });
    std::vector<Row> packed;
    std::vector<int> lds_base;
    const int coop_groups = PEDN_TF_COOP_GROUPS;  // rows with more multi-entry groups get a workgroup of their own (2..12 measured: 8 best)
    int lds_limit = PEDN_TF_LDS_ROWS;  // diagnostics: PEDN_TF_LDS_LIMIT=n gives a row at most n LDS rows (the rest goes to ent_p)
    if (const char* d = getenv("PEDN_TF_LDS_LIMIT")) lds_limit = std::max(0, std::min(atoi(d), PEDN_TF_LDS_ROWS));
    {
      std::vector<char> taken(rows.size(), 0);
      for (size_t a0 = 0; a0 < rows.size(); ++a0) {
        if (taken[a0]) continue;
        if (rows[a0].groups > coop_groups) {  // a workgroup of its own: four records of the same row (coop)
          taken[a0] = 1;
          for (int k = 0; k < 4; ++k) { packed.push_back(rows[a0]); lds_base.push_back(0); }
          continue;
        }
        int fill = 0, cnt = 0;
        for (size_t k = a0; k < rows.size() && cnt < 4; ++k) {
          if (taken[k] || rows[k].groups > coop_groups) continue;
          const int need = std::min(rows[k].need, lds_limit);
          if (k != a0 && fill + need > PEDN_TF_LDS_ROWS) continue;
          taken[k] = 1;
          packed.push_back(rows[k]);
          lds_base.push_back(fill);
          fill += need;
          ++cnt;
        }
        for (; cnt < 4; ++cnt) { packed.push_back(Row{-1, 0, 0, 0, 0}); lds_base.push_back(0); }
      }
    }
    s->h_slot_trow.assign(n_slots, -1);
    s->inline_tf_ok = !packed.empty();
    std::vector<int32_t> rwords(std::max<size_t>(packed.size(), 1) * PEDN_TROW_WORDS, 0), gwords;
    std::vector<int32_t> prow(std::max(m->n_pair, 1), -1);  // product -> where its probability is, -1: the constant 1
    int n_over = 0, n_multi = 0;  // n_over: rows of ent_p, for probabilities beyond a workgroup's LDS rows
    auto put_d = [](int32_t* w, double x) { memcpy(w, &x, 8); };
    auto put_f = [](int32_t* w, float x) { memcpy(w, &x, 4); };
    for (size_t ri = 0; ri < packed.size(); ++ri) {
      const int n = packed[ri].node, i = packed[ri].i;
      int32_t* w = &rwords[ri * PEDN_TROW_WORDS];
      if (n < 0) { w[0] = -1; continue; }
      if (ri % 4 != 0 && packed[ri - 1].node == n && packed[ri - 1].i == i) {  // coop: copy of the workgroup's first record
        std::copy(w - PEDN_TROW_WORDS, w, w);
        continue;
      }
      const int s0 = m->node_slot_ptr[n], d = m->node_slot_ptr[n + 1] - s0;
      const int ta = m->node_turn_ptr[n] + i * (d - 1);
      w[0] = d; w[1] = ta; w[2] = n_multi; w[4] = m->turn_pair_ptr[ta]; w[5] = m->turn_pair_ptr[ta + d - 1];
      for (int jj = 0; jj < d - 1; ++jj) {
        w[84 + 3 * jj] = m->turn_pair_ptr[ta + jj]; w[85 + 3 * jj] = m->turn_pair_ptr[ta + jj + 1]; w[86 + 3 * jj] = s->h_turn_mode[ta + jj];
      }
      std::vector<int> used;  // outgoing slots this row's multi-entry groups refer to
      int n_prob = 0, n_g = 0, any_sep = 0, over = 0;
      const int lds_rows = std::min(packed[ri].need, lds_limit);
      for (int g = m->node_grp_ptr[n]; g < m->node_grp_ptr[n + 1]; ++g) {
        const int a = m->grp_ent_ptr[g], b = m->grp_ent_ptr[g + 1];
        if (grp_row[g] != i || b - a < 1 || (b - a == 1 && shortcut)) continue;
        double sumd = 0.0;
        for (int e = a; e < b; ++e) sumd = (e == a) ? m->ent_dist[e] : sumd + m->ent_dist[e];
        gwords.resize(gwords.size() + 32, 0);
        int32_t* G = &gwords[gwords.size() - 32];
        G[0] = b - a; G[1] = m->grp_allphys[g];
        for (int k = 0; k < PEDN_MAX_DEGREE - 1; ++k) G[2 + k] = G[9 + k] = -1;
        for (int e = a; e < b; ++e) {
          const int k = e - a, l = m->ent_link[e];
          if (l >= 0) {
            int slot = -1;
            for (int j = 0; j < d; ++j)
              if (l < L && m->slot_out_link[s0 + j] == l) slot = j;
            if (slot < 0) { pedn_destroy(s); return fail(nullptr, PEDN_E_ARG, "softmax entry is not an outgoing link of its node"); }
            size_t u = std::find(used.begin(), used.end(), slot) - used.begin();
            if (u == used.size()) used.push_back(slot);
            G[2 + k] = (int)u;
          }
          if (ent_pair[e] >= 0) {
            if (n_prob < lds_rows) G[9 + k] = lds_base[ri] + n_prob;
            else { G[9 + k] = PEDN_TF_LDS_ROWS + n_over++; over = 1; }
            ++n_prob;
            prow[ent_pair[e]] = G[9 + k];
          }
          put_d(&G[16 + 2 * k], (m->pf_alpha * m->ent_dist[e]) / (sumd + 1e-6));
        }
        ++n_g; ++n_multi;
      }
      if (used.size() > (size_t)(PEDN_MAX_DEGREE - 1)) { pedn_destroy(s); return fail(nullptr, PEDN_E_ARG, "a row refers to more than 7 downstream links"); }
      for (int e = 0; e < PEDN_MAX_DEGREE - 1; ++e) {
        const int slot = used.empty() ? -1 : used[e < (int)used.size() ? e : 0];
        int32_t* E = e < 5 ? &w[8 + 10 * e] : &w[64 + 10 * (e - 5)];
        const int l = slot >= 0 ? m->slot_out_link[s0 + slot] : 0;  // only rows with groups come here, and L > 0 then
        E[0] = l; E[1] = m->link_rev[l]; E[2] = m->link_sep[l];
        put_f(&E[3], (float)(m->link_length[l] * m->link_width[l]));
        put_d(&E[4], m->link_vf[l]); put_d(&E[6], m->link_kc[l]); put_d(&E[8], m->link_length[l]);
        if (slot >= 0 && E[2]) any_sep = 1;
      }
      w[3] = n_g; w[6] = (int)used.size(); w[7] = any_sep; w[107] = over; w[108] = packed[ri].groups > coop_groups;
      w[109] = lds_base[ri];
      s->h_slot_trow[s0 + i] = (int)ri;
      if (over || packed[ri].groups > coop_groups || lds_rows > PEDN_TF_INL_ROWS) s->inline_tf_ok = false;
    }
    gwords.resize(gwords.size() + 8 * 32, 0);  // turn_frac_body reads a chunk of four records ahead; padding has n = 0
    rows = packed;
    {  // the workgroups (quads of rows, heaviest first) whose dependent chains are longer than a link-update workgroup lasts: inside
       // link_turn_kernel they are dispatched in front of the link update, the short ones behind it (launch_step)
      int heavy_groups = 2;
      if (const char* d = getenv("PEDN_TF_HEAVY_GROUPS")) heavy_groups = atoi(d);
      s->n_tf_heavy_quads = 0;
      for (size_t q = 0; q * 4 < rows.size(); ++q)
        if (rows[q * 4].groups > heavy_groups) s->n_tf_heavy_quads = (int)q + 1;
    }
    v.n_multi = n_multi;
    v.n_trow = (int)rows.size();
    s->n_over = n_over;
    TRY(upload(s, rwords.data(), rwords.size(), &v.trow_words));
    TRY(upload(s, gwords.data(), gwords.size(), &v.tgrp_words));
    TRY(upload(s, prow.data(), prow.size(), &v.pair_row));
    for (int q = 0; q < m->n_pair; ++q)
      if (!pconst[q] && prow[q] < 0) { pedn_destroy(s); return fail(nullptr, PEDN_E_ARG, "(turn, od) product without a softmax entry on a dynamic node"); }
    v.n_turns = m->n_turns;
  }
  TRY(upload(s, m->od_w, (size_t)m->n_od * v.T1, &v.od_w));
  {  // corridors: one lane of link_kernel updates both directions
    std::vector<CorrRec> cr;
    for (int l = 0; l < L; ++l)
      if (l < m->link_rev[l]) {
        CorrRec c{};
        c.a = l; c.b = m->link_rev[l]; c.Pa = lp[c.a]; c.Pb = lp[c.b];
        cr.push_back(c);
      }
    v.n_pairs_corr = (int)cr.size();
    v.pairs_adj = 1;
    for (size_t p = 0; p < cr.size(); ++p)
      if (cr[p].a != 2 * (int)p || cr[p].b != 2 * (int)p + 1) v.pairs_adj = 0;
    if (const char* f = getenv("PEDN_PAIRS_ADJ")) if (atoi(f) == 0) v.pairs_adj = 0;   // diagnostic: always through the record
    TRY(upload(s, cr.data(), cr.size(), &v.corr_rec));
  }
  {  // bin nodes into blocks of 8 waves: first-fit over the nodes ordered by (expected load, slot count) decreasing, so that
    // every block is full regardless of node degree.  The 8 waves of a block meet at two barriers, so a block lasts as long
    // as its slowest wave, and the slow waves are those of busy links (look-backs, diffusion, binomial draws).  The junctions
    // that many OD routes pass -- those with many (turn, od) products -- are the likely ones and go first (delft: 32.3 ->
    // 31.6 us; PEDN_PACK_BY_LOAD=0 orders by slot count only).  Re-packing at run time by the cost the waves measure
    // themselves (ticks up to the first barrier, sampled every 100 / 250 steps) was tried and changed nothing (30.5 us either
    // way): the spread inside a block comes from step-to-step variation, not from which junctions share it.
    // Better than the estimate: the MEASURED cost of every node's slowest wave (pedn_model_desc.node_cost, from a calibration run of
    // the profiling build, tools/pack_calibrate.py).  tools/pack_analysis.py: the wait at the first barrier is static (92 % of it is
    // explained by the waves' mean arrival times); half of it is a node's waves waiting for each other -- inherent --, the other half is
    // the packing's, and packing by measured cost removes at most half of that: 3.4-3.9 % of a wave's life.
    // PEDN_PACK_BY_LOAD=0: by degree only, =1: by the static estimate even when a measured cost is given.
    std::vector<int> order(N);
    std::vector<double> load(N, 0.0);
    for (int n = 0; n < N; ++n) order[n] = n;
    auto deg = [&](int n) { return m->node_slot_ptr[n + 1] - m->node_slot_ptr[n]; };
    int by_load = m->node_cost ? 2 : 1;
    if (const char* f = getenv("PEDN_PACK_BY_LOAD")) by_load = std::min(atoi(f), by_load);
    if (by_load == 2)
      for (int n = 0; n < N; ++n) load[n] = (double)m->node_cost[n];
    else if (by_load == 1)
      for (int n = 0; n < N; ++n) load[n] = m->turn_pair_ptr[m->node_turn_ptr[n + 1]] - m->turn_pair_ptr[m->node_turn_ptr[n]];
    s->packed_by = by_load;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] != load[b] ? load[a] > load[b] : deg(a) > deg(b); });
    std::vector<std::vector<int>> bins;
    std::vector<int> fill;
    for (int n : order) {
      int d = deg(n), chosen = -1;
      for (size_t b = 0; b < bins.size(); ++b)
        if (fill[b] + d <= 8) { chosen = (int)b; break; }
      if (chosen < 0) { bins.emplace_back(); fill.push_back(0); chosen = (int)bins.size() - 1; }
      bins[chosen].push_back(n);
      fill[chosen] += d;
    }
    SlotRec idle{};
    idle.node = -1;
    idle.trow = -1;
    std::vector<SlotRec> rec((bins.size() + 1) * 8, idle);   // one more block of idle records behind the bins: prewarm_chains
    for (size_t b = 0; b < bins.size(); ++b) {
      int wave = 0, base = 0;
      for (int n : bins[b]) {
        int d = deg(n);
        for (int k = 0; k < d; ++k) {
          SlotRec& R = rec[b * 8 + wave++];
          R.node = n; R.slot = k; R.base = base; R.m = d;
          R.kind = m->node_kind[n]; R.dyn = s->h_slot_dyn[m->node_slot_ptr[n] + k];
          R.trow = s->h_slot_trow.empty() ? -1 : s->h_slot_trow[m->node_slot_ptr[n] + k];
          R.lin = m->slot_in_link[m->node_slot_ptr[n] + k];
          R.lout = m->slot_out_link[m->node_slot_ptr[n] + k];
          R.turn0 = m->node_turn_ptr[n];
          R.demand_row = m->node_demand_row[n];
          if (R.lin < L) { R.Pin = lp[R.lin]; R.Pout = lp[R.lout]; }
        }
        base += d * d;  // <= 64 tiles because sum(d) <= 8
      }
    }
    s->n_blocks = (int)bins.size();
    for (int n = 0; n < N; ++n) s->max_degree = std::max(s->max_degree, deg(n));
    if (const char* f = getenv("PEDN_NODE_MD")) if (atoi(f) == 8) s->max_degree = 8;  // diagnostic: the general instantiation
    {  // LDS of node_kernel: 8 (LP: 16) rows of 64 doubles + the m*m tiles of the fullest block
      int tiles = 1;
      for (const auto& bin : bins) {
        int t2 = 0;
        for (int n : bin) t2 += deg(n) * deg(n);
        tiles = std::max(tiles, t2);
      }
      s->node_lds = (size_t)((m->node_model == PEDN_NODE_OPTIMAL ? 16 : 8) + tiles) * 64 * sizeof(double);
    }
    // (register budget of the node kernels: node_kernel_waves() in pedn_kernels.hpp)
    // The link update of t and the turning fractions of t+1 share one launch (both only read what node_kernel(t) and earlier
    // launches wrote); PEDN_FUSE_TP=0 gives the fractions a launch of their own in front of node_kernel(t+1).
    s->fuse_tp = 1;
    if (const char* f = getenv("PEDN_FUSE_TP")) s->fuse_tp = atoi(f) != 0;
    if (const char* f = getenv("PEDN_FUSE_OBS")) s->fuse_obs = atoi(f) != 0;
    for (SlotRec& R : rec) { R.act = -1; R.lp = -1; }
    // two chains of launches (one per half of the replicas) in pedn_run; PEDN_STREAMS=1|2, pedn_set_streams.  Default from 640
    // replicas: one chain's launch leaves wave places empty while it fills and drains, and the other half's launches use them
    // (delft x 1024: 51.7 -> 45.3 us per step in round 2; round 5, melbourne / delft: x 640 25.8 -> 23.6 / 36.2 -> 34.3, x 768 27.1 -> 24.8,
    // x 896 31.3 -> 27.5 / 45.3 -> 40.7; x 512 19.2 either way).  The halves are whole 128-replica segments (view_of): 640 = 384 + 256.
    s->chains = v.RS >= 640 ? 2 : 1;
    if (const char* f = getenv("PEDN_STREAMS")) s->chains = atoi(f) == 2 ? 2 : 1;
    s->two_streams = s->chains > 1;
    // Owner-wave plan of pedn_run (launch_step: lazy): node_kernel<LU>(t + 1) performs the link update of t, one launch per step.  The
    // default for models whose second launch is the link update alone (no turning fractions computed on the device): melbourne x 1024
    // 37.4 -> 32.7-33.7 us per step on one chain, 34.6 -> 30.3 on two (profiles/r04_owner_wave.txt).  With dynamic rows the second launch
    // stays (node_kernel(t + 1) needs the fractions that need node_kernel(t)'s flows) and the longer node_kernel costs more than the
    // link-update workgroups that leave it save (delft x 1024: 42.1-42.6 -> 44.2).  PEDN_LINK_OWNER=0|1 overrides.
    s->link_owner = v.n_trow == 0 && m->node_model != PEDN_NODE_OPTIMAL;
    if (const char* f = getenv("PEDN_LINK_OWNER")) s->link_owner = atoi(f) != 0;
    // Single-launch plan with dynamic turning fractions: the rows computed inside node_kernel<LU, TF> by the slot waves themselves.  That
    // instantiation is built for 2 waves per SIMD (1 block per CU): only where the whole grid is one generation of them -- small
    // batches, which are chains of latencies and gain a whole launch (nine_intersections x 256: 17.2 -> ... us per step).
    s->node_lds_tf = s->node_lds + (size_t)8 * PEDN_TF_INL_LDS * sizeof(double);
    v.tf_lds_off = (int32_t)(s->node_lds / sizeof(double));
    bool inl_possible = s->inline_tf_ok && m->node_model != PEDN_NODE_OPTIMAL && s->node_lds_tf <= 160 * 1024;
    if (inl_possible && s->node_lds_tf > 64 * 1024) {   // more dynamic LDS than a kernel gets by default: ask for it (gfx950: 160 KB per CU)
      for (int pr = 0; pr < 2 && inl_possible; ++pr) {
        DevView& vv = s->v;
        const int keep = vv.pr;
        vv.pr = pr;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel_for(s, true, true)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->node_lds_tf) != hipSuccess)
          inl_possible = false;
        vv.pr = keep;
      }
      (void)hipGetLastError();
    }
    // ... by HELPER waves (node_kernel_h: sixteen waves per workgroup at 128 VGPRs, still one workgroup per CU): the row (~5 us) and the
    // slot wave's own chain (~5 us) side by side -- nine_intersections x 256 16.2 -> 12.5 us per step, 45_intersections x 1024 (272
    // workgroups) 17.9 -> 13.6, x 1280 (340) 20.4 -> 14.7 under pedn_run's two chains and 20.5 -> 19.1 step by step; x 1536 (408) still
    // wins under pedn_run (21.3 -> 17.5) but not step by step (22.3 -> 23.1), at 544 (x 2048) two launches per step are faster either
    // way (22.9 against 23.7).  The default up to 352 workgroups; PEDN_INLINE_TF=0 two launches, 1 the slot waves' own rows, 2 helper
    // waves (forced whatever the grid).  profiles/r04_inline_helpers.txt
    s->inline_tf = inl_possible && (size_t)(v.RS / 64) * (size_t)s->n_blocks <= 352;
    s->inline_help = s->inline_tf;
    if (const char* f = getenv("PEDN_INLINE_TF")) { s->inline_tf = atoi(f) != 0 && inl_possible; s->inline_help = atoi(f) == 2 && inl_possible; }
    if (s->inline_help && s->node_lds_tf > 64 * 1024) {
      for (int pr = 0; pr < 2; ++pr) {
        DevView& vv = s->v;
        const int keep = vv.pr;
        vv.pr = pr;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel_for(s, true, true)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->node_lds_tf) != hipSuccess)
          s->inline_help = 0;
        vv.pr = keep;
      }
      (void)hipGetLastError();
    }
    if (s->inline_tf) s->link_owner = 1;
    // 0 = by batch: two chains where the step is not a pure chain of latencies any more -- from 4096 envs, and from 1024 with per-env
    // scenarios (45_intersections: 2048 envs plain 24.8-25.2 -> 24.7-25.7 us per env step, randomised 27.5-27.9 -> 25.6-26.0;
    // 4096 envs 40.6 -> 35.9; profiles/r04_rl_chains.txt); PEDN_RL_CHAINS=1|2 forces
    s->rl_chains = 0;
    if (const char* f = getenv("PEDN_RL_CHAINS")) s->rl_chains = atoi(f) == 2 ? 2 : 1;
    s->node_lp = m->node_model == PEDN_NODE_OPTIMAL;
    if (s->node_lp) {  // tableau workspace: one per (regular node, replica group), sized for the largest such node
      int n_lp = 0, max_m = 0;
      for (SlotRec& R : rec)
        if (R.node >= 0 && R.kind == 1 && R.slot == 0) { R.lp = n_lp++; max_m = std::max(max_m, R.m); }
      const int E = max_m * (max_m - 1), rows = 2 * max_m + E, cols = 3 * E + 2 * max_m;
      v.lp_stride = (size_t)(rows + 1) * (cols + 1) * 64;
      v.lp_bstride = (size_t)rows * 64;
      const size_t waves = (size_t)std::max(n_lp, 1) * (v.RS / 64);
      TRY(dalloc(s, waves * v.lp_stride, &v.lp_ws));
      TRY(dalloc(s, waves * v.lp_bstride, &v.lp_basis));
    }
    s->h_slot_rec = rec;
    TRY(upload(s, rec.data(), rec.size(), &v.slot_rec));
    s->d_slot_rec = const_cast<SlotRec*>(v.slot_rec);
  }
  // ---- dynamic state
  {
    size_t RS = v.RS, T1 = v.T1;
    for (int f = 0; f < 7; ++f) TRY(dalloc(s, (size_t)s->rows64[f] * (f < 4 ? v.Lall : L) * RS, &v.f64[f]));
    for (int g = 0; g < 6; ++g) TRY(dalloc(s, (size_t)s->rows32[g] * L * RS, &v.f32[g]));
    TRY(dalloc(s, (size_t)L * RS, &v.rsum));
    TRY(dalloc(s, (size_t)L * RS, &v.front));
    TRY(dalloc(s, (size_t)L * RS, &v.back));
    TRY(dalloc(s, (size_t)L * RS, &v.sepw));
    TRY(dalloc(s, (size_t)L * RS, &v.sepnp));
    HIP_TRY(s, hipMemset(v.sepnp, 0, std::max<size_t>((size_t)L * RS, 1) * sizeof(double)));
    TRY(dalloc(s, (size_t)m->n_turns * RS, &v.tf));
    TRY(dalloc(s, (size_t)m->n_demand * T1 * RS, &v.demand));
    TRY(dalloc(s, (size_t)std::max(s->n_over, 1) * RS, &v.ent_p));  // probabilities that do not fit a node's LDS rows
    for (int k = 0; k < 2; ++k) TRY(dalloc(s, (size_t)std::max(m->n_turns, 1) * RS, &v.tfd[k]));
    TRY(dalloc(s, (size_t)m->n_pair * T1, &s->d_pair_pod));
    v.pair_pod = s->d_pair_pod;
    TRY(dalloc(s, (size_t)std::max(m->n_turns, 1) * T1, &s->d_turn_tab));
    v.turn_tab = s->d_turn_tab;
    HIP_TRY(s, hipMemset(s->d_turn_tab, 0, (size_t)std::max(m->n_turns, 1) * T1 * sizeof(double)));
    TRY(dalloc(s, RS, &v.flags));
    TRY(dalloc(s, 4, &s->d_clock));
    HIP_TRY(s, hipMemset(s->d_clock, 0, 4 * sizeof(int32_t)));
    v.clock = s->d_clock;
  }
  // initial widths, turning fractions, demand (broadcast to every replica)
  {
    const double* w0[3] = {m->front_gate0, m->back_gate0, m->sep_width0};
    double* dst[3] = {v.front, v.back, v.sepw};
    for (int k = 0; k < 3; ++k) TRY(push_rows(s, dst[k], w0[k], L, 0, 1, PEDN_ALL));
    TRY(push_rows(s, v.tf, m->tf_init, m->n_turns, 0, 1, PEDN_ALL));
    TRY(push_rows(s, v.demand, m->demand, m->n_demand * v.T1, 0, 1, PEDN_ALL));
    HIP_TRY(s, hipMemsetAsync(v.ent_p, 0, (size_t)std::max(s->n_over, 1) * v.RS * 8, s->stream));
    for (int k = 0; k < 2; ++k) HIP_TRY(s, hipMemsetAsync(v.tfd[k], 0, (size_t)std::max(m->n_turns, 1) * v.RS * 8, s->stream));
    // replica-uniform shortcuts: everything starts uniform; dynamic nodes always use their per-replica rows
    const double qnan = __builtin_nan("");
    s->h_front_u.assign(m->front_gate0, m->front_gate0 + L);
    s->h_back_u.assign(m->back_gate0, m->back_gate0 + L);
    s->h_tf_u.assign(m->tf_init, m->tf_init + m->n_turns);
    s->h_node_dyn.assign(m->node_dyn, m->node_dyn + N);
    for (int n = 0; n < N; ++n)
      if (m->node_dyn[n])
        for (int k = m->node_turn_ptr[n]; k < m->node_turn_ptr[n + 1]; ++k) s->h_tf_u[k] = qnan;
    TRY(dalloc(s, (size_t)L, &s->d_front_u));
    TRY(dalloc(s, (size_t)L, &s->d_back_u));
    TRY(dalloc(s, (size_t)m->n_turns, &s->d_tf_u));
    v.front_u = s->d_front_u; v.back_u = s->d_back_u; v.tf_u = s->d_tf_u;
    TRY(push_uniform(s));
    TRY(tabulate_pair_pod(s));
  }
  {
    int rc = reset_state(s);
    if (rc != PEDN_OK) { std::string keep = g_last_error; pedn_destroy(s); g_last_error = keep; return rc; }
  }
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (s->chains > 1) {
    int got = 1;
    TRY(warm_chain_streams(s, &got));
    s->chains = s->warmed_chains = got;
    s->two_streams = s->chains > 1;
    prewarm_chains(s);
  }
#undef TRY
  *out = s;
  return PEDN_OK;
}

int pedn_destroy(pedn_sim* s) {
  if (!s) return PEDN_OK;
  hipSetDevice(s->device);
  if (s->clocked) hipDeviceSynchronize();   // clocked steps may still run on a caller's stream (or in a graph being replayed there)
  if (s->stream2) hipStreamSynchronize(s->stream2);
  if (s->stream) hipStreamSynchronize(s->stream);
  for (void* p : s->allocs) hipFree(p);
  if (s->rl_pin) hipHostFree(s->rl_pin);
  for (pedn_sim::Stage& st : s->stage) {
    if (st.dev) hipFree(st.dev);
    if (st.pin) hipHostFree(st.pin);
    if (st.done) hipEventDestroy(st.done);
  }
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->stream2) hipStreamDestroy(s->stream2);
  if (s->ev_fork) hipEventDestroy(s->ev_fork);
  if (s->ev_join) hipEventDestroy(s->ev_join);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
  return PEDN_OK;
}

int pedn_reset(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);
  s->tp_ready = -1;
  s->last_t = -1;
  s->link_pending = -1;   // discarded: the state it would complete is being cleared
  return reset_state(s);
}

int pedn_reset_lazy(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);
  s->tp_ready = -1;
  s->last_t = -1;
  s->link_pending = -1;
  return reset_state_lazy(s);
}

// values (host) -> rows of a [rows][RS] device array, one replica or all
static int push_rows(pedn_sim* s, double* dst, const double* values, int n_rows, size_t row0, size_t row_stride, int replica) {
  if (n_rows <= 0) return PEDN_OK;
  DevView& v = s->v;
  if (replica != PEDN_ALL && (replica < 0 || replica >= v.R)) return fail(s, PEDN_E_ARG, "replica out of range");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);       // every setter that goes through here (demand, turning fractions, widths) is ordered behind both chains
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, (size_t)n_rows * 8, &st);
  if (rc != PEDN_OK || (rc = stage_upload(s, st, values, (size_t)n_rows * 8)) != PEDN_OK) return rc;
  int r0 = replica == PEDN_ALL ? 0 : replica, r1 = replica == PEDN_ALL ? v.RS : replica + 1;
  size_t n = (size_t)n_rows * (r1 - r0);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst, (const double*)st->dev, n_rows,
                     row0, row_stride, v.RS, r0, r1, 0, 1);
  HIP_TRY(s, hipGetLastError());
  return stage_commit(s, st);
}

int pedn_set_demand(pedn_sim* s, int32_t node, int32_t replica, const double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  std::vector<double> full(s->v.T1, 0.0);  // copied into the pinned staging buffer before push_rows returns
  for (int i = 0; i < n && i < s->v.T1; ++i) full[i] = values[i];
  return push_rows(s, s->v.demand, full.data(), s->v.T1, (size_t)row * s->v.T1, 1, replica);
}

int pedn_get_demand(pedn_sim* s, int32_t node, int32_t replica, double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  const int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  DevView& v = s->v;
  if (replica < 0 || replica >= v.R) return fail(s, PEDN_E_ARG, "replica out of range");
  if (n < 0 || n > v.T1) return fail(s, PEDN_E_ARG, "more values than time indices");
  if (n == 0) return PEDN_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy2D(values, 8, v.demand + (size_t)row * v.T1 * v.RS + replica, (size_t)v.RS * 8, 8, n, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

int pedn_set_demand_matrix(pedn_sim* s, int32_t node, const double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  const int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  DevView& v = s->v;
  if (n < 0 || n > v.T1) return fail(s, PEDN_E_ARG, "more demand values than time indices");
  if (n == 0) return PEDN_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  const size_t bytes = (size_t)v.R * n * 8;
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, bytes, &st);
  if (rc != PEDN_OK || (rc = stage_upload(s, st, values, bytes)) != PEDN_OK) return rc;
  const size_t lanes = (size_t)v.T1 * v.RS;
  hipLaunchKernelGGL(demand_matrix_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s->stream, v.demand, (const double*)st->dev,
                     (size_t)row, n, v.T1, v.R, v.RS);
  HIP_TRY(s, hipGetLastError());
  return stage_commit(s, st);
}

int pedn_set_demand_rows(pedn_sim* s, int32_t node, const int32_t* replicas, int32_t n_rep, const double* values, int32_t n) {
  if (!s || !values || !replicas) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  const int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  DevView& v = s->v;
  if (n < 0 || n > v.T1) return fail(s, PEDN_E_ARG, "more demand values than time indices");
  for (int k = 0; k < n_rep; ++k)
    if (replicas[k] < 0 || replicas[k] >= v.R) return fail(s, PEDN_E_ARG, "replica out of range");
  if (n == 0 || n_rep <= 0) return PEDN_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  // staging layout: values [n_rep][n] (f64), then the replica ids (int32)
  const size_t vbytes = (size_t)n_rep * n * 8, bytes = vbytes + (size_t)n_rep * 4;
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, bytes, &st);
  if (rc != PEDN_OK || (rc = stage_upload(s, st, values, vbytes)) != PEDN_OK || (rc = stage_upload(s, st, replicas, (size_t)n_rep * 4, vbytes)) != PEDN_OK)
    return rc;
  const size_t lanes = (size_t)v.T1 * n_rep;
  hipLaunchKernelGGL(demand_rows_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s->stream, v.demand, (const double*)st->dev,
                     (const int32_t*)((const char*)st->dev + vbytes), n_rep, (size_t)row, n, v.T1, v.RS);
  HIP_TRY(s, hipGetLastError());
  return stage_commit(s, st);
}

int pedn_draw_demand(pedn_sim* s, int32_t node, uint64_t seed, const int32_t* pattern, const double* base, const double* peak,
                     const int32_t* spike_start, const int32_t* spike_len, const double* spike_height) {
  if (!s || !pattern || !base || !peak || !spike_start || !spike_len || !spike_height) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  const int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  DevView& v = s->v;
  for (int r = 0; r < v.R; ++r) {
    if (pattern[r] < 0 || pattern[r] > 2) return fail(s, PEDN_E_ARG, "demand pattern outside 0..2");
    if (!(base[r] >= 0.0) || !(peak[r] >= 0.0) || base[r] + 2.0 * peak[r] > 500.0) return fail(s, PEDN_E_ARG, "demand rate outside [0, 500]");
  }
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  // staging layout: pattern, spike_start, spike_len (int32 [R] each, padded to 8 bytes), then base, peak, spike_height (f64 [R])
  const size_t R = (size_t)v.R, ioff = ((3 * R * 4 + 7) / 8) * 8, bytes = ioff + 3 * R * 8;
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, bytes, &st);
  if (rc != PEDN_OK) return rc;
  unsigned char* h = (unsigned char*)st->pin;
  memcpy(h, pattern, R * 4);
  memcpy(h + R * 4, spike_start, R * 4);
  memcpy(h + 2 * R * 4, spike_len, R * 4);
  memcpy(h + ioff, base, R * 8);
  memcpy(h + ioff + R * 8, peak, R * 8);
  memcpy(h + ioff + 2 * R * 8, spike_height, R * 8);
  HIP_TRY(s, hipMemcpyAsync(st->dev, st->pin, bytes, hipMemcpyHostToDevice, s->stream));
  const unsigned char* d = (const unsigned char*)st->dev;
  const size_t lanes = (size_t)v.T1 * v.RS;
  hipLaunchKernelGGL(draw_demand_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s->stream, v.demand, (size_t)row, v.T1, v.R,
                     v.RS, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), v.replica_offset, (uint32_t)node,
                     (const int32_t*)d, (const double*)(d + ioff), (const double*)(d + ioff + R * 8), (const int32_t*)(d + R * 4),
                     (const int32_t*)(d + 2 * R * 4), (const double*)(d + ioff + 2 * R * 8));
  HIP_TRY(s, hipGetLastError());
  return stage_commit(s, st);
}

int pedn_set_od_weights(pedn_sim* s, int32_t od, const double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (od < 0 || od >= s->n_od || n != s->v.T1) return fail(s, PEDN_E_ARG, "od index or length out of range");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy((void*)(s->v.od_w + (size_t)od * s->v.T1), values, (size_t)n * 8, hipMemcpyHostToDevice));
  std::copy(values, values + n, s->h_od_w.begin() + (size_t)od * s->v.T1);
  s->tp_ready = -1;  // fractions of the next step may already sit in tfd, computed with the old weights
  return tabulate_pair_pod(s);
}

int pedn_set_turning_fractions(pedn_sim* s, int32_t node, int32_t replica, const double* tf, int32_t n) {
  if (!s || !tf) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  int a = s->node_turn_ptr[node], b = s->node_turn_ptr[node + 1];
  if (n != b - a) return fail(s, PEDN_E_ARG, "turning-fraction count does not match m(m-1)");
  int rc = push_rows(s, s->v.tf, tf, n, (size_t)a, 1, replica);
  if (rc != PEDN_OK) return rc;
  // a dynamic node recomputes its fractions every step (network.py:272-275); until then a read returns what was imposed
  s->h_tf_set_epoch[node] = s->step_epoch;
  // (the imposed values go into the buffer of the last step too, see below: a REPEAT of that step must recompute them, not take them
  // for its own -- under the single-launch plan tp_ready names that very step; found by tools/gpu_fuzz_plans.py)
  if (s->h_node_dyn[node]) s->tp_ready = -1;
  if (s->h_node_dyn[node] && s->last_t >= 0 && (rc = push_rows(s, s->v.tfd[s->last_t & 1], tf, n, (size_t)a, 1, replica)) != PEDN_OK) return rc;
  for (int k = 0; k < n; ++k)
    s->h_tf_u[a + k] = (replica == PEDN_ALL && !s->h_node_dyn[node]) ? tf[k] : __builtin_nan("");
  return push_uniform(s);
}

int pedn_get_turning_fractions(pedn_sim* s, int32_t node, int32_t replica, double* tf, int32_t n) {
  if (!s || !tf) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  if (replica < 0 || replica >= s->v.R) return fail(s, PEDN_E_ARG, "replica out of range");
  int a = s->node_turn_ptr[node], b = s->node_turn_ptr[node + 1];
  if (n != b - a) return fail(s, PEDN_E_ARG, "turning-fraction count does not match m(m-1)");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  const double* src = (s->h_node_dyn[node] && s->last_t >= 0) ? s->v.tfd[s->last_t & 1] : s->v.tf;
  HIP_TRY(s, hipMemcpy2D(tf, 8, src + (size_t)a * s->v.RS + replica, (size_t)s->v.RS * 8, 8, n, hipMemcpyDeviceToHost));
  if (s->v.pod_pr && s->ttab_r_stale) {  // the per-replica tables were derived on the device: fetch the host copy once
    s->h_ttab_r.assign((size_t)std::max(s->n_turns, 1) * s->v.R, 0.0);
    HIP_TRY(s, hipMemcpy2D(s->h_ttab_r.data(), (size_t)s->v.R * 8, s->d_turn_tab_r, (size_t)s->v.RS * 8, (size_t)s->v.R * 8, std::max(s->n_turns, 1), hipMemcpyDeviceToHost));
    s->ttab_r_stale = false;
  }
  if (s->h_node_dyn[node] && s->last_t >= 0 && s->h_tf_set_epoch[node] != s->step_epoch) {  // rows tabulated on the host (SlotRec.dyn == 2)
    const int s0 = s->h_node_slot_ptr[node], d = s->h_node_slot_ptr[node + 1] - s0;
    for (int i = 0; i < d; ++i)
      if (s->h_slot_dyn[s0 + i] == 2)
        for (int jj = 0; jj < d - 1; ++jj) {
          const int tn = a + i * (d - 1) + jj;
          tf[tn - a] = s->v.pod_pr ? s->h_ttab_r[(size_t)tn * s->v.R + replica] : s->h_ttab[(size_t)s->last_t * s->n_turns + tn];
        }
  }
  return PEDN_OK;
}

int pedn_set_width(pedn_sim* s, int32_t which, int32_t link, int32_t replica, double value) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (link < 0 || link >= s->v.L || which < 0 || which > 3) return fail(s, PEDN_E_ARG, "link or selector out of range");
  if (replica != PEDN_ALL && (replica < 0 || replica >= s->v.R)) return fail(s, PEDN_E_ARG, "replica out of range");
  double* dst = which == PEDN_W_FRONT ? s->v.front : which == PEDN_W_BACK ? s->v.back : which == PEDN_W_SEP ? s->v.sepw : s->v.sepnp;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);     // the link update records the gate / reads the separator width of its own step
  if (which == PEDN_W_BACK) s->tp_ready = -1;  // capacity fallback of the turn probabilities (path_finder.py:575-576)
  join_forked(s);   // (as every setter: ordered behind both chains)
  // one launch, its values as kernel arguments: the row's replicas and the link's entry of the replica-uniform shortcut (front_u / back_u:
  // NaN = "differs between replicas"; a link the RL agents write per replica keeps NaN, push_uniform)
  double* uni = nullptr;
  double uval = 0.0;
  if (which <= PEDN_W_BACK) {
    std::vector<double>& hu = which == PEDN_W_FRONT ? s->h_front_u : s->h_back_u;
    hu[link] = replica == PEDN_ALL ? value : __builtin_nan("");
    uval = (link < (int)s->h_rl_link.size() && s->h_rl_link[link]) ? __builtin_nan("") : hu[link];
    uni = (which == PEDN_W_FRONT ? s->d_front_u : s->d_back_u) + link;
  }
  const int r0 = replica == PEDN_ALL ? 0 : replica, n = replica == PEDN_ALL ? s->v.RS : 1;
  hipLaunchKernelGGL(set_width_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst + (size_t)link * s->v.RS, r0, n, value, uni, uval);
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_set_widths(pedn_sim* s, int32_t which, const double* values) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (which < 0 || which > 3) return fail(s, PEDN_E_ARG, "selector out of range");
  DevView& v = s->v;
  if (which == PEDN_W_BACK) s->tp_ready = -1;
  if (v.L == 0) return PEDN_OK;
  double* dst = which == PEDN_W_FRONT ? v.front : which == PEDN_W_BACK ? v.back : which == PEDN_W_SEP ? v.sepw : v.sepnp;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  size_t bytes = (size_t)v.L * v.R * 8;
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, bytes, &st);
  if (rc != PEDN_OK || (rc = stage_upload(s, st, values, bytes)) != PEDN_OK) return rc;
  size_t n = (size_t)v.L * v.R;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst, (const double*)st->dev, v.L,
                     (size_t)0, (size_t)1, v.RS, 0, v.R, 1, v.R);
  HIP_TRY(s, hipGetLastError());
  if ((rc = stage_commit(s, st)) != PEDN_OK) return rc;
  if (which <= PEDN_W_BACK) {
    std::vector<double>& u = which == PEDN_W_FRONT ? s->h_front_u : s->h_back_u;
    for (int l = 0; l < v.L; ++l) {
      bool same = true;
      for (int r = 1; r < v.R && same; ++r) same = values[(size_t)l * v.R + r] == values[(size_t)l * v.R];
      u[l] = same ? values[(size_t)l * v.R] : __builtin_nan("");
    }
    return push_uniform(s);
  }
  return PEDN_OK;
}

int pedn_reset_widths(pedn_sim* s, const double* front, const double* back, const double* sep) {
  if (!s || !front || !back || !sep) return fail(s, PEDN_E_ARG, "null argument");
  DevView& v = s->v;
  s->tp_ready = -1;
  if (v.L == 0) return PEDN_OK;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  int rc;
  if ((rc = push_rows(s, v.front, front, v.L, 0, 1, PEDN_ALL)) || (rc = push_rows(s, v.back, back, v.L, 0, 1, PEDN_ALL)) ||
      (rc = push_rows(s, v.sepw, sep, v.L, 0, 1, PEDN_ALL)))
    return rc;
  HIP_TRY(s, hipMemsetAsync(v.sepnp, 0, (size_t)v.L * v.RS * sizeof(double), s->stream));
  s->h_front_u.assign(front, front + v.L);
  s->h_back_u.assign(back, back + v.L);
  return push_uniform(s);
}

int pedn_get_widths(pedn_sim* s, int32_t which, double* values) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (which < 0 || which > 3) return fail(s, PEDN_E_ARG, "selector out of range");
  DevView& v = s->v;
  if (v.L == 0) return PEDN_OK;
  const double* src = which == PEDN_W_FRONT ? v.front : which == PEDN_W_BACK ? v.back : which == PEDN_W_SEP ? v.sepw : v.sepnp;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy2D(values, (size_t)v.R * 8, src, (size_t)v.RS * 8, (size_t)v.R * 8, v.L, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

// lu: the instantiation whose slot waves perform the link update of step t-1 themselves (node_kernel<..., LU = true>)
// tf: ... and compute their own rows of turning fractions (node_kernel<.., TF> / with helper waves node_kernel_h)
static node_kernel_fn node_kernel_for(const pedn_sim* s, bool lu, bool tf) {
  const bool h = s->v.hist != 0;  // recent-history mode: the instantiations that mask the history rows
  const bool d6 = s->max_degree <= 6;  // loops and the row of turning fractions unrolled for 6 instead of 8 corridors per node
#define PEDN_NK(PR_, LP_, LU_, TF_) (h ? (d6 ? node_kernel<PR_, LP_, true, 6, LU_, TF_> : node_kernel<PR_, LP_, true, 8, LU_, TF_>) \
                                       : (d6 ? node_kernel<PR_, LP_, false, 6, LU_, TF_> : node_kernel<PR_, LP_, false, 8, LU_, TF_>))
  if (s->node_lp) return s->v.pr ? PEDN_NK(true, true, false, false) : PEDN_NK(false, true, false, false);
  if (lu && tf && s->inline_help) {   // helper waves compute the rows: sixteen waves per workgroup
    if (s->v.pr) return h ? (d6 ? node_kernel_h<true, true, 6> : node_kernel_h<true, true, 8>) : (d6 ? node_kernel_h<true, false, 6> : node_kernel_h<true, false, 8>);
    return h ? (d6 ? node_kernel_h<false, true, 6> : node_kernel_h<false, true, 8>) : (d6 ? node_kernel_h<false, false, 6> : node_kernel_h<false, false, 8>);
  }
  if (lu && tf) return s->v.pr ? PEDN_NK(true, false, true, true) : PEDN_NK(false, false, true, true);
  if (lu) return s->v.pr ? PEDN_NK(true, false, true, false) : PEDN_NK(false, false, true, false);
  return s->v.pr ? PEDN_NK(true, false, false, false) : PEDN_NK(false, false, false, false);
#undef PEDN_NK
}

// the node kernel of a clocked env step (pedn_rl_step_clocked): the step index comes from DevView.clock
static node_kernel_fn clocked_node_kernel_for(const pedn_sim* s) {
  const bool h = s->v.hist != 0, d6 = s->max_degree <= 6;
#define PEDN_NKC(PR_) (h ? (d6 ? node_kernel<PR_, false, true, 6, false, false, true> : node_kernel<PR_, false, true, 8, false, false, true>) \
                         : (d6 ? node_kernel<PR_, false, false, 6, false, false, true> : node_kernel<PR_, false, false, 8, false, false, true>))
  return s->v.pr ? PEDN_NKC(true) : PEDN_NKC(false);   // (not built for the node LP: pedn_rl_clock_begin refuses)
#undef PEDN_NKC
}

// The link update of step t as a launch of its own (the second launch of a step of a model without dynamic turning fractions, and
// the flush of a pending update under the owner-wave plan): one replica per lane.  e >= 0: start / stop events ev[e], ev[e + 1].
static unsigned link_blocks(const DevView& v, bool one_r) {
  return v.n_pairs_corr > 0 ? (unsigned)(((size_t)v.n_pairs_corr * (one_r ? v.subRS : v.subRS / 2) + 255) / 256) : 0u;
}
static void launch_link_update(pedn_sim* s, const DevView& v, hipStream_t stream, int t, hipEvent_t* ev, int e) {
  auto launch = [&](auto kernel, dim3 grid, dim3 block, auto... args) {
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev[e], ev[e + 1], 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
  };
  const unsigned nlb = link_blocks(v, true);
  if (nlb == 0) return;
  if (v.pr) { if (v.hist) launch(link_kernel_1r<true, true>, dim3(nlb), dim3(256), v, t); else launch(link_kernel_1r<true, false>, dim3(nlb), dim3(256), v, t); }
  else { if (v.hist) launch(link_kernel_1r<false, true>, dim3(nlb), dim3(256), v, t); else launch(link_kernel_1r<false, false>, dim3(nlb), dim3(256), v, t); }
}

// this launch's share of the batch: the whole of it on the engine's stream (half = -1) or one half of the replicas per stream
// (half = index of the chain, 0 or 1: each steps its share of the replicas on its own stream)
static DevView view_of(const pedn_sim* s, int half, hipStream_t* stream) {
  DevView v = s->v;
  *stream = s->stream;
  if (half >= 0) {
    // RS is a multiple of 128 (a wave of link_body covers 128 replicas): when the 128-replica segments do not halve, chain 0 takes one more
    const int a = ((s->v.RS / 128 + 1) / 2) * 128;
    v.sub0 = half ? a : 0;
    v.subRS = half ? s->v.RS - a : a;
    *stream = chain_stream(s, half);
  }
  return v;
}
static inline bool last_chain(const pedn_sim* s, int half) { return half < 0 || half == s->run_chains - 1; }

// Owner-wave plan: the link update of the last step launched is still to be done (link_pending); do it now.
static void flush_links(pedn_sim* s, int half, hipEvent_t* ev) {
  if (s->link_pending < 0) return;
  hipStream_t stream;
  const DevView v = view_of(s, half, &stream);
  launch_link_update(s, v, stream, s->link_pending, ev, 4);
  if (last_chain(s, half)) s->link_pending = -1;
}

// One step = node_kernel(t), then ONE launch with the link update of t and -- where they apply -- the turn probabilities of
// t+1 (models with softmax groups) and the RL observations / rewards of t (observe >= 0: the accumulate flag of
// rl_observe; only pedn_rl_step asks for it).  ev != nullptr: per-launch start/stop events {turn_prob, node, link} for
// pedn_profile_step.  Returns 1 through *observed when the observations were part of the launch.
// Every entry point that looks at or changes what a link update reads or writes calls this first: under the owner-wave plan the link
// update of the last step launched may still be pending (link_pending) -- the next step's node kernel would perform it.
static int join_chains(pedn_sim* s, int n);
// (also: pedn_rl_step may have left the two halves of the batch stepping as two chains on two streams across calls -- `forked` -- so
// that the engine's stream does not order stream2's work yet: joined first.  Anything that enqueues on the engine's stream or reads
// device memory calls this, i.e. every entry point but the stepping calls themselves and the pure host getters.)
static inline void join_forked(pedn_sim* s) {
  if (s->clocked) clock_end(s);   // whoever needs the host's view of the state ends a clocked section first (synchronises)
  if (s->forked) {
    s->forked = 0;
    join_chains(s, 2);
  }
}
static inline void pending_links_first(pedn_sim* s) {
  s->touched = 1;
  join_forked(s);
  if (s->link_pending >= 0) flush_links(s, -1, nullptr);
}

// The first launch of a kernel on a stream that has not run it yet costs the runtime several tens of microseconds (measured: the first
// two-chain range of a process took 32.6-33.9 us per step over 20 steps, the following ones 29.0-29.3): pay that when the plan is
// chosen.  Every step kernel is launched once on every chain's stream over NOTHING -- node_kernel on the block of idle slot records
// behind the bins (its waves meet at the two barriers and leave), the link update with zero corridors.
static void prewarm_chains(pedn_sim* s) {
  if (s->chains < 2) return;
  DevView v = s->v;
  v.slot_rec = s->d_slot_rec + (size_t)s->n_blocks * 8;   // the idle block
  DevView vl = s->v;
  vl.n_pairs_corr = 0;
  vl.n_trow = 0;
  RlView q{};
  for (int c = 0; c < s->chains; ++c) {
    hipStream_t st = chain_stream(s, c);
    for (int lu = 0; lu < 2; ++lu) hipLaunchKernelGGL(node_kernel_for(s, lu != 0, false), dim3(1, 1), dim3(512), s->node_lds, st, v, 2);
    if (s->inline_tf) hipLaunchKernelGGL(node_kernel_for(s, true, true), dim3(1, 1), dim3(s->inline_help ? 1024 : 512), s->node_lds_tf, st, v, 2);
    if (vl.hist) { hipLaunchKernelGGL((link_turn_kernel<false, false, true>), dim3(1), dim3(256), 0, st, vl, 1, 0u, 0u, 0u, q, 0);
                   hipLaunchKernelGGL((link_kernel_1r<false, true>), dim3(1), dim3(256), 0, st, vl, 1); }
    else { hipLaunchKernelGGL((link_turn_kernel<false, false, false>), dim3(1), dim3(256), 0, st, vl, 1, 0u, 0u, 0u, q, 0);
           hipLaunchKernelGGL((link_kernel_1r<false, false>), dim3(1), dim3(256), 0, st, vl, 1); }
  }
  for (int c = s->chains - 1; c >= 0; --c) hipStreamSynchronize(chain_stream(s, c));
}

// lazy (owner-wave plan, pedn_run): the link update of t is left to node_kernel<LU>(t + 1) -- or to flush_links -- and this step's
// node_kernel performs the pending one of t - 1.
static int launch_step(pedn_sim* s, int t, hipEvent_t* ev = nullptr, int observe = -1, bool* observed = nullptr,
                       const double* fold_actions = nullptr, int half = -1, bool lazy = false) {
  // half = -1: the whole batch on the engine's stream; 0 / 1: the first / second half of the replicas on stream / stream2 (the
  // caller, pedn_run, launches both halves of a step and does the per-step bookkeeping once, after the second one)
  if (half < 0 && t - 1 > s->valid_hi) {   // a step that skips ahead after a lazy reset (chains: their callers do this before the fork)
    const int rc = catch_up(s, t - 1);
    if (rc != PEDN_OK) return rc;
  }
  hipStream_t stream;
  DevView v = view_of(s, half, &stream);
  lazy = lazy && !s->node_lp && v.n_pairs_corr > 0;
  // a pending link update is flushed when this is not the step it waits for, or when this step starts with the stand-alone turning
  // fractions (they read num_pedestrians[t - 1] as stored)
  // single-launch plan (inline_tf): node_kernel<LU, TF> computes the device-computed rows itself -- wherever it can be the LU kernel
  const bool inl_plan = lazy && s->inline_tf && v.n_trow > 0;
  const bool can_inl = inl_plan && s->link_pending == t - 1 && t >= 2;
  const bool tf_alone = v.n_trow > 0 && s->tp_ready != t && !can_inl;
  const bool flush_now = s->link_pending >= 0 && (!(lazy && s->link_pending == t - 1) || tf_alone);
  const bool lu = lazy && s->link_pending == t - 1 && t >= 2 && !flush_now;   // (half 0 of a pair leaves link_pending as it is)
  const bool inl = can_inl && lu;
  if (flush_now) flush_links(s, half, nullptr);
  DevView vn = v;                 // node_kernel's view: with the action rows when it applies the gater actions itself
  vn.rl_actions = fold_actions;
  const unsigned rgroups = (unsigned)(v.subRS / 64);
  // the turning fractions of t + 1 ride in the launch behind node_kernel(t) -- except behind the last step of the horizon, where
  // pair_pod / turn_tab have no row T + 1 to read (they hold T + 1 rows, 0..T) and nothing would consume the result
  // (and not under the single-launch plan, where the next step computes its own rows)
  const bool groups = v.n_trow > 0, fused = groups && s->fuse_tp && t + 1 <= v.T1 - 1 && !inl_plan;
  const bool obs_fused = observe >= 0 && s->rl_ready && s->fuse_obs;
  auto launch = [&](auto kernel, dim3 grid, dim3 block, int e, auto... args) {
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev[e], ev[e + 1], 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
  };
  if (tf_alone) {  // first step of an episode, a repeated or an out-of-order step
    const unsigned nb = (unsigned)((v.n_trow + 3) / 4) * rgroups;  // one wave per (row of a dynamic node, 64 replicas)
    if (v.pr) { if (v.hist) launch(turn_frac_kernel<true, true>, dim3(nb), dim3(256), 0, v, t); else launch(turn_frac_kernel<true, false>, dim3(nb), dim3(256), 0, v, t); }
    else { if (v.hist) launch(turn_frac_kernel<false, true>, dim3(nb), dim3(256), 0, v, t); else launch(turn_frac_kernel<false, false>, dim3(nb), dim3(256), 0, v, t); }
    s->tp_ran = 1;
  }
  const size_t nlds = inl ? s->node_lds_tf : s->node_lds;
  const dim3 nblock(inl && s->inline_help ? 1024 : 512);   // helper waves: sixteen per workgroup
  if (ev) hipExtLaunchKernelGGL(node_kernel_for(s, lu, inl), dim3(rgroups, (unsigned)s->n_blocks), nblock, nlds, stream, ev[2], ev[3], 0, vn, t);
  else hipLaunchKernelGGL(node_kernel_for(s, lu, inl), dim3(rgroups, (unsigned)s->n_blocks), nblock, nlds, stream, vn, t);
  if (inl && last_chain(s, half)) s->tp_ready = t;   // the fractions of t are in tfd[t & 1] (a following step that cannot inline computes its own)
  // link update: one replica per lane as a launch of its own and with per-replica parameters, two inside link_turn_kernel (link_body)
  const bool one_r = v.pr || (!fused && !obs_fused);
  const unsigned nlb = lazy ? 0u : link_blocks(v, one_r);   // lazy: no link-update workgroups in this step's second launch
  if (last_chain(s, half)) s->link_pending = lazy ? t : -1;
  s->second_launch = 1;
  if (fused || obs_fused) {
    const unsigned ntb = fused ? (unsigned)((v.n_trow + 3) / 4) * rgroups : 0u;
    const unsigned nth = fused ? (unsigned)s->n_tf_heavy_quads * rgroups : 0u;  // of which in front of the link update
    const unsigned nob = obs_fused ? (unsigned)s->rl.n_agents * rgroups : 0u;  // one block per (agent, 64 replicas)
    RlView q = s->rl;
    if (!obs_fused) q.n_agents = 0;
    const int acc = observe > 0 ? 1 : 0;
    const dim3 grid(nlb + ntb + nob), block(256);
    // instantiation by (per-replica parameters, observations in the launch, recent-history mode)
#define PEDN_LT(PR_, OBS_) do { if (v.hist) launch(link_turn_kernel<PR_, OBS_, true>, grid, block, 4, v, t, nlb, ntb, nth, q, acc); \
                                else launch(link_turn_kernel<PR_, OBS_, false>, grid, block, 4, v, t, nlb, ntb, nth, q, acc); } while (0)
    if (obs_fused) {
      if (v.pr) PEDN_LT(true, true);
      else PEDN_LT(false, true);
    } else {
      if (v.pr) PEDN_LT(true, false);
      else PEDN_LT(false, false);
    }
#undef PEDN_LT
    if (fused && last_chain(s, half)) s->tp_ready = t + 1;   // the other chains of this step still have to see the old value
  } else if (nlb > 0) {
    launch_link_update(s, v, stream, t, ev, 4);
  }
  else s->second_launch = 0;
  if (observed) *observed = obs_fused;
  if (last_chain(s, half)) {
    if (s->valid_hi != 0x7fffffff && t > s->valid_hi) s->valid_hi = s->v.valid_hi = t;   // rows <= t are written once this step's launches are
    s->last_t = t;
    ++s->step_epoch;
  }
  return PEDN_OK;
}

int pedn_step(pedn_sim* s, int32_t t) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (t < 1 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  int rc = clock_end(s);
  if (rc != PEDN_OK) return rc;
  // The reference's own calling sequence -- for t in range(1, T): network_loading(t) -- on a batch that pedn_run would step as two chains:
  // from the third consecutive step that nothing looked at, the halves of the replicas step on two streams and STAY forked across the
  // calls (like pedn_rl_step's chains); whatever reads or changes the state next joins them (join_forked, as after an RL step).
  // melbourne x 1024 in a Python loop: 32.6 -> 28.9 us per step, delft 48.9 -> 43.9 (pedn_run: 28.6 / 43.4).
  const bool in_sequence = !s->touched && t == s->last_t + 1 && s->chains > 1 && s->warmed_chains >= 2 && s->v.RS >= 256;
  s->step_streak = in_sequence ? std::min(s->step_streak + 1, 2) : 0;
  const bool two = s->step_streak >= 2;
  if (!two) join_forked(s);
  // owner-wave plan: this step's link update stays pending -- the next step's node kernel performs it, or whatever call looks at
  // or changes the state first (pending_links_first)
  bool lazy = s->link_owner != 0;
  if (lazy && s->inline_tf) {   // (without device-computed rows both plans are two launches per step for such a caller)
    s->touch_streak = s->touched ? std::min(s->touch_streak + 1, 2) : std::max(s->touch_streak - 1, 0);
    if (s->touch_streak >= 2) lazy = false;
  }
  s->touched = 0;
  if (two) {
    if (!s->forked) {
      if (s->link_pending >= 0 && s->link_pending != t - 1) { pending_links_first(s); s->touched = 0; }   // a stale pending update: whole batch, before the fork
      if (t - 1 > s->valid_hi && (rc = catch_up(s, t - 1)) != PEDN_OK) return rc;
      if ((rc = fork_chains(s, 2)) != PEDN_OK) return rc;
      s->forked = 1;
    }
    s->run_chains = 2;
    for (int c = 0; c < 2 && rc == PEDN_OK; ++c) rc = launch_step(s, t, nullptr, -1, nullptr, nullptr, c, lazy);
    s->run_chains = 1;
    if (rc != PEDN_OK) return rc;
  } else if ((rc = launch_step(s, t, nullptr, -1, nullptr, nullptr, -1, lazy)) != PEDN_OK) return rc;
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_profile_step(pedn_sim* s, int32_t t, float ms[3]) {
  if (!s || !ms) return fail(s, PEDN_E_ARG, "null argument");
  if (t < 1 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);   // (a pending link update is left to the first step of the range, as in pedn_run)
  // hipExtLaunchKernelGGL start/stop events carry the dispatch's own begin/end timestamps (what rocprofv3 reports),
  // not the enqueue-to-completion interval an ordinary hipEventRecord bracket would measure.
  hipEvent_t ev[6];
  for (int i = 0; i < 6; ++i) HIP_TRY(s, hipEventCreate(&ev[i]));
  s->tp_ran = 0;
  launch_step(s, t, ev);
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  ms[0] = ms[1] = ms[2] = 0.0f;
  if (s->tp_ran) HIP_TRY(s, hipEventElapsedTime(&ms[0], ev[0], ev[1]));  // stand-alone turn probabilities (normally fused into [2])
  HIP_TRY(s, hipEventElapsedTime(&ms[1], ev[2], ev[3]));
  if (s->second_launch) HIP_TRY(s, hipEventElapsedTime(&ms[2], ev[4], ev[5]));
  for (int i = 0; i < 6; ++i) hipEventDestroy(ev[i]);
  return PEDN_OK;
}

// pedn_run's plan for the range [t0, t1): how many chains of launches (each a share of the replicas on its own stream)?
static int chains_for(const pedn_sim* s, int t0, int t1) {
  if (s->chains < 2 || t1 - t0 < 8) return 1;
  return s->v.RS >= 256 ? 2 : 1;   // (every chain at least one 128-replica segment, view_of)
}

// Fork: every other chain's stream waits for what the engine's stream holds so far.
static int fork_chains(pedn_sim* s, int n) {
  HIP_TRY(s, hipEventRecord(s->ev_fork, s->stream));
  for (int c = 1; c < n; ++c) HIP_TRY(s, hipStreamWaitEvent(chain_stream(s, c), s->ev_fork, 0));
  return PEDN_OK;
}
// Join of the chains of launches: the engine's stream waits for everything enqueued on the other streams.  If the event path fails
// the streams are drained instead, so that no call ever returns with work on them that the engine's stream does not order.
static int join_chains(pedn_sim* s, int n) {
  hipError_t e = hipSuccess;
  for (int c = 1; c < n && e == hipSuccess; ++c) {
    e = hipEventRecord(s->ev_join, chain_stream(s, c));
    if (e == hipSuccess) e = hipStreamWaitEvent(s->stream, s->ev_join, 0);
  }
  if (e != hipSuccess) {
    for (int c = n - 1; c >= 0; --c) hipStreamSynchronize(chain_stream(s, c));
    return fail(s, PEDN_E_DEVICE, std::string("joining the chains of launches: ") + hipGetErrorString(e));
  }
  return PEDN_OK;
}

int pedn_run(pedn_sim* s, int32_t t0, int32_t t1) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (t0 < 1 || t1 > s->v.T1 || t0 > t1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  // Replicas are independent, so the two halves of the batch can run as two chains of launches on two streams: while one
  // half is in its link update (few, long waves) the other half's node_kernel fills the machine, and the ramp and the tail of
  // every launch overlap the other chain's work.  Forking and joining costs two cross-stream waits per CALL, so only ranges of
  // several steps take this plan; everything else (single steps, RL steps, profiling) stays on the one stream.
  // Owner-wave plan (link_owner): one launch per step for models without dynamic turning fractions -- node_kernel<LU>(t) performs the
  // link update of t - 1 -- plus one link_kernel for the last step of the range.
  const bool lazy = s->link_owner != 0;
  int rc = clock_end(s);
  if (rc != PEDN_OK) return rc;
  join_forked(s);
  const int nch = chains_for(s, t0, t1);
  if (nch > 1) {
    if (s->link_pending >= 0 && s->link_pending != t0 - 1) pending_links_first(s);   // a stale pending update: on the whole batch, before the fork
    if (t0 - 1 > s->valid_hi && (rc = catch_up(s, t0 - 1)) != PEDN_OK) return rc;
    if ((rc = fork_chains(s, nch)) != PEDN_OK) return rc;
    s->run_chains = nch;
    for (int t = t0; t < t1; ++t)
      for (int c = 0; c < nch; ++c) launch_step(s, t, nullptr, -1, nullptr, nullptr, c, lazy);
    s->run_chains = 1;
    rc = join_chains(s, nch);    // the last step's link update stays pending on the joined stream (pending_links_first)
    if (rc != PEDN_OK) return rc;
  } else {
    for (int t = t0; t < t1; ++t)
      if ((rc = launch_step(s, t, nullptr, -1, nullptr, nullptr, -1, lazy)) != PEDN_OK) return rc;
  }
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_plan_info(pedn_sim* s, int32_t* info, int32_t n) {
  if (!s || !info || n < 4) return fail(s, PEDN_E_ARG, "pedn_plan_info: null argument or fewer than 4 entries");
  info[0] = s->chains;
  info[1] = s->link_owner && !s->node_lp && s->v.n_pairs_corr > 0;   // (with device-computed rows: the single-launch plan, inline_tf)
  info[2] = s->stream_probe_attempts;
  info[3] = (int32_t)(s->stream_probe_ms * 1000.0f + 0.5f);
  if (n >= 5) info[4] = s->packed_by;   // bins of node_kernel packed by 0 degree, 1 the static load estimate, 2 measured node cost
  return PEDN_OK;
}

int pedn_set_streams(pedn_sim* s, int32_t n) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (n != 1 && n != 2) return fail(s, PEDN_E_ARG, "1 or 2 chains of launches");
  HIP_TRY(s, hipSetDevice(s->device));
  int rc = clock_end(s);
  if (rc != PEDN_OK) return rc;
  if (n > s->warmed_chains) {
    pending_links_first(s);
    int got = 1;
    if ((rc = warm_chain_streams(s, &got)) != PEDN_OK) return rc;
    s->warmed_chains = got;
  }
  const int before = s->chains;
  s->chains = n > 1 ? std::min<int>(n, std::max(s->warmed_chains, 1)) : 1;
  s->two_streams = s->chains > 1;
  if (s->chains > before) prewarm_chains(s);   // (every chain's stream: hipSetDevice above)
  return PEDN_OK;
}

// Steps [t0, t1) under pedn_run's plan with every launch bracketed by its dispatch timestamps.  rows: one per launch,
// {step, chain, kind (0 stand-alone turning fractions, 1 node_kernel, 2 the launch behind it), start, end} in ms after the start of
// the first launch of the range.
struct ProfRow { int t, chain, kind; float start, end; };
static int profile_range(pedn_sim* s, int t0, int t1, std::vector<ProfRow>& rows, int* chains) {
  const int halves = chains_for(s, t0, t1), n = (t1 - t0) * halves;
  const bool two = halves > 1;
  const bool lazy = s->link_owner != 0;   // pedn_run's plan
  struct Events {  // destroyed on every way out of the function
    std::vector<hipEvent_t> e;
    ~Events() { for (hipEvent_t x : e) if (x) hipEventDestroy(x); }
  } evs;
  evs.e.assign((size_t)(n + halves) * 6, nullptr);   // the last `halves` sets: the trailing link update of the owner-wave plan
  std::vector<hipEvent_t>& ev = evs.e;
  for (auto& e : ev) HIP_TRY(s, hipEventCreate(&e));
  std::vector<int> tp_ran((size_t)n, 0), second((size_t)n, 0);
  if (two) {
    if (s->link_pending >= 0 && s->link_pending != t0 - 1) pending_links_first(s);
    int rc = PEDN_OK;
    if (t0 - 1 > s->valid_hi && (rc = catch_up(s, t0 - 1)) != PEDN_OK) return rc;
    if ((rc = fork_chains(s, halves)) != PEDN_OK) return rc;
    s->run_chains = halves;
  }
  for (int t = t0, k = 0; t < t1; ++t)
    for (int h = 0; h < halves; ++h, ++k) {
      s->tp_ran = 0;
      launch_step(s, t, &ev[(size_t)k * 6], -1, nullptr, nullptr, two ? h : -1, lazy);
      tp_ran[k] = s->tp_ran;
      second[k] = s->second_launch;
    }
  const bool flushed = s->link_pending >= 0;
  for (int h = 0; h < halves; ++h) flush_links(s, two ? h : -1, &ev[(size_t)(n + h) * 6]);
  s->run_chains = 1;
  if (two) {
    const int rc = join_chains(s, halves);
    if (rc != PEDN_OK) return rc;
  }
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  // the first launch of the range (chain 0): its node_kernel, or the stand-alone turning fractions in front of it
  hipEvent_t origin = tp_ran[0] ? ev[0] : ev[2];
  rows.clear();
  for (int k = 0; k < n; ++k) {
    const int t = t0 + k / halves, chain = k % halves;
    for (int kind = 0; kind < 3; ++kind) {
      if ((kind == 0 && !tp_ran[k]) || (kind == 2 && !second[k])) continue;
      ProfRow r{t, chain, kind, 0.0f, 0.0f};
      HIP_TRY(s, hipEventElapsedTime(&r.start, origin, ev[(size_t)k * 6 + 2 * kind]));
      HIP_TRY(s, hipEventElapsedTime(&r.end, origin, ev[(size_t)k * 6 + 2 * kind + 1]));
      rows.push_back(r);
    }
  }
  for (int h = 0; h < halves && flushed; ++h) {   // the trailing link update: the second launch of the range's last step
    ProfRow r{t1 - 1, h, 2, 0.0f, 0.0f};
    HIP_TRY(s, hipEventElapsedTime(&r.start, origin, ev[(size_t)(n + h) * 6 + 4]));
    HIP_TRY(s, hipEventElapsedTime(&r.end, origin, ev[(size_t)(n + h) * 6 + 5]));
    rows.push_back(r);
  }
  *chains = halves;
  return PEDN_OK;
}

int pedn_profile_run(pedn_sim* s, int32_t t0, int32_t t1, float ms[3], int32_t* chains) {
  if (!s || !ms || !chains) return fail(s, PEDN_E_ARG, "null argument");
  if (t0 < 1 || t1 > s->v.T1 || t0 >= t1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);   // (a pending link update is left to the first step of the range, as in pedn_run)
  std::vector<ProfRow> rows;
  int ch = 1;
  const int rc = profile_range(s, t0, t1, rows, &ch);
  if (rc != PEDN_OK) return rc;
  double sum[3] = {0, 0, 0};
  int cnt[3] = {0, 0, 0};
  for (const ProfRow& r : rows) { sum[r.kind] += (double)r.end - (double)r.start; ++cnt[r.kind]; }
  for (int i = 0; i < 3; ++i) ms[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0.0f;
  *chains = ch;
  return PEDN_OK;
}

int pedn_profile_timeline(pedn_sim* s, int32_t t0, int32_t t1, float* out, int32_t capacity, int32_t* n_rows, int32_t* chains) {
  if (!s || !out || !n_rows || !chains) return fail(s, PEDN_E_ARG, "null argument");
  if (t0 < 1 || t1 > s->v.T1 || t0 >= t1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);   // (a pending link update is left to the first step of the range, as in pedn_run)
  std::vector<ProfRow> rows;
  int ch = 1;
  const int rc = profile_range(s, t0, t1, rows, &ch);
  if (rc != PEDN_OK) return rc;
  if ((int)rows.size() > capacity) return fail(s, PEDN_E_ARG, "timeline buffer too small: " + std::to_string(rows.size()) + " rows");
  for (size_t i = 0; i < rows.size(); ++i) {
    out[5 * i] = (float)rows[i].t; out[5 * i + 1] = (float)rows[i].chain; out[5 * i + 2] = (float)rows[i].kind;
    out[5 * i + 3] = rows[i].start; out[5 * i + 4] = rows[i].end;
  }
  *n_rows = (int32_t)rows.size();
  *chains = ch;
  return PEDN_OK;
}

int pedn_synchronize(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PEDN_OK;
}

int pedn_error_flags(pedn_sim* s, uint32_t* flags) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  std::vector<uint32_t> h(s->v.RS);
  HIP_TRY(s, hipMemcpy(h.data(), s->v.flags, (size_t)s->v.RS * 4, hipMemcpyDeviceToHost));
  uint32_t any = 0;
  for (int r = 0; r < s->v.R; ++r) {
    any |= h[r];
    if (flags) flags[r] = h[r];
  }
  return (int)any;
}

int pedn_read(pedn_sim* s, int32_t field, int32_t t0, int32_t t1, int32_t c0, int32_t c1, int32_t r0, int32_t r1, void* out) {
  if (!s || !out) return fail(s, PEDN_E_ARG, "null argument");
  DevView& v = s->v;
  if (field < 0 || field >= PEDN_N_FIELDS) return fail(s, PEDN_E_ARG, "unknown field");
  int cols = field < 4 ? v.Lall : v.L;
  if (t0 < 0 || t1 > v.T1 || t0 >= t1 || c0 < 0 || c1 > cols || c0 >= c1 || r0 < 0 || r1 > v.R || r0 >= r1)
    return fail(s, PEDN_E_ARG, "read range out of bounds");
  const int rows = field < 7 ? s->rows64[field] : s->rows32[field - 7], mask = field < 7 ? v.m64[field] : v.m32[field - 7];
  if (rows < v.T1) {  // recent-history mode: only the newest `rows` time indices of this field exist
    // sending_flow / receiving_flow of step t are entries t - 1 (node.py:206, link.py:367): their newest entry is one behind
    const int newest = std::max((field == F_S || field == F_R) ? s->last_t - 1 : s->last_t, 0);
    if (t1 - 1 > newest || t0 <= newest - rows)
      return fail(s, PEDN_E_ARG, "time index outside the field's ring (recent-history mode keeps the last " + std::to_string(rows) +
                                 " entries of this field; the newest is " + std::to_string(newest) + ")");
  }
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  // (no wait here: everything the gather reads was written by earlier work on this stream)
  size_t n = (size_t)(t1 - t0) * (c1 - c0) * (r1 - r0);
  size_t esz = field < 7 ? 8 : 4;
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, n * esz, &st);
  if (rc != PEDN_OK) return rc;
  unsigned blocks = (unsigned)((n + 255) / 256);
  // lazy reset: rows that were neither written nor cleared since the reset read as their initial value.  sending / receiving flow of
  // step t are entries t - 1; the gate record and avg_travel_time below the window were restored by the reset itself.
  int hi = 0x7fffffff;
  if (s->valid_hi != 0x7fffffff) {
    hi = s->valid_hi;
    if (field == F_S || field == F_R) hi = std::max(hi - 1, 0);
    else if (field == F_GATE) hi = 0x7fffffff;
    else if (field == 7 + G_ATT) hi = std::max(hi, v.W - 1);
  }
  // small reads (a controller looking at a few values after every step): the gather writes STRAIGHT into the pinned buffer over the bus --
  // no copy command behind the kernel (its ~10 us of stream latency were half of such a read)
  const bool direct = n * esz <= PEDN_DIRECT_READ_BYTES;
  void* const dst = direct ? st->pin : st->dev;
  if (field < 7)
    hipLaunchKernelGGL(gather_kernel<double>, dim3(blocks), dim3(256), 0, s->stream, (const double*)v.f64[field], (double*)dst, t0,
                       t1 - t0, c0, c1 - c0, r0, r1 - r0, cols, v.RS, mask, hi, (field == F_S || field == F_R) ? -1.0 : 0.0);
  else
    hipLaunchKernelGGL(gather_kernel<float>, dim3(blocks), dim3(256), 0, s->stream, (const float*)v.f32[field - 7], (float*)dst, t0,
                       t1 - t0, c0, c1 - c0, r0, r1 - r0, cols, v.RS, mask, hi, 0.0f);
  HIP_TRY(s, hipGetLastError());
  if (!direct) HIP_TRY(s, hipMemcpyAsync(st->pin, st->dev, n * esz, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  memcpy(out, st->pin, n * esz);
  return stage_commit(s, st);
}

int pedn_flush(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);   // joins forked chains, ends a clocked section, performs a pending link update -- all on pedn_stream()
  if (s->valid_hi != 0x7fffffff) {   // a zero-copy consumer may look at any row: finish what a lazy reset left out
    const int rc = catch_up(s, s->v.T1 - 1);
    if (rc != PEDN_OK) return rc;
    s->valid_hi = s->v.valid_hi = 0x7fffffff;
  }
  return PEDN_OK;
}

void* pedn_device_ptr(pedn_sim* s, int32_t field, int64_t* columns, int64_t* replica_stride) {
  if (!s || field < 0 || field >= PEDN_N_FIELDS) return nullptr;
  if ((s->link_pending >= 0 || s->forked || s->clocked || s->valid_hi != 0x7fffffff) && pedn_flush(s) != PEDN_OK) return nullptr;
  if (columns) *columns = field < 4 ? s->v.Lall : s->v.L;
  if (replica_stride) *replica_stride = s->v.RS;
  return field < 7 ? (void*)s->v.f64[field] : (void*)s->v.f32[field - 7];
}

int pedn_history_rows(pedn_sim* s, int32_t field) {
  if (!s || field < 0 || field >= PEDN_N_FIELDS) return fail(s, PEDN_E_ARG, "unknown field");
  return field < 7 ? s->rows64[field] : s->rows32[field - 7];
}

void* pedn_stream(pedn_sim* s) { return s ? (void*)s->stream : nullptr; }

int pedn_timer_begin(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipEventRecord(s->ev0, s->stream));
  return PEDN_OK;
}

int pedn_timer_end(pedn_sim* s, float* ms) {
  if (!s || !ms) return fail(s, PEDN_E_ARG, "null argument");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipEventRecord(s->ev1, s->stream));
  HIP_TRY(s, hipEventSynchronize(s->ev1));
  HIP_TRY(s, hipEventElapsedTime(ms, s->ev0, s->ev1));
  return PEDN_OK;
}

int pedn_set_link_params(pedn_sim* s, const double* kc, const double* kj, const double* vf, const int32_t* fft, const int32_t* tau_sw,
                         const float* tt0) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  s->tp_ready = -1;
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  DevView& v = s->v;
  if (!kc) {  // back to the shared parameters
    v.pr = 0;
    return PEDN_OK;
  }
  if (!kj || !vf || !fft || !tau_sw || !tt0) return fail(s, PEDN_E_ARG, "all six parameter matrices are required");
  for (size_t i = 0; i < (size_t)v.L * v.R; ++i) {
    if (!(kj[i] > kc[i]) || !(kc[i] > 0.0) || !(vf[i] > 0.0)) return fail(s, PEDN_E_ARG, "need 0 < k_critical < k_jam and free_flow_speed > 0");
    if (fft[i] < 0 || tau_sw[i] < 0) return fail(s, PEDN_E_ARG, "negative look-back");
    if (v.hist && tau_sw[i] + 2 > s->rows64[F_CO]) return fail(s, PEDN_E_ARG, "shock-wave look-back longer than the cumulative_outflow ring (recent-history mode)");
  }
  for (size_t i = 0; i < (size_t)v.L * v.R; ++i)
    if (fft[i] > 32767 || tau_sw[i] > 32767) return fail(s, PEDN_E_ARG, "look-back beyond 32767 steps");
  int rc;
  if ((rc = ensure_link_records(s)) != PEDN_OK) return rc;
  // the six matrices through one staging slot, packed into the 32-byte records on the device
  const size_t n = (size_t)v.L * v.R, bytes = n * (3 * 8 + 2 * 4 + 4);
  pedn_sim::Stage* st;
  if ((rc = stage_acquire(s, bytes, &st)) != PEDN_OK) return rc;
  unsigned char* h = (unsigned char*)st->pin;
  memcpy(h, kc, n * 8); memcpy(h + n * 8, kj, n * 8); memcpy(h + 2 * n * 8, vf, n * 8);
  memcpy(h + 3 * n * 8, fft, n * 4); memcpy(h + 3 * n * 8 + n * 4, tau_sw, n * 4); memcpy(h + 3 * n * 8 + 2 * n * 4, tt0, n * 4);
  HIP_TRY(s, hipMemcpyAsync(st->dev, st->pin, bytes, hipMemcpyHostToDevice, s->stream));
  const unsigned char* d = (const unsigned char*)st->dev;
  const size_t lanes = (size_t)v.L * v.RS;
  hipLaunchKernelGGL(pack_link_params_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s->stream, s->d_prm, (const double*)d,
                     (const double*)(d + n * 8), (const double*)(d + 2 * n * 8), (const int32_t*)(d + 3 * n * 8),
                     (const int32_t*)(d + 3 * n * 8 + n * 4), (const float*)(d + 3 * n * 8 + 2 * n * 4), v.L, v.R, v.RS);
  HIP_TRY(s, hipGetLastError());
  if ((rc = stage_commit(s, st)) != PEDN_OK) return rc;
  v.prm = s->d_prm;
  v.pr = 1;
  return PEDN_OK;
}

int pedn_get_link_params(pedn_sim* s, double* kc, double* kj, double* vf, int32_t* fft, int32_t* tau_sw, float* tt0) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->v.pr || !s->d_prm) return fail(s, PEDN_E_ARG, "no per-replica link parameters are set");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  const DevView& v = s->v;
  const size_t n = (size_t)v.L * v.R, bytes = n * (3 * 8 + 2 * 4 + 4);
  pedn_sim::Stage* st;
  int rc = stage_acquire(s, bytes, &st);
  if (rc != PEDN_OK) return rc;
  unsigned char* d = (unsigned char*)st->dev;
  hipLaunchKernelGGL(unpack_link_params_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (const LinkPR*)s->d_prm,
                     (double*)d, (double*)(d + n * 8), (double*)(d + 2 * n * 8), (int32_t*)(d + 3 * n * 8), (int32_t*)(d + 3 * n * 8 + n * 4),
                     (float*)(d + 3 * n * 8 + 2 * n * 4), v.L, v.R, v.RS);
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipMemcpyAsync(st->pin, st->dev, bytes, hipMemcpyDeviceToHost, s->stream));
  if ((rc = stage_commit(s, st)) != PEDN_OK) return rc;
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  const unsigned char* h = (const unsigned char*)st->pin;
  if (kc) memcpy(kc, h, n * 8);
  if (kj) memcpy(kj, h + n * 8, n * 8);
  if (vf) memcpy(vf, h + 2 * n * 8, n * 8);
  if (fft) memcpy(fft, h + 3 * n * 8, n * 4);
  if (tau_sw) memcpy(tau_sw, h + 3 * n * 8 + n * 4, n * 4);
  if (tt0) memcpy(tt0, h + 3 * n * 8 + 2 * n * 4, n * 4);
  return PEDN_OK;
}

int pedn_set_od_weights_per_replica(pedn_sim* s, const double* w) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  DevView& v = s->v;
  s->tp_ready = -1;
  if (!w) {
    v.pod_pr = 0;
    return PEDN_OK;
  }
  if (s->n_pair == 0) return PEDN_OK;
  int rc;
  if ((rc = ensure_pod_tables(s)) != PEDN_OK) return rc;
  if ((rc = push_matrix(s, s->d_od_w_r, w, s->n_od)) != PEDN_OK) return rc;
  return scenario_pod_tables(s);
}

int pedn_get_od_weights_per_replica(pedn_sim* s, double* w) {
  if (!s || !w) return fail(s, PEDN_E_ARG, "null argument");
  if (!s->v.pod_pr || !s->d_od_w_r) return fail(s, PEDN_E_ARG, "no per-replica OD weights are set");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy2D(w, (size_t)s->v.R * 8, s->d_od_w_r, (size_t)s->v.RS * 8, (size_t)s->v.R * 8, s->n_od, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

int pedn_randomize_scenarios(pedn_sim* s, uint64_t seed, double link_fraction, int32_t what, const int32_t* origin_nodes, int32_t n_origins) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!(link_fraction >= 0.0 && link_fraction <= 1.0)) return fail(s, PEDN_E_ARG, "link_fraction outside [0, 1]");
  if ((what & 4) && n_origins > 0 && !origin_nodes) return fail(s, PEDN_E_ARG, "origin nodes missing");
  HIP_TRY(s, hipSetDevice(s->device));
  DevView& v = s->v;
  pending_links_first(s);
  s->tp_ready = -1;
  const uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
  int rc;
  if ((what & 1) && v.n_pairs_corr > 0) {
    const int k = (int)((double)v.n_pairs_corr * link_fraction);   // int(len(valid_links) * 0.2), env_loader.py:393
    if ((rc = ensure_link_records(s)) != PEDN_OK) return rc;
    if (!s->d_max_tau && (rc = dalloc(s, 1, &s->d_max_tau)) != PEDN_OK) return rc;
    HIP_TRY(s, hipMemsetAsync(s->d_max_tau, 0, sizeof(int), s->stream));
    // recent-history mode: the cumulative_outflow ring must cover the longest shock-wave look-back drawn -- checked BEFORE the draw
    // replaces the scenario in use: the records are drawn into a second buffer and copied over only when the check has passed (a copy,
    // not a swap of the two pointers: DevView.prm stays what captured launches carry, pedn_rl_clock_signature)
    LinkPR* dst = s->d_prm;
    if (v.hist) {
      if (!s->d_prm_draw && (rc = dalloc(s, (size_t)v.L * v.RS, &s->d_prm_draw)) != PEDN_OK) return rc;
      dst = s->d_prm_draw;
    }
    hipLaunchKernelGGL(rand_links_kernel, dim3((unsigned)((v.RS + 63) / 64)), dim3(64), 0, s->stream, v, dst, k, k0, k1, s->d_max_tau);
    HIP_TRY(s, hipGetLastError());
    if (v.hist) {
      int mt = 0;
      HIP_TRY(s, hipMemcpyAsync(&mt, s->d_max_tau, sizeof(int), hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(s, hipStreamSynchronize(s->stream));
      if (mt + 2 > s->rows64[F_CO])   // the engine keeps the scenario it had
        return fail(s, PEDN_E_ARG, "a drawn shock-wave look-back (" + std::to_string(mt) + ") is longer than the cumulative_outflow ring (recent-history mode)");
      HIP_TRY(s, hipMemcpyAsync(s->d_prm, s->d_prm_draw, (size_t)v.L * v.RS * sizeof(LinkPR), hipMemcpyDeviceToDevice, s->stream));
    }
    v.prm = s->d_prm;
    v.pr = 1;
  }
  if ((what & 2) && s->n_pair > 0) {
    if ((rc = ensure_pod_tables(s)) != PEDN_OK) return rc;
    const size_t lanes = (size_t)s->n_od * v.RS;
    hipLaunchKernelGGL(rand_od_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s->stream, s->d_od_w_r, s->n_od, v.RS, k0, k1, v.replica_offset);
    if ((rc = scenario_pod_tables(s)) != PEDN_OK) return rc;
  }
  if ((what & 4) && n_origins > 0) {
    std::vector<int32_t> hn((size_t)2 * n_origins);
    for (int i = 0; i < n_origins; ++i) {
      const int node = origin_nodes[i];
      if (node < 0 || node >= s->n_nodes || s->node_demand_row[node] < 0) return fail(s, PEDN_E_ARG, "origin node without a virtual link");
      hn[i] = s->node_demand_row[node];
      hn[n_origins + i] = node;
    }
    pedn_sim::Stage* st;
    if ((rc = stage_acquire(s, hn.size() * 4, &st)) != PEDN_OK) return rc;
    if ((rc = stage_upload(s, st, hn.data(), hn.size() * 4)) != PEDN_OK) return rc;
    const size_t lanes = (size_t)v.T1 * v.RS;
    hipLaunchKernelGGL(rand_demand_kernel, dim3((unsigned)((lanes + 255) / 256), (unsigned)n_origins), dim3(256), 0, s->stream, v.demand,
                       (const int32_t*)st->dev, (const int32_t*)st->dev + n_origins, v.T1, v.R, v.RS, k0, k1, v.replica_offset);
    HIP_TRY(s, hipGetLastError());
    if ((rc = stage_commit(s, st)) != PEDN_OK) return rc;
  }
  return PEDN_OK;
}

int pedn_rl_configure(pedn_sim* s, const pedn_rl_desc* d, int32_t* n_actions, int32_t* n_obs) {
  if (!s || !d) return fail(s, PEDN_E_ARG, "null argument");
  if (d->n_agents < 1) return fail(s, PEDN_E_ARG, "no agents");
  if (d->obs_mode < 1 || d->obs_mode > 5) return fail(s, PEDN_E_ARG, "obs_mode must be 1..5");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  static const int fpl_of[6] = {0, 3, 4, 5, 2, 7};  // builders.py:47-58
  const int fpl = fpl_of[d->obs_mode];
  DevView& v = s->v;
  std::vector<int32_t> act_off(d->n_agents), obs_off(d->n_agents), slot_agent, slot_idx;
  int A = 0, O = 0;
  for (int a = 0; a < d->n_agents; ++a) {
    const int n = d->agent_link_ptr[a + 1] - d->agent_link_ptr[a];
    act_off[a] = A; obs_off[a] = O;
    for (int k = d->agent_link_ptr[a]; k < d->agent_link_ptr[a + 1]; ++k) {
      int l = d->agent_links[k];
      if (l < 0 || l >= v.L) return fail(s, PEDN_E_ARG, "agent link out of range");
    }
    if (d->agent_type[a] == 0) {
      if (n != 2) return fail(s, PEDN_E_ARG, "separator agent needs (forward, reverse)");
      if (d->normalize && (d->obs_mode == 3 || d->obs_mode == 4))
        return fail(s, PEDN_E_ARG, "normalised separator observations index out of range for option3/option4 (IndexError in the reference, builders.py:189-198)");
      slot_agent.push_back(a); slot_idx.push_back(0);
      A += 1; O += 4;
    } else if (d->agent_type[a] == 1) {
      if (n < 1 || n > PEDN_MAX_DEGREE) return fail(s, PEDN_E_ARG, "gater outdegree outside 1..8");
      if (d->normalize && d->obs_mode == 4)
        return fail(s, PEDN_E_ARG, "normalised gater observations index out of range for option4 (IndexError in the reference, builders.py:229-236)");
      for (int i = 0; i < n; ++i) { slot_agent.push_back(a); slot_idx.push_back(i); }
      A += n; O += n * fpl;
    } else return fail(s, PEDN_E_ARG, "agent_type must be 0 or 1");
  }
  RlView& q = s->rl;
  int rc;
#define UP(src, n, dst) if ((rc = upload(s, src, n, dst)) != PEDN_OK) return rc
  UP(d->agent_type, (size_t)d->n_agents, &q.agent_type);
  UP(d->agent_link_ptr, (size_t)d->n_agents + 1, &q.agent_link_ptr);
  UP(d->agent_links, (size_t)d->agent_link_ptr[d->n_agents], &q.agent_links);
  UP(act_off.data(), act_off.size(), &q.agent_act_off);
  UP(obs_off.data(), obs_off.size(), &q.agent_obs_off);
  UP(slot_agent.data(), slot_agent.size(), &q.slot_agent);
  UP(slot_idx.data(), slot_idx.size(), &q.slot_idx);
#undef UP
  if ((rc = dalloc(s, (size_t)v.R * A, &q.actions)) != PEDN_OK) return rc;
  // observations and rewards in ONE allocation, rewards right behind the observations: a fetch of both is one copy (rl_fetch)
  if ((rc = dalloc(s, (size_t)v.R * O + (size_t)v.R * d->n_agents, &q.obs)) != PEDN_OK) return rc;
  q.rew = q.obs + (size_t)v.R * O;
  HIP_TRY(s, hipMemset(q.obs, 0, (size_t)v.R * O * sizeof(float)));
  HIP_TRY(s, hipMemset(q.rew, 0, (size_t)v.R * d->n_agents * sizeof(float)));
  q.n_agents = d->n_agents; q.A = A; q.O = O; q.obs_mode = d->obs_mode; q.normalize = d->normalize; q.reward_mode = d->reward_mode;
  q.fpl = fpl; q.max_delta_sep = d->max_delta_sep; q.max_delta_gate = d->max_delta_gate; q.min_sep = d->min_sep;
  // widths of controlled links are written per replica by rl_apply_kernel: never take the uniform shortcut for them
  s->h_rl_link.assign((size_t)v.L, 0);
  for (int a = 0; a < d->n_agents; ++a)
    for (int k = d->agent_link_ptr[a]; k < d->agent_link_ptr[a + 1]; ++k) s->h_rl_link[d->agent_links[k]] = 1;
  {
    std::vector<LinkP> lp((size_t)v.L);
    HIP_TRY(s, hipMemcpy(lp.data(), v.lp, lp.size() * sizeof(LinkP), hipMemcpyDeviceToHost));
    for (int a = 0; a < d->n_agents; ++a)
      for (int k = d->agent_link_ptr[a]; k < d->agent_link_ptr[a + 1]; ++k) {
        s->h_rl_link[lp[d->agent_links[k]].rev] = 1;
      }
  }
  if ((rc = push_uniform(s)) != PEDN_OK) return rc;
  {  // a gater action (back gate of outgoing link l of the gater node = front gate of its reverse) is consumed by exactly one
     // wave of node_kernel, the slot (lin = reverse(l), lout = l): that wave can apply it.  A separator's width is read at both
     // ends of its corridor, so agent sets with a separator keep the separate rl_apply_kernel launch.
    bool only_gaters = true;
    for (SlotRec& R : s->h_slot_rec) R.act = -1;
    for (int a = 0, slot = 0; a < d->n_agents; ++a) {
      const int n = d->agent_link_ptr[a + 1] - d->agent_link_ptr[a];
      if (d->agent_type[a] == 0) { only_gaters = false; slot += 1; continue; }
      for (int i = 0; i < n; ++i, ++slot) {
        const int l = d->agent_links[d->agent_link_ptr[a] + i];
        int hits = 0;
        for (SlotRec& R : s->h_slot_rec)
          if (R.node >= 0 && R.lout == l && R.lin < v.L) { R.act = slot; ++hits; }
        if (hits != 1) only_gaters = false;
      }
    }
    HIP_TRY(s, hipStreamSynchronize(s->stream));
    HIP_TRY(s, hipMemcpy(s->d_slot_rec, s->h_slot_rec.data(), s->h_slot_rec.size() * sizeof(SlotRec), hipMemcpyHostToDevice));
    s->rl_fold = only_gaters;
    if (const char* f = getenv("PEDN_RL_FOLD")) s->rl_fold = s->rl_fold && atoi(f) != 0;
    v.rl_A = A;
    v.rl_max_delta_gate = d->max_delta_gate;
    v.rl_actions = nullptr;
  }
  s->rl_ready = true;
  if (n_actions) *n_actions = A;
  if (n_obs) *n_obs = O;
  return PEDN_OK;
}

int pedn_rl_apply_actions(pedn_sim* s, const double* actions, int32_t on_device) {
  if (!s || !actions) return fail(s, PEDN_E_ARG, "null argument");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  DevView& v = s->v;
  RlView& q = s->rl;
  const size_t bytes = (size_t)v.R * q.A * sizeof(double);
  RlView qq = q;
  if (on_device) qq.actions = const_cast<double*>(actions);  // read the caller's rows in place: no staging copy
  pedn_sim::Stage* st = nullptr;
  if (!on_device) {   // the host buffer is borrowed for the call only
    int rc;
    if (bytes <= PEDN_IN_PLACE_BYTES) {
      if ((rc = stage_in_place(s, actions, bytes, &st)) != PEDN_OK) return rc;
      qq.actions = (double*)st->pin;
    } else if ((rc = upload_through_stage(s, q.actions, actions, bytes)) != PEDN_OK) return rc;
  }
  size_t n = (size_t)q.A * v.RS;
  hipLaunchKernelGGL(rl_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, qq);
  HIP_TRY(s, hipGetLastError());
  if (st) return stage_commit(s, st);
  return PEDN_OK;
}

// observations / rewards -> the caller's buffers: both copies land in ONE pinned buffer, one wait for the stream, then two memcpys (into
// pageable memory each copy is staged and waited for by the runtime on its own)
static int rl_fetch(pedn_sim* s, float* obs, float* rewards) {
  if (!obs && !rewards) return PEDN_OK;
  const RlView& q = s->rl;
  const size_t nb_obs = (size_t)s->v.R * q.O * sizeof(float), nb_rew = (size_t)s->v.R * q.n_agents * sizeof(float);
  if (s->rl_pin_bytes < nb_obs + nb_rew) {
    if (s->rl_pin) HIP_TRY(s, hipHostFree(s->rl_pin));
    s->rl_pin = nullptr;
    s->rl_pin_bytes = 0;
    HIP_TRY(s, hipHostMalloc(&s->rl_pin, nb_obs + nb_rew, hipHostMallocDefault));
    s->rl_pin_bytes = nb_obs + nb_rew;
  }
  char* pin = (char*)s->rl_pin;
  if (obs && rewards) HIP_TRY(s, hipMemcpyAsync(pin, q.obs, nb_obs + nb_rew, hipMemcpyDeviceToHost, s->stream));   // (contiguous: pedn_rl_configure)
  else if (obs) HIP_TRY(s, hipMemcpyAsync(pin, q.obs, nb_obs, hipMemcpyDeviceToHost, s->stream));
  else HIP_TRY(s, hipMemcpyAsync(pin + nb_obs, q.rew, nb_rew, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  if (obs) memcpy(obs, pin, nb_obs);
  if (rewards) memcpy(rewards, pin + nb_obs, nb_rew);
  return PEDN_OK;
}

int pedn_rl_observe(pedn_sim* s, int32_t t, int32_t accumulate, float* obs, float* rewards) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  if (t < 0 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 0..T");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);
  if (t > s->valid_hi) catch_up(s, t);   // observations of a step that has not run read its rows as they were initialised
  DevView& v = s->v;
  RlView& q = s->rl;
  if (v.hist) hipLaunchKernelGGL(rl_observe_kernel<true>, dim3((unsigned)q.n_agents * (unsigned)(v.RS / 64)), dim3(256), 0, s->stream, v, q, t, accumulate);
  else hipLaunchKernelGGL(rl_observe_kernel<false>, dim3((unsigned)q.n_agents * (unsigned)(v.RS / 64)), dim3(256), 0, s->stream, v, q, t, accumulate);
  HIP_TRY(s, hipGetLastError());
  return rl_fetch(s, obs, rewards);
}

int pedn_rl_fetch(pedn_sim* s, float* obs, float* rewards) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  HIP_TRY(s, hipSetDevice(s->device));
  join_forked(s);   // (ends a clocked section; both chains' work in front of the copies)
  return rl_fetch(s, obs, rewards);
}

int pedn_rl_step(pedn_sim* s, const double* actions, int32_t on_device, int32_t t, int32_t action_gap, float* obs, float* rewards) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (action_gap < 1 || t < 1 || t + action_gap - 1 > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  int rc = PEDN_OK;
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  if (s->clocked) {   // back to host-side step indices (the fold decision below reads tp_ready)
    HIP_TRY(s, hipSetDevice(s->device));
    if ((rc = clock_end(s)) != PEDN_OK) return rc;
  }
  RlView& q = s->rl;
  // gater-only agent sets: node_kernel of the first sub-step applies the actions (one launch less).  Not when the turning
  // fractions of step t still have to be computed by their own launch in front of node_kernel: their capacity fallback reads
  // the gate widths the actions are about to change (path_finder.py:575-576).
  const double* fold = nullptr;
  pedn_sim::Stage* fold_stage = nullptr;   // host actions read in place by node_kernel of the first sub-step (stage_in_place)
  if (actions) {
    const bool stand_alone_tf = s->v.n_trow > 0 && s->tp_ready != t;
    if (s->rl_fold && !stand_alone_tf) {
      HIP_TRY(s, hipSetDevice(s->device));
      if (on_device) fold = actions;
      else {
        const size_t bytes = (size_t)s->v.R * q.A * sizeof(double);
        if (bytes <= PEDN_IN_PLACE_BYTES) {
          if ((rc = stage_in_place(s, actions, bytes, &fold_stage)) != PEDN_OK) return rc;
          fold = (const double*)fold_stage->pin;
        } else {
          if ((rc = upload_through_stage(s, q.actions, actions, bytes)) != PEDN_OK) return rc;
          fold = q.actions;
        }
      }
    } else if ((rc = pedn_rl_apply_actions(s, actions, on_device)) != PEDN_OK) return rc;   // (performs a pending link update first)
  }
  // Two chains that stay forked ACROSS calls (rl_chains): the two halves of the envs step on two streams, call after call, and are
  // joined only when something needs the whole batch (join_forked: an observation fetch, a setter, a read, a synchronise).  A step is two
  // dependent launches whose durations barely depend on the batch size, so the two half-batch chains run side by side almost for free
  // (45_intersections x 2048: 25.0 -> 22.x us per env step).  Only when nothing of this call runs on the engine's stream alone: the
  // actions are applied inside node_kernel (or there are none) and the observations ride in the second launch.
  // (actions from the host go through the engine's own action buffer, which the other chain's previous step may still be reading)
  if (t - 1 > s->valid_hi) {   // skipping ahead after a lazy reset
    join_forked(s);
    if ((rc = catch_up(s, t - 1)) != PEDN_OK) return rc;
  }
  const bool by_batch = s->v.RS >= 4096 || (s->v.pr && s->v.RS >= 1024);
  // on_device == 2: the caller chains this call between its own streams and pedn_stream() with events (no host synchronisation):
  // everything must then be ordered by the engine's stream alone
  const bool two = on_device != 2 && (s->rl_chains == 2 || (s->rl_chains == 0 && by_batch)) && s->warmed_chains >= 2 && s->v.RS >= 256 && s->fuse_obs && (!actions || (fold != nullptr && on_device)) && !obs && !rewards &&
                   s->link_pending < 0 && !(s->v.n_trow > 0 && !s->fuse_tp);
  if (!two) join_forked(s);
  else if (!s->forked) {
    if ((rc = fork_chains(s, 2)) != PEDN_OK) return rc;
    s->forked = 1;
  }
  for (int k = 0; k < action_gap; ++k) {
    const bool last = k == action_gap - 1;
    bool observed = false;
    // (always two launches per env step: the observations are a second launch either way, and with the rows of t + 1 riding in it the
    // two launches are shorter than the single-launch / owner-wave plans -- profiles/r04_rl_small_batches.txt, r04_rl_owner.txt)
    if (two) {
      s->run_chains = 2;
      for (int c = 0; c < 2; ++c) launch_step(s, t + k, nullptr, k > 0 ? 1 : 0, &observed, k == 0 ? fold : nullptr, c, false);
      s->run_chains = 1;
      if ((last && (obs || rewards)) || !observed) join_forked(s);
    } else
      if ((rc = launch_step(s, t + k, nullptr, k > 0 ? 1 : 0, &observed, k == 0 ? fold : nullptr, -1, false)) != PEDN_OK) return rc;
      if (k == 0 && fold_stage && (rc = stage_commit(s, fold_stage)) != PEDN_OK) return rc;   // its only reader has been launched
    HIP_TRY(s, hipGetLastError());
    if (!observed) {
      if ((rc = pedn_rl_observe(s, t + k, k > 0, last ? obs : nullptr, last ? rewards : nullptr)) != PEDN_OK) return rc;
    } else if (last && (obs || rewards)) {
      if ((rc = rl_fetch(s, obs, rewards)) != PEDN_OK) return rc;
    }
  }
  return PEDN_OK;
}

// ---- device-resident step clock: env steps whose launches have constant arguments (a captured graph of them can be replayed)
static int clock_end(pedn_sim* s) {
  if (!s->clocked) return PEDN_OK;
  s->clocked = false;
  // the clocked steps may sit on a caller's stream (or in a graph replayed on one): everything on the device first
  HIP_TRY(s, hipDeviceSynchronize());
  int32_t h[4] = {0, 0, 0, 0};
  HIP_TRY(s, hipMemcpy(h, s->d_clock, sizeof(h), hipMemcpyDeviceToHost));
  const int t_now = h[0];
  if (t_now > s->clock_t0) {   // steps clock_t0 .. t_now - 1 ran
    s->last_t = t_now - 1;
    s->step_epoch += t_now - s->clock_t0;
    s->tp_ready = (s->v.n_trow > 0 && t_now <= s->v.T1 - 1) ? t_now : -1;   // link_turn_kernel(t_now - 1) left the fractions of t_now
    s->link_pending = -1;
  }
  s->valid_hi = s->v.valid_hi = h[2];
  return PEDN_OK;
}

int pedn_rl_clock_begin(pedn_sim* s, int32_t t) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  if (t < 1 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 1..T");
  if (!s->fuse_obs || !s->fuse_tp || s->v.n_pairs_corr == 0 || s->node_lp)
    return fail(s, PEDN_E_ARG, "the clocked step needs the fused second launch (PEDN_FUSE_OBS / PEDN_FUSE_TP), physical links and the classic node model");
  HIP_TRY(s, hipSetDevice(s->device));
  pending_links_first(s);   // (ends a clocked section that is still open)
  int rc;
  if (t - 1 > s->valid_hi && (rc = catch_up(s, t - 1)) != PEDN_OK) return rc;
  // The fractions of step t must be in place: the stand-alone launch that would compute them reads the gate widths AFTER the actions of
  // step t are applied (capacity fallback, path_finder.py:575-576) -- only the ordinary pedn_rl_step can order that.  True for every
  // step but the first one after a reset or a setter.
  if (s->v.n_trow > 0 && s->tp_ready != t)
    return fail(s, PEDN_E_ARG, "turning fractions of step " + std::to_string(t) + " are not prepared: run this step with pedn_rl_step first");
  hipLaunchKernelGGL(set_clock_kernel, dim3(1), dim3(64), 0, s->stream, s->d_clock, t, s->valid_hi);
  HIP_TRY(s, hipGetLastError());
  s->clocked = true;
  s->clock_t0 = t;
  return PEDN_OK;
}

int pedn_rl_step_clocked(pedn_sim* s, const double* actions, int32_t action_gap, void* stream) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->clocked) return fail(s, PEDN_E_ARG, "pedn_rl_clock_begin has not been called (or the clocked section was ended by another call)");
  if (action_gap < 1) return fail(s, PEDN_E_ARG, "action_gap < 1");
  HIP_TRY(s, hipSetDevice(s->device));
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  const DevView& v = s->v;
  RlView q = s->rl;
  const double* fold = nullptr;
  if (actions) {
    if (s->rl_fold) fold = actions;   // gater-only agent set: the consuming node_kernel wave clips and applies the action
    else {
      RlView qq = q;
      qq.actions = const_cast<double*>(actions);
      const size_t n = (size_t)q.A * v.RS;
      hipLaunchKernelGGL(rl_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, qq);
    }
  }
  const unsigned rgroups = (unsigned)(v.RS / 64);
  const bool fused = v.n_trow > 0;   // the fractions of t + 1 ride in the second launch (its workgroups idle behind the last step)
  const unsigned nlb = link_blocks(v, v.pr != 0);
  const unsigned ntb = fused ? (unsigned)((v.n_trow + 3) / 4) * rgroups : 0u, nth = fused ? (unsigned)s->n_tf_heavy_quads * rgroups : 0u;
  const unsigned nob = (unsigned)q.n_agents * rgroups;
  for (int k = 0; k < action_gap; ++k) {
    DevView vn = v;
    vn.rl_actions = k == 0 ? fold : nullptr;
    hipLaunchKernelGGL(clocked_node_kernel_for(s), dim3(rgroups, (unsigned)s->n_blocks), dim3(512), s->node_lds, st, vn, -1);
    const dim3 grid(nlb + ntb + nob), block(256);
    const int acc = k > 0 ? 1 : 0;
    if (v.pr) { if (v.hist) hipLaunchKernelGGL((link_turn_kernel<true, true, true, true>), grid, block, 0, st, v, -1, nlb, ntb, nth, q, acc);
                else hipLaunchKernelGGL((link_turn_kernel<true, true, false, true>), grid, block, 0, st, v, -1, nlb, ntb, nth, q, acc); }
    else { if (v.hist) hipLaunchKernelGGL((link_turn_kernel<false, true, true, true>), grid, block, 0, st, v, -1, nlb, ntb, nth, q, acc);
           else hipLaunchKernelGGL((link_turn_kernel<false, true, false, true>), grid, block, 0, st, v, -1, nlb, ntb, nth, q, acc); }
  }
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_rl_clocked(pedn_sim* s) { return s && s->clocked ? 1 : 0; }

// What the launches of pedn_rl_step_clocked are made of: the device view they carry BY VALUE (pointers, sizes, per-replica-scenario
// switches -- not valid_hi, which a clocked launch takes from the clock), the RL view, and what selects kernels and grids.
uint64_t pedn_rl_clock_signature(pedn_sim* s) {
  if (!s) return 0;
  DevView v;
  memcpy(&v, &s->v, sizeof v);   // byte image, padding included (see below)
  v.valid_hi = 0;
  v.rl_actions = nullptr;
  uint64_t h = 1469598103934665603ull;   // FNV-1a
  auto mix = [&](const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  };
  static_assert(std::is_trivially_copyable<DevView>::value && std::is_trivially_copyable<RlView>::value, "hashed as bytes");
  // (hashed as bytes, padding included: s->v lives in the value-initialised pedn_sim and is only ever changed field by field, so
  // its padding keeps the image it was created with; a spurious difference would cost one more capture, never a stale graph)
  mix(&v, sizeof v);
  mix(&s->rl, sizeof s->rl);
  const int64_t sel[8] = {s->rl_fold, s->n_tf_heavy_quads, s->n_blocks, (int64_t)s->node_lds, s->max_degree, s->node_lp, s->fuse_tp, s->fuse_obs};
  mix(sel, sizeof sel);
  return h;
}

int pedn_rl_clock_end(pedn_sim* s, int32_t* t) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  const int rc = clock_end(s);
  if (rc != PEDN_OK) return rc;
  if (t) *t = s->last_t + 1;
  return PEDN_OK;
}

int pedn_rl_step_many(pedn_sim** sims, int32_t n, const double* actions, int32_t t, int32_t action_gap, float* obs, float* rewards) {
  if (!sims || n < 1) return fail(nullptr, PEDN_E_ARG, "no engines");
  for (int k = 0; k < n; ++k)
    if (!sims[k] || !sims[k]->rl_ready) return fail(sims[k], PEDN_E_ARG, "null handle or pedn_rl_configure has not been called");
  size_t row = 0;
  for (int k = 0; k < n; ++k) {   // every engine's launches first (own stream each: they overlap) ...
    pedn_sim* s = sims[k];
    const int rc = pedn_rl_step(s, actions ? actions + row * (size_t)s->rl.A : nullptr, 0, t, action_gap, nullptr, nullptr);
    if (rc != PEDN_OK) return rc;
    row += (size_t)s->v.R;
  }
  row = 0;
  for (int k = 0; k < n; ++k) {   // ... then one fetch each
    pedn_sim* s = sims[k];
    const int rc = pedn_rl_fetch(s, obs ? obs + row * (size_t)s->rl.O : nullptr, rewards ? rewards + row * (size_t)s->rl.n_agents : nullptr);
    if (rc != PEDN_OK) return rc;
    row += (size_t)s->v.R;
  }
  return PEDN_OK;
}

void* pedn_rl_device_ptr(pedn_sim* s, int32_t which) {
  if (!s || !s->rl_ready) return nullptr;
  return which == 0 ? (void*)s->rl.actions : which == 1 ? (void*)s->rl.obs : which == 2 ? (void*)s->rl.rew : nullptr;
}

int pedn_device_math(int32_t device, int32_t op, int32_t n, const double* a, const double* b, uint64_t seed, double* out) {
  if (n <= 0 || !a || !out) return fail(nullptr, PEDN_E_ARG, "bad argument");
  HIP_TRY(nullptr, hipSetDevice(device));
  struct Buf {   // freed on every way out
    double* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
  } da, db, dout;
  HIP_TRY(nullptr, hipMalloc((void**)&da.p, (size_t)n * 8));
  HIP_TRY(nullptr, hipMalloc((void**)&dout.p, (size_t)n * 8));
  HIP_TRY(nullptr, hipMemcpy(da.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
  if (b) {
    HIP_TRY(nullptr, hipMalloc((void**)&db.p, (size_t)n * 8));
    HIP_TRY(nullptr, hipMemcpy(db.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(device_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, n, (const double*)da.p, (const double*)db.p,
                     (uint32_t)seed, (uint32_t)(seed >> 32), dout.p);
  HIP_TRY(nullptr, hipGetLastError());
  HIP_TRY(nullptr, hipDeviceSynchronize());
  HIP_TRY(nullptr, hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

}  // extern "C"

#ifdef PEDN_PHASE_PROFILE
// profiling build only (make phase-profile): read (zero = 0) or clear (zero = 1) the 16 phase accumulators of node_kernel
extern "C" int pedn_debug_tphases(unsigned long long* out, int zero) {
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_tphase)) != hipSuccess) return -1;
  if (zero) return (int)hipMemset(dev, 0, sizeof(unsigned long long) * 4096 * 8);
  return (int)hipMemcpy(out, dev, sizeof(unsigned long long) * 4096 * 8, hipMemcpyDeviceToHost);
}
// [workgroup][role 0 long turning-fraction rows / 1 link update / 2 short rows / 3 observations, start, end, 0] of the LAST launch of
// link_turn_kernel (zero = 1 clears)
extern "C" int pedn_debug_lt_timeline(unsigned long long* out, int n_blocks, int zero) {
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_lt_time)) != hipSuccess) return -1;
  if (zero) return (int)hipMemset(dev, 0, sizeof(unsigned long long) * PEDN_LT_BLOCKS * 4);
  if (n_blocks > PEDN_LT_BLOCKS) n_blocks = PEDN_LT_BLOCKS;
  return (int)hipMemcpy(out, dev, sizeof(unsigned long long) * (size_t)n_blocks * 4, hipMemcpyDeviceToHost);
}
// raw accumulators of the first n_waves waves of the grid: [wave][12] = 9 phase sums, -, count, lifetime (tools/pack_analysis.py)
extern "C" int pedn_debug_phase_waves(unsigned long long* out, int n_waves) {
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_phase)) != hipSuccess) return -1;
  if (n_waves > PEDN_PHASE_WAVES) n_waves = PEDN_PHASE_WAVES;
  return (int)hipMemcpy(out, dev, sizeof(unsigned long long) * (size_t)n_waves * 12, hipMemcpyDeviceToHost);
}
extern "C" int pedn_debug_phases(unsigned long long* out, int zero) {
  const size_t n = (size_t)PEDN_PHASE_WAVES * 12;
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_phase)) != hipSuccess) return -1;
  if (zero) return (int)hipMemset(dev, 0, n * sizeof(unsigned long long));
  std::vector<unsigned long long> h(n);
  if (hipMemcpy(h.data(), dev, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  for (size_t w = 0; w < (size_t)PEDN_PHASE_WAVES; ++w)
    for (int i = 0; i < 12; ++i) out[i] += h[w * 12 + i];
  return 0;
}
#endif
