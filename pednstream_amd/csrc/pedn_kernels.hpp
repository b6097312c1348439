// pedn_kernels.hpp -- the HIP kernels of the network_loading hot path and the device functions they share.
// Included once by pedn_hip.hip (single translation unit, compiled with -ffp-contract=off).
//
// Kernels per step t (reference network.py:266-287):
//   node_kernel       one wave per (node slot, 64 replicas), one block per bin of nodes with <= 8 slots in total:
//                     sending flow of the slot's incoming link, receiving flow of its outgoing link, the node's flow
//                     distribution through LDS, cumulative counts
//                                                                 node.py:164-221,230-242,272-300; link.py:216-416
//                     <LU>: the slot wave first performs the link update of step t-1 for its incoming link (every input was written
//                     by earlier launches): ONE launch per step for models without device-computed fractions (pedn_hip.hip: launch_step)
//                     <LU, TF>: and computes its own row of turning fractions too (small batches of models with short dynamic rows)
//   link_kernel_1r    the launch behind it under the two-launch plan when no node has dynamic fractions -- and the one that performs a
//                     link update still pending under the owner-wave plan: one lane per (corridor = link pair, replica):
//                     pedestrians, density, fundamental diagram, travel time and its moving average
//                                                                                       link.py:133-188; functions.py:112-134
//   link_turn_kernel  the launch behind it otherwise: [turning fractions of t+1, long rows | link update of t, two replicas per
//                     lane | turning fractions of t+1, short rows | RL observations of t] as independent workgroups
//   turn_frac_kernel  the turning fractions on their own (first step of an episode): one wave per (row of a dynamic node, 64
//                     replicas): logit route choice -> the row's turning fractions      path_finder.py:561-737
// plus the batched RL glue (rl_apply_kernel, rl_observe_kernel), the per-replica scenario randomisers (rand_*_kernel, pod_*_kernel),
// state initialisation and host<->device helpers.
#pragma once
#include "pedn_math.hpp"
#include "pedn_types.hpp"

#include <type_traits>

using namespace pedn;

// Optional wave-lifetime profile of node_kernel (make phase-profile -> libpedn_hip_phase.so, tools/phase_profile.py):
// s_memtime stamps at the phase boundaries, summed over all waves into g_phase.  Not part of the product build.
#ifdef PEDN_PHASE_PROFILE
#define PEDN_PHASE_WAVES (1 << 17)
__device__ unsigned long long g_phase[PEDN_PHASE_WAVES * 12];  // [wave of the grid][phase]: plain stores, no contended atomics
#define PH(i, dep) do { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) : "v"(dep) : "memory"); ph[i] = _t; } while (0)
__device__ unsigned long long g_tphase[4096 * 8];  // turn_frac_body: [row][stamp] of replica group 0
#define TPH(i, dep) do { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) : "v"(dep) : "memory"); tph[i] = _t; } while (0)
// timeline of link_turn_kernel's workgroups (tools/lt_timeline.py): [workgroup][role, start, end of its last wave, 0] in ticks of the
// constant 100 MHz clock (s_memrealtime, the same counter on every XCD)
#define PEDN_LT_BLOCKS (1 << 15)
__device__ unsigned long long g_lt_time[PEDN_LT_BLOCKS * 4];
#else
#define PH(i, dep)
#define TPH(i, dep)
#endif

// History row of time index t: all T+1 rows are kept in full-record mode (mask = all ones); in recent-history mode
// (PEDN_HIST_RECENT) the fields nothing looks far back into are rings of a power-of-two number of rows -- see pedn_create.
// HIST is a template parameter of every kernel that touches a history: in full-record mode the index is the time index itself, so
// the row arithmetic of accesses to different fields at the same time index is shared (one AND per access and field cost
// node_kernel +57 % scalar instructions and ~1 us when the masks were applied unconditionally).
#define R64(F, t) (HIST ? ((t) & v.m64[F]) : (t))
#define R32(G, t) (HIST ? ((t) & v.m32[G]) : (t))

__device__ __forceinline__ size_t at(int t, int col, int cols, int RS, int r) {
  return ((size_t)t * (size_t)cols + (size_t)col) * (size_t)RS + (size_t)r;
}
// start of the 64-replica segment [r0, r0 + 64) of row (t, col): wave-uniform, so `rowp(...)[lane]` is a scalar base plus a
// 32-bit lane offset (no 64-bit per-lane address arithmetic)
template <typename T>
__device__ __forceinline__ T* rowp(T* base, int t, int col, int cols, int RS, int r0) {
  return base + (((size_t)t * (size_t)cols + (size_t)col) * (size_t)RS + (size_t)r0);
}
__device__ __forceinline__ float clip01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
__device__ __forceinline__ double clip_d(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }  // np.clip
// Link parameters as seen by one lane: the shared record, or (PR) the replica's own k_critical / k_jam / free-flow speed
// and the quantities derived from them on the host.
template <bool PR>
__device__ __forceinline__ LinkP lane_params(const DevView& v, const LinkP& P, int l, int r) {
  LinkP Q = P;
  if (PR) {
    // the record as two 16-byte loads
    const double2* rec = reinterpret_cast<const double2*>(v.prm + ((size_t)l * v.RS + r));
    const double2 lo = rec[0], hi = rec[1];
    Q.kc = lo.x; Q.kj = lo.y; Q.vf = hi.x;
    const int w0 = __double2loint(hi.y), w1 = __double2hiint(hi.y);
    Q.tt0 = __int_as_float(w0);
    Q.fft = (int)(short)(w1 & 0xffff); Q.tau_sw = w1 >> 16;
    Q.derive();
  }
  return Q;
}

__device__ __forceinline__ int wrap_idx(int i, int T1, uint32_t& fl) {
  if (i < 0) i += T1;
  if (i < 0 || i >= T1) { fl |= PEDN_F_INDEX; return 0; }
  return i;
}

// Link.get_density (link.py:190-197) / Separator.get_density (:427-428) at history index t
template <bool HIST>
__device__ __forceinline__ float dens_at(const DevView& v, const LinkP& P, int l, int t, int r) {
  if (P.sep) return v.f32[G_K][at(R32(G_K, t), l, v.L, v.RS, r)];
  float n = v.f32[G_N][at(R32(G_N, t), l, v.L, v.RS, r)] + v.f32[G_N][at(R32(G_N, t), P.rev, v.L, v.RS, r)];
  return n / P.area32;
}

// Everything of step t' = t-1 a slot wave needs from HBM whose address does not depend on data: fetched as ONE batch of
// independent loads at the top of node_kernel (the wave then waits once instead of once per dependent use; the kernel is
// latency-bound: 76 % of its wave-cycles were s_waitcnt with the loads left where the formulas use them).
struct SlotIn {
  float n_in, n_out, k_in, att_in;        // num_pedestrians[t'] of the incoming / outgoing link, density and avg travel time
  double co_in, s_prev, front_in, sepw_in;   // incoming link: cumulative_outflow[t'], sending_flow[t'-1], front gate, separator width
  double co_sw, ci_out, r_prev, back_out, sepw_out;  // outgoing link: cumulative_outflow[t'+1-tau_sw], cumulative_inflow[t'],
                                                     // receiving_flow[t'-1], back gate, separator width
};

// Link.cal_sending_flow (link.py:216-370) incl. get_outflow (:199-214) for t' >= free_flow_tau
// ci_look = cumulative_inflow[max(0, t' + 1 - tau)] of the link, loaded by the caller (tau from x.att_in as below) so that the load is
// in flight while the caller draws the receiving side's binomial
// vhi: DevView.valid_hi as it stands for THIS step (the host's value, or the device clock's -- see node_kernel)
template <bool HIST>
__device__ double send_flow(const DevView& v, const LinkP& P, int l, int tp, int r, const SlotIn& x, double ci_look, const int vhi, uint32_t& fl) {
  const int RS = v.RS, T1 = v.T1;
  const float nself = x.n_in, nrev = x.n_out, kk = x.k_in, att = x.att_in;
  float dens = P.sep ? kk : (nself + nrev) / P.area32;
  int tau = __float2int_rn(att / (float)v.dt);  // link.py:260
  if (tau <= 0) fl |= PEDN_F_SAME_STEP;
  int idx = tp + 1 - tau;
  if (idx < 0) idx = 0;
  float cf = clip01((kk - P.kc32) / P.dk32);  // link.py:282
  double ff = ci_look - x.co_in;  // the one data-dependent look-back of the common path (loaded by the caller at row idx)
  if (!(ff > 0.0)) ff = 0.0;
  double bnd = (double)(cf * nself) + (double)(1.0f - cf) * ff;  // link.py:284-288
  double smax = x.front_in * P.kc * P.vf * v.dt;                 // link.py:296
  double s = smax < bnd ? smax : bnd;
  double orig = s;
  RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)tp, 0u};
  if (s > 0.0) {
    const bool free_flow = dens <= P.kc32;  // link.py:323
    double i0 = 0.0, i1 = 0.0, i2 = 0.0, i3 = 0.0;
    if (free_flow) {
      const double* in = v.f64[F_IN];
      const int w0 = wrap_idx(tp - tau, T1, fl), w1 = wrap_idx(tp - tau - 1, T1, fl), w2 = wrap_idx(tp - tau - 2, T1, fl), w3 = wrap_idx(tp - tau - 3, T1, fl);
      i0 = in[at(w0, l, v.Lall, RS, r)], i1 = in[at(w1, l, v.Lall, RS, r)];
      i2 = in[at(w2, l, v.Lall, RS, r)], i3 = in[at(w3, l, v.Lall, RS, r)];
      // lazy reset: a wrapped index lands in a row of the FUTURE, which holds the last episode's value instead of the untouched 0
      i0 = w0 > vhi ? 0.0 : i0; i1 = w1 > vhi ? 0.0 : i1; i2 = w2 > vhi ? 0.0 : i2; i3 = w3 > vhi ? 0.0 : i3;
    }
    // Philox call 0 of the sending binomial needs no data: drawn while the four loads are in flight
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    key.site = 0u;
    if (!v.meanfield) rng_words(key, 0u, w);
    float rf = clip01(dens / P.kj32);                          // link.py:315
    float p = 0.7f + (float)(0.85 - 0.7) * pedn_powf(rf, 0.8f);     // link.py:317
    bool diffusion_used = false;
    if (free_flow) {
      float F = 1.0f / (1.0f + P.gamma32 * att);
      float G = 1.0f - F;
      float G2, G3;
      pedn_powf_2_3(G, G2, G3);
      double d = (double)F * i0 + (double)(F * G) * i1 + (double)(F * G2) * i2 + (double)(F * G3) * i3;
      d = ceil(d);
      if (d > 0.0) {  // link.py:326-330
        double mix = 0.8 * d + (1 - 0.8) * s;
        s = floor(mix < s ? mix : s);
        diffusion_used = true;
      }
    }
    if (!diffusion_used) s = rng_binomial_w((long long)floor(s), (double)p, key, w, v.meanfield);  // link.py:336-338,342-344
    if (s < 0.0) fl |= PEDN_F_NEG_SENDING;
  }
  if (P.act > 0.0 && s > 1.0) {  // link.py:351-358
    key.site = 1u;
    s -= rng_binomial((long long)floor(s), P.act, key, v.meanfield);
  }
  if (!(s > 0.0)) s = 0.0;
  double sm = floor(0.8 * s + 0.2 * x.s_prev);  // link.py:364
  s = orig < sm ? orig : sm;
  if (s < 0.0) fl |= PEDN_F_NEG_SENDING;
  return s;
}

// Link/Separator.cal_receiving_flow[_with_reverse] (link.py:372-416,480-512); the reverse link is the slot's incoming link
// rp = Binomial(trunc(num_pedestrians of the reverse link), 0.9) (link.py:381-382), drawn by the caller (recv_reverse_peds)
__device__ __forceinline__ double recv_reverse_peds(const DevView& v, const LinkP& P, int l, int tp, int r, const SlotIn& x, uint32_t& fl) {
  if (P.sep) return 0.0;
  const float nrev = x.n_in;
  if (nrev < 0.0f) fl |= PEDN_F_NEG_BINOM;
  RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)tp, 2u};
  return rng_binomial((long long)nrev, 0.9, key, v.meanfield);
}
__device__ double recv_flow(const DevView& v, const LinkP& P, int l, int tp, int r, const SlotIn& x, double s_rev, double rp, uint32_t& fl) {
  double kjA = P.sep ? P.kj * (P.length * x.sepw_out) : P.kjA;
  int tsw = P.tau_sw;
  double b;
  if (P.sep) {
    if (tp + 1 - tsw < 0) b = kjA;
    else {
      if (tsw <= 0) fl |= PEDN_F_SAME_STEP;
      b = x.co_sw + kjA - x.ci_out;
    }
  } else {
    if (tp + 1 - tsw < 0) b = kjA - rp;
    else {
      if (tsw <= 0) fl |= PEDN_F_SAME_STEP;
      b = x.co_sw + kjA - rp - x.ci_out;
      if (!(b > 0.0)) b = 0.0;
    }
  }
  double rmax = x.back_out * P.kc * P.vf * v.dt;  // link.py:393
  double rr = rmax < b ? rmax : b;
  if (!(rr > 0.0)) rr = 0.0;
  if (x.r_prev >= 0.0) {  // link.py:400-401
    double sm = floor(rr * 0.8 + x.r_prev * 0.2);
    rr = sm < rr ? sm : rr;
  }
  if (P.sep) return rr > 0.0 ? rr : 0.0;
  rr = rr - s_rev;  // link.py:415-416
  return rr > 0.0 ? rr : 0.0;
}

// BiDirectionalFd.__call__ + the travel-time part of Link.update_speeds for one direction and one replica; pure arithmetic
struct SpeedOut { float spd, tt, lf, att, rs; };

// the speed noise of (link, step, replica) (functions.py:132-133): depends on no simulation state, so callers draw it while
// their loads are in flight
__device__ __forceinline__ double speed_noise(const DevView& v, const LinkP& P, int l, int t, int r) {
  if (!(P.noise > 0.0) || v.meanfield) return 0.0;
  RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)t, 3u};
  return P.noise * rng_z(key);
}

__device__ __forceinline__ SpeedOut speed_calc(const DevView& v, const LinkP& P, int l, int t, int r, float ks, float ko,
                                               float rsum_prev, float tt_old, double nz) {
  float ke = P.sep ? ks : ks + (float)P.bi * ko;  // functions.py:113
  bool is64;  // true: the speed is a Python float (binary64) at this point, false: np.float32
  double v64 = 0.0;
  float v32 = 0.0f;
  if (P.fd == 2 && ke <= (float)P.kc) {
    v32 = (float)P.vf * (1.0f - ke / (float)P.kj);
    is64 = false;
  } else if (ke <= (float)P.kc) {
    v64 = P.vf;
    is64 = true;
  } else {
    if (P.fd == 0) v32 = (float)((P.kc * P.vf) / (P.kj - P.kc)) * ((float)P.kj / ke - 1.0f);
    else if (P.fd == 1) v32 = ((float)(-P.vf) * (ke - (float)P.kj)) / (float)(P.kj - P.kc);
    else v32 = (float)(P.vf * P.kc) * (1.0f / ke - (float)(1 / P.kj));
    is64 = false;
    if (!(v32 > 0.0f)) { v64 = 0.0; is64 = true; }  // Python max(0, x) returns the int 0
  }
  if (P.noise > 0.0) {  // functions.py:132-133, nz = speed_noise(...)
    if (is64) v64 = v64 + nz;
    else v32 = v32 + (float)nz;
  }
  if (is64) { if (!(v64 > 0.0)) v64 = 0.0; }
  else if (!(v32 > 0.0f)) { v64 = 0.0; is64 = true; }
  SpeedOut o;
  o.spd = is64 ? (float)v64 : v32;
  if (is64) o.tt = v64 > 0.0 ? (float)(P.length / v64) : (float)(P.length / 0.05);  // link.py:177
  else o.tt = (float)P.length / v32;
  o.lf = ks * o.spd;               // link.py:181
  o.rs = rsum_prev + o.tt;         // link.py:183-186, float32 running sum
  o.att = P.tt0;
  if (t >= v.W) {
    o.rs = o.rs - tt_old;
    o.att = o.rs / (float)v.W;
  }
  return o;
}

// ------------------------------------------------------------------------------------------------- kernels
// wave-uniform values out of a per-lane register: a record is fetched with ONE vector load (lane k holds its k-th word)
// and its fields are broadcast with v_readlane.  A chain of scalar loads would wait for each record in turn; vector loads
// return in order and can be issued a record ahead.
__device__ __forceinline__ int rdl(int x, int k) { return __builtin_amdgcn_readlane(x, k); }
__device__ __forceinline__ float rdl_f(int x, int k) { return __int_as_float(rdl(x, k)); }
__device__ __forceinline__ double rdl_d(int x, int k) { return __hiloint2double(rdl(x, k + 1), rdl(x, k)); }
// a wave-uniform value as a per-lane one the compiler cannot see through: conditions on it become v_cmp / v_cndmask
// instead of scalar branches, which would cut a block of independent arithmetic chains into pieces that run one after the other
__device__ __forceinline__ int as_vector(int x) {
  int y;
  asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(x));
  return y;
}

// Turning fractions of step t for one row (= incoming slot) of one dynamic node and 64 replicas, by ONE wave
// (PathFinder.calculate_node_turning_fractions, path_finder.py:717-737).  Every softmax group (od, up) and every turn
// (up, down) belongs to exactly one row, so rows are independent and the wave never meets a barrier:
//   phase 1  density and capacity of the outgoing links the row's groups refer to (update_node_turn_probs :565-579) --
//            one batch of independent loads, kept in registers
//   phase 2  P(down | up, od) of every group with more than one downstream (:580-589) into the wave's share of the block's
//            LDS rows (lane-private; rows that do not fit go to ent_p); a group with a single downstream has P = e/e = 1
//            exactly and is never evaluated.  Four groups are evaluated side by side in straight-line code: one group is a
//            chain of ~40 dependent binary64 operations and a table look-up, and a junction of delft has 24 of them in a row
//   phase 3  tf[turn] = sum over the turn's (od) products P(down | up, od) * P(od | up) in the reference's order (:668-686),
//            check_fractions of the row (:691-715), into tfd[t & 1]
// Only rows with at least one such group come here; a row whose products are all constant has replica-independent
// fractions that the host tabulates per step (tabulate_pair_pod) and node_kernel reads with scalar loads.
// (Inside node_kernel the rows that sum ~50 products were the slowest waves of their block and of the launch: 24 % of a
// wave's lifetime on delft was the barrier behind them.)  P(od | up) is replica independent (:599-615), tabulated per
// (step, product) on the host.  Row record, group records and product tables arrive by vector loads (see rdl).
// FUSED: the wave runs inside link_turn_kernel next to the link update of step t-1, which has not stored
// num_pedestrians[t-1] / density[t-1] yet -- they are recomputed here from [t-2] and the flows of t-1 with the arithmetic of
// the link update (link.py:133-136), so the parts of that launch do not depend on each other.
// lds: PEDN_TF_LDS_DOUBLES doubles of the workgroup's LDS (the kernel owns the buffer: the parts of link_turn_kernel share one)
#define PEDN_TF_LDS_DOUBLES ((PEDN_TF_LDS_ROWS + PEDN_MAX_DEGREE - 1) * 64)
// INL: the row is computed by ONE wave of node_kernel for itself (the single-launch plan of small batches): `row` and `r0` are given,
// `lds` is the wave's private share (PEDN_TF_INL_LDS doubles: its probabilities' LDS rows rebased to 0 -- the row record's word 109
// holds the base the host gave the row inside its quad -- then PEDN_MAX_DEGREE - 1 rows in which the fractions are handed back), coop
// rows do not come here.
#define PEDN_TF_INL_LDS ((PEDN_TF_INL_ROWS + PEDN_MAX_DEGREE - 1) * 64)
template <bool PR, bool FUSED, bool HIST, bool INL = false>
__device__ __forceinline__ void turn_frac_body(const DevView& v, int t, unsigned block, double* lds, int inl_row = 0, int inl_r0 = 0,
                                               int inl_w0 = 0, int inl_w1 = 0) {
  double* const sP = lds;                               // [PEDN_TF_LDS_ROWS][64] probabilities of the workgroup's rows
  double* const sAcc = lds + PEDN_TF_LDS_ROWS * 64;     // [PEDN_MAX_DEGREE - 1][64] coop rows: the turns' sums on their way to wave 0
  constexpr int NE = PEDN_MAX_DEGREE - 1;
  const int RS = v.RS;
  const unsigned rgroups = (unsigned)(v.subRS / 64);
  const int lane = (int)(threadIdx.x & 63);
  const int wave = INL ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int row = INL ? inl_row : (int)(block / rgroups) * 4 + wave;
  const int r = INL ? inl_r0 + lane : v.sub0 + (int)(block % rgroups) * 64 + lane;
  if (row >= v.n_trow) return;  // wave-uniform; no barrier below
  const int* rw = v.trow_words + (size_t)row * PEDN_TROW_WORDS;
  const int w0 = INL ? inl_w0 : rw[lane], w1 = INL ? inl_w1 : rw[64 + lane];   // INL: the caller fetched the record with its own batch of loads
  const int m = rdl(w0, 0), turn0 = rdl(w0, 1), grp0 = rdl(w0, 2), n_grp = rdl(w0, 3), Q0 = rdl(w0, 4), Q1 = rdl(w0, 5);
  const int n_used = rdl(w0, 6), any_sep = rdl(w0, 7), overflow = rdl(w1, 43), coop = INL ? 0 : rdl(w1, 44), part = coop ? wave : 0;
  const int lds0 = INL ? rdl(w1, 45) : 0;   // first LDS row of this row inside its quad
  if (m < 0) return;  // padding record of a block with fewer than four rows
#ifdef PEDN_PHASE_PROFILE
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TPH(0, w0 + w1);
#endif
  uint32_t fl = 0;
  const int t2 = wrap_idx(t - 2, v.T1, fl);
  const bool ppr = v.pod_pr != 0;  // per-replica OD weights: tables indexed [product][replica] instead of [step][product]
  // ---- everything whose address is known now, issued together: tabulated turns, first chunk of the product tables, first
  // pair of group records, and (below) the phase-1 history values
  const int* gw = v.tgrp_words + (size_t)grp0 * 32;  // padded: reading a pair of records ahead stays in bounds
  int gcur = gw[(size_t)part * 64 + lane];
  const int nq = Q1 - Q0;
  const double* pod = v.pair_pod + (size_t)t * v.n_pair + Q0;
  int pcode = lane < nq ? v.pair_row[Q0 + lane] : -1;
  double pw = (lane < nq && !ppr) ? pod[lane] : 0.0;
  const double ttab = lane < m - 1 ? (ppr ? 0.0 : v.turn_tab[(size_t)t * v.n_turns + turn0 + lane]) : 0.0;

  // ---- phase 1
  float kf[NE];
  double cap[NE];
  auto entry = [&](int e, int& link, int& rev, int& sep, float& area32, double& vf, double& kc, double& length) {
    const int ww = e < 5 ? w0 : w1, b = e < 5 ? 8 + 10 * e : 10 * (e - 5);
    link = rdl(ww, b); rev = rdl(ww, b + 1); sep = rdl(ww, b + 2); area32 = rdl_f(ww, b + 3);
    vf = rdl_d(ww, b + 4); kc = rdl_d(ww, b + 6); length = rdl_d(ww, b + 8);
  };
  // E0..E1: a compile-time range of entries, loaded unconditionally (entries >= n_used repeat entry 0 in the record): the
  // compiler issues the whole batch before the first use.  SEP: some entry is a separator link.
  auto batch = [&](auto e0_t, auto e1_t, auto sep_t) {
    constexpr int E0 = decltype(e0_t)::value, E1 = decltype(e1_t)::value;
    constexpr bool SEP = decltype(sep_t)::value;
    float n_l[E1 - E0], n_r[E1 - E0], k_l[E1 - E0];
    double d_l[E1 - E0], d_r[E1 - E0], c_l[E1 - E0], sw[E1 - E0], snp[E1 - E0];
#pragma unroll
    for (int e = E0; e < E1; ++e) {
      int link, rev, sep; float area32; double vf, kc, length;
      entry(e, link, rev, sep, area32, vf, kc, length);
      const int i = e - E0;
      if (!FUSED) {
        n_l[i] = v.f32[G_N][at(R32(G_N, t - 1), link, v.L, RS, r)];
        n_r[i] = v.f32[G_N][at(R32(G_N, t - 1), rev, v.L, RS, r)];
        if (SEP) k_l[i] = v.f32[G_K][at(R32(G_K, t - 1), link, v.L, RS, r)];
      } else {
        n_l[i] = v.f32[G_N][at(R32(G_N, t - 2), link, v.L, RS, r)];
        n_r[i] = v.f32[G_N][at(R32(G_N, t - 2), rev, v.L, RS, r)];
        d_l[i] = v.f64[F_IN][at(R64(F_IN, t - 1), link, v.Lall, RS, r)] - v.f64[F_OUT][at(R64(F_OUT, t - 1), link, v.Lall, RS, r)];
        d_r[i] = v.f64[F_IN][at(R64(F_IN, t - 1), rev, v.Lall, RS, r)] - v.f64[F_OUT][at(R64(F_OUT, t - 1), rev, v.Lall, RS, r)];
        if (SEP) { sw[i] = v.sepw[(size_t)link * RS + r]; snp[i] = v.sepnp[(size_t)link * RS + r]; }
      }
      c_l[i] = v.f64[F_R][at(R64(F_R, t2), link, v.L, RS, r)];
    }
#pragma unroll
    for (int e = E0; e < E1; ++e) {
      int link, rev, sep; float area32; double vf, kc, length;
      entry(e, link, rev, sep, area32, vf, kc, length);
      const int i = e - E0;
      float pl = n_l[i], pr = n_r[i];
      if (FUSED) { pl = (float)((double)pl + d_l[i]); pr = (float)((double)pr + d_r[i]); }  // num_pedestrians[t-1], link.py:133-135
      float k = (pl + pr) / area32;  // Link.get_density, link.py:190-197
      if (SEP && sep) {              // Separator.get_density = density[t-1], link.py:427-428
        if (!FUSED) k = k_l[i];
        else k = snp[i] != 0.0 ? (float)((double)pl / (length * sw[i])) : pl / (float)(length * sw[i]);
      }
      double c = c_l[i];
      if (!(c >= 0.0)) {  // no receiving flow recorded yet: back_gate_width * v_f * k_c * dt (:575-576)
        const double vfr = PR ? v.prm[(size_t)link * RS + r].vf : vf, kcr = PR ? v.prm[(size_t)link * RS + r].kc : kc;
        c = v.back[(size_t)link * RS + r] * vfr * kcr * v.dt;
      }
      kf[e] = k;
      cap[e] = c;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I4 = std::integral_constant<int, 4>;
  using I7 = std::integral_constant<int, NE>;
#pragma unroll
  for (int e = 0; e < NE; ++e) { kf[e] = 0.0f; cap[e] = 0.0; }
  if (any_sep) batch(I0{}, I4{}, std::true_type{});
  else batch(I0{}, I4{}, std::false_type{});
  if (n_used > 4) {  // junctions with more than five arms
    if (any_sep) batch(I4{}, I7{}, std::true_type{});
    else batch(I4{}, I7{}, std::false_type{});
  }
  TPH(1, kf[0] + kf[1] + kf[2] + kf[3] + (float)(cap[0] + cap[1] + cap[2] + cap[3]) + (float)(gcur + pcode) + (float)(pw + ttab));
  // ---- phase 2
  auto put_prob = [&](int q, double p) {
    if (q >= PEDN_TF_LDS_ROWS) v.ent_p[(size_t)(q - PEDN_TF_LDS_ROWS) * RS + r] = p;
    else if (q >= 0) sP[(q - lds0) * 64 + lane] = p;
  };
  auto pick = [&](int idx, float& k, double& c) {  // entry idx of the row record, -1: virtual link (:577-579)
    k = 0.0f;
    c = 100.0;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      k = idx == u ? kf[u] : k;
      c = idx == u ? cap[u] : c;
    }
  };
  // the two groups of one record pair x NE_ entries as ONE straight-line block (no branch before the stores): their divisions,
  // exp chains and table look-ups overlap.  Absent groups / entries (e >= n) run on harmless values and are discarded.
  auto eval_pair = [&](auto ne_t, int gv, int na, int nb) {
    constexpr int NE_ = decltype(ne_t)::value;
    double ce[2][NE_], ex[2][NE_], xx[2][NE_], sumc[2], esum[2], p[2][NE_];
    float ke[2][NE_];
    bool any_special = false;
    const int nav = as_vector(na), nbv = as_vector(nb);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = h * 32, n = h ? nbv : nav;
#pragma unroll
      for (int e = 0; e < NE_; ++e) pick(as_vector(rdl(gv, b + 2 + e)), ke[h][e], ce[h][e]);
      sumc[h] = ce[h][0];
#pragma unroll
      for (int e = 1; e < NE_; ++e) { const double s2 = sumc[h] + ce[h][e]; sumc[h] = e < n ? s2 : sumc[h]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = h * 32, n = h ? nbv : nav;
      const int allphys = as_vector(rdl(gv, b + 1));
#pragma unroll
      for (int e = 0; e < NE_; ++e) {
        // (x - 2)+ / 8 in float32 when every downstream is a physical link, else in binary64 (:581,583); a division by 8 is
        // the multiplication by 0.125, bit for bit
        float xf = ke[h][e] - 2.0f;
        if (!(xf > 0.0f)) xf = 0.0f;
        double xd = (double)ke[h][e] - 2.0;
        if (!(xd > 0.0)) xd = 0.0;
        const double nd = allphys ? (double)((float)v.pf_beta * (xf * 0.125f)) : v.pf_beta * (xd * 0.125);
        const double u = rdl_d(gv, b + 16 + 2 * e) + nd - (v.pf_omega * ce[h][e]) / (sumc[h] + 1e-6) + v.pf_eps;
        xx[h][e] = e < n ? -v.pf_temp * u : -1.0;
        bool sp;
        ex[h][e] = pedn_exp_main(xx[h][e], sp);
        any_special = any_special || sp;
      }
    }
    if (any_special) {  // |x| < 2^-54 or >= 512: the full function (rare)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < NE_; ++e) ex[h][e] = pedn_exp(xx[h][e]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = h ? nbv : nav;
      esum[h] = ex[h][0];
#pragma unroll
      for (int e = 1; e < NE_; ++e) { const double s2 = esum[h] + ex[h][e]; esum[h] = e < n ? s2 : esum[h]; }
#pragma unroll
      for (int e = 0; e < NE_; ++e) p[h][e] = ex[h][e] / esum[h];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = h * 32, n = h ? nb : na;
#pragma unroll
      for (int e = 0; e < NE_; ++e)
        if (e < n) put_prob(rdl(gv, b + 9 + e), p[h][e]);
    }
  };
  auto eval_slow = [&](int gv, int b, int n) {  // any group size
    const int allphys = rdl(gv, b + 1);
    double ce[NE], ex[NE];
    float ke[NE];
    double sumc = 0.0, esum = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e < n) {
        pick(rdl(gv, b + 2 + e), ke[e], ce[e]);
        sumc = (e == 0) ? ce[e] : sumc + ce[e];
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e < n) {
        double nd;
        if (allphys) {
          float x = ke[e] - 2.0f;
          if (!(x > 0.0f)) x = 0.0f;
          nd = (double)((float)v.pf_beta * (x / 8.0f));
        } else {
          double x = (double)ke[e] - 2.0;
          if (!(x > 0.0)) x = 0.0;
          nd = v.pf_beta * (x / 8.0);
        }
        const double u = rdl_d(gv, b + 16 + 2 * e) + nd - (v.pf_omega * ce[e]) / (sumc + 1e-6) + v.pf_eps;
        ex[e] = pedn_exp(-v.pf_temp * u);
        esum = (e == 0) ? ex[e] : esum + ex[e];
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (e < n) put_prob(rdl(gv, b + 9 + e), ex[e] / esum);
  };
  // pairs of groups: this wave takes every `pstep`-th pair from `part` on (a row with many groups is shared out over the
  // four waves of its workgroup: coop)
  const int n_pairs = (n_grp + 1) / 2, pstep = coop ? 4 : 1;
  for (int pi = part; pi < n_pairs; pi += pstep) {
    const int gnext = pi + pstep < n_pairs ? gw[(size_t)(pi + pstep) * 64 + lane] : 0;
    // the record behind the row's last group belongs to another row: size 0, evaluated on harmless values, nothing stored
    const int na = rdl(gcur, 0), nb = 2 * pi + 1 < n_grp ? rdl(gcur, 32) : 0;
    const int nmax = max(na, nb);
    // (one softmax group at a time at 80 VGPRs / six workgroups per CU was measured slower: profiles/r04_lean_link_turn.txt)
    if (nmax <= 2 && !(v.tf_general & 1)) eval_pair(std::integral_constant<int, 2>{}, gcur, na, nb);
    else if (nmax <= 3 && !(v.tf_general & 1)) eval_pair(std::integral_constant<int, 3>{}, gcur, na, nb);
    else { eval_slow(gcur, 0, na); eval_slow(gcur, 32, nb); }
    gcur = gnext;
  }
  TPH(2, lane);
  // ---- phase 3
  if (!INL && coop) __syncthreads();  // workgroup-uniform: the four waves hold the same row
  double* out = v.tfd[t & 1];
  int chunk0 = 0;  // products [chunk0, chunk0 + 64) of the row are in pcode / pw
  auto prob_of = [&](int code) -> double {
    return code < 0 ? 1.0 : code < PEDN_TF_LDS_ROWS ? sP[(code - lds0) * 64 + lane] : v.ent_p[(size_t)(code - PEDN_TF_LDS_ROWS) * RS + r];
  };
  auto turn_sum = [&](int jj) -> double {
    const int tq0 = rdl(w1, 20 + 3 * jj) - Q0, tq1 = rdl(w1, 21 + 3 * jj) - Q0, mode = rdl(w1, 22 + 3 * jj);
    double acc = 0.0;
    if (mode) {
      acc = ppr ? v.turn_tab_r[(size_t)(turn0 + jj) * RS + r] : __hiloint2double(rdl(__double2hiint(ttab), jj), rdl(__double2loint(ttab), jj));
    } else if (!ppr && !overflow && nq <= 64 && !(v.tf_general & 2)) {
      // every probability is the constant 1 or in LDS: four products per pass, no branch in between
      auto lds_prob = [&](int q) -> double {
        const int c = rdl(pcode, q);
        const double e = sP[(c < 0 ? 0 : c - lds0) * 64 + lane];
        return c < 0 ? 1.0 : e;
      };
      int q = tq0;
      for (; q + 4 <= tq1; q += 4) {
        double e[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          e[k] = lds_prob(q + k);
          w[k] = __hiloint2double(rdl(__double2hiint(pw), q + k), rdl(__double2loint(pw), q + k));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += e[k] * w[k];
      }
      for (; q < tq1; ++q) acc += lds_prob(q) * __hiloint2double(rdl(__double2hiint(pw), q), rdl(__double2loint(pw), q));
    } else {
      for (int q = tq0; q < tq1; ++q) {
        if (q < chunk0 || q >= chunk0 + 64) {  // rows with more than 64 products: another chunk of the tables
          chunk0 = q;
          pcode = chunk0 + lane < nq ? v.pair_row[Q0 + chunk0 + lane] : -1;
          pw = (chunk0 + lane < nq && !ppr) ? pod[chunk0 + lane] : 0.0;
        }
        const int code = rdl(pcode, q - chunk0);
        const double w = ppr ? v.pair_pod_r[(size_t)(Q0 + q) * RS + r]
                             : __hiloint2double(rdl(__double2hiint(pw), q - chunk0), rdl(__double2loint(pw), q - chunk0));
        acc += prob_of(code) * w;
      }
    }
    return acc;
  };
  double rowsum = 0.0;
  double* const sOut = lds + PEDN_TF_INL_ROWS * 64;   // INL: [PEDN_MAX_DEGREE - 1][64] the row's fractions on their way back to node_kernel
  if (!INL && coop) {  // the turns shared out over the waves, their sums handed to wave 0 through LDS
    for (int jj = part; jj < m - 1; jj += 4) sAcc[jj * 64 + lane] = turn_sum(jj);
    __syncthreads();
    if (part != 0) return;
    for (int jj = 0; jj < m - 1; ++jj) {
      const double acc = sAcc[jj * 64 + lane];
      out[(size_t)(turn0 + jj) * RS + r] = acc;
      rowsum = (jj == 0) ? acc : rowsum + acc;
    }
  } else if (!INL) {
    for (int jj = 0; jj < m - 1; ++jj) {
      const double acc = turn_sum(jj);
      out[(size_t)(turn0 + jj) * RS + r] = acc;
      rowsum = (jj == 0) ? acc : rowsum + acc;
    }
  } else {
    for (int jj = 0; jj < m - 1; ++jj) {
      const double acc = turn_sum(jj);
      sOut[jj * 64 + lane] = acc;
      rowsum = (jj == 0) ? acc : rowsum + acc;
    }
  }
  TPH(3, rowsum);
#ifdef PEDN_PHASE_PROFILE
  if (block % rgroups == 0 && lane == 0 && row < 4096) {
    for (int i = 0; i < 3; ++i) g_tphase[row * 8 + i] += tph[i + 1] - tph[i];
    g_tphase[row * 8 + 4] += 1;
    g_tphase[row * 8 + 5] = n_grp; g_tphase[row * 8 + 6] = nq; g_tphase[row * 8 + 7] = tph[0];
  }
#endif
  if (!INL) {
    if (fabs(rowsum - 1) > 1e-3) {  // check_fractions, :700-714
      for (int jj = 0; jj < m - 1; ++jj) {
        const double f = out[(size_t)(turn0 + jj) * RS + r];
        out[(size_t)(turn0 + jj) * RS + r] = rowsum > 1e-6 ? f / rowsum : 1.0 / (double)(m - 1);
      }
    }
  } else {  // the same on the wave's LDS rows, where node_kernel picks the fractions up; stored for the readers of tfd too
    const bool fix = fabs(rowsum - 1) > 1e-3;
    for (int jj = 0; jj < m - 1; ++jj) {
      const double a = sOut[jj * 64 + lane];
      const double f = fix ? (rowsum > 1e-6 ? a / rowsum : 1.0 / (double)(m - 1)) : a;
      out[(size_t)(turn0 + jj) * RS + r] = f;
      sOut[jj * 64 + lane] = f;
    }
  }
  if (fl) atomicOr(&v.flags[r], fl);
}

// stand-alone launch: first step of an episode, or a step that does not follow the previous one
template <bool PR, bool HIST>
__global__ __launch_bounds__(256) void turn_frac_kernel(DevView v, int t) {
  __shared__ double lds[PEDN_TF_LDS_DOUBLES];
  turn_frac_body<PR, false, HIST>(v, t, blockIdx.x, lds);
}

// RegularNode.solve('optimal') (node.py:249-271) for 64 replicas of one node, one LP per lane: dense primal simplex on the
// full tableau with Bland's rule, the operations of oracle/pedn_oracle.c: pedn_oracle_lp in the same order (see there for
// the programme).  The tableau of lane `lane` lives in HBM, element (k, c) at T[(k * W + c) * 64 + lane]; `tile` holds the
// node's turning fractions on entry ([i * m + j][64]) and floor(flow i -> j) on return; s / r: [slot][64] in LDS.
// Not a fast path: no scenario of the reference selects this node model.
__device__ __noinline__ bool lp_solve(double* T, int32_t* B, int m, const double* s, const double* r, double* tile, int lane) {
  const int E = m * (m - 1), R = 2 * m + E, N = 3 * E + 2 * m, W = N + 1;
  const double tol = 1e-9, w = PEDN_LP_PENALTY;
  auto el = [&](int k, int c) -> double& { return T[((size_t)k * W + c) * 64 + lane]; };
  for (int x = 0; x < (R + 1) * W; ++x) T[(size_t)x * 64 + lane] = 0.0;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      if (i == j) continue;
      const int e = i * (m - 1) + (j < i ? j : j - 1);
      el(i, e) = 1.0;
      el(m + j, e) = 1.0;
    }
  for (int k = 0; k < 2 * m; ++k) {
    el(k, 3 * E + k) = 1.0;
    el(k, N) = k < m ? s[k * 64 + lane] : r[(k - m) * 64 + lane];
    B[k * 64 + lane] = 3 * E + k;
  }
  for (int e = 0; e < E; ++e) {
    const int i = e / (m - 1), jj = e % (m - 1), k = 2 * m + e;
    const double tfe = tile[(i * m + (jj < i ? jj : jj + 1)) * 64 + lane];
    for (int c = i * (m - 1); c < (i + 1) * (m - 1); ++c) el(k, c) = tfe;
    el(k, e) = tfe - 1;
    el(k, E + 2 * e) = 1.0;
    el(k, E + 2 * e + 1) = -1.0;
    B[k * 64 + lane] = E + 2 * e;
  }
  for (int c = 0; c < N; ++c) el(R, c) = c < E ? -1.0 : (c < 3 * E ? w : 0.0);
  for (int k = 2 * m; k < R; ++k)
    for (int c = 0; c <= N; ++c) el(R, c) = el(R, c) - w * el(k, c);
  bool done = false, ok = true;
  for (int it = 0; it < 40 * (R + N); ++it) {
    int j = -1;
    if (!done)
      for (int c = 0; c < N; ++c)
        if (j < 0 && el(R, c) < -tol) j = c;
    if (j < 0) done = true;
    if (!__any(!done)) break;
    if (!done) {
      int i = -1, bi = 0;
      double best = 0.0;
      for (int k = 0; k < R; ++k) {
        const double a = el(k, j);
        if (a > tol) {
          const double ratio = el(k, N) / a;
          const int bk = B[k * 64 + lane];
          if (i < 0 || ratio < best - 1e-12 || (fabs(ratio - best) <= 1e-12 && bk < bi)) { best = ratio; i = k; bi = bk; }
        }
      }
      if (i < 0) { done = true; ok = false; }
      else {
        const double piv = el(i, j);
        for (int c = 0; c <= N; ++c) el(i, c) = el(i, c) / piv;
        for (int k = 0; k <= R; ++k) {
          if (k == i) continue;
          const double f = el(k, j);
          if (f == 0.0) continue;
          for (int c = 0; c <= N; ++c) el(k, c) = el(k, c) - f * el(i, c);
        }
        B[i * 64 + lane] = j;
      }
    }
  }
  if (!done) ok = false;
  for (int x = 0; x < m * m; ++x) tile[x * 64 + lane] = 0.0;
  if (ok)
    for (int k = 0; k < R; ++k) {
      const int e = B[k * 64 + lane];
      if (e < E) {
        const int i = e / (m - 1), jj = e % (m - 1);
        tile[(i * m + (jj < i ? jj : jj + 1)) * 64 + lane] = floor(el(k, N));
      }
    }
  return ok;
}

// One block = 8 waves = a bin of nodes whose slot counts add up to <= 8; one wave per (node slot, 64 replicas).
// Register budget: node_kernel_waves() below.  At 8 waves per SIMD: 59..63 VGPRs, no vector spill, scalar registers spilled into VGPR
// lanes, 4 blocks per CU.  tests/test_kernel_resources.py guards the budget (no scratch, no vector spill): one more live register in
// the wrong place turns the scalar spills into 16 vector spills (+11 us).
// LP: the node model is the linear programme of assign_flows_type 'optimal' instead of the classic proportional rule.
// MD: degree the row / column loops and the row of turning fractions are unrolled for (the host picks 6 when no node of the
// model has more incident corridors: 4 vector registers less in a kernel that lives on its last one)
// LU (link update by the owner wave): every physical link is the incoming link of exactly one slot, and everything the link
// update of step t-1 needs for that link -- inflow / outflow[t-1] of both directions of the corridor, num_pedestrians[t-2], the
// running sum, travel_time[t-1-W] -- was written by EARLIER launches.  So the slot wave of node_kernel(t) performs
// Network.update_link_states(t-1) (network.py:257-264) for its incoming link itself: it derives num_pedestrians[t-1] of both
// directions and avg_travel_time[t-1] of the incoming link in registers (the values it would otherwise load), stores the
// incoming link's rows, and carries on.  The gate record of a link (link.py:188) is written by the wave that holds the link as
// its OUTGOING link: that wave loads the back gate anyway, and reads it before it applies an RL action to it.  pedn_run then needs
// ONE launch per step (+ one trailing link_kernel for the last step of the range); see launch_step.
// TF (with LU; the single-launch plan of small batches with dynamic turning fractions): a slot wave whose row of fractions is computed
// on the device computes it ITSELF (turn_frac_body<.., INL>: from num_pedestrians[t-2] and the flows of t-1, the arithmetic the second
// launch of step t-1 would have used), right behind its batch of loads -- no launch in front of or behind node_kernel.
// node_step: one step of one workgroup (bx = replica group, by = bin of nodes).
// HELP (with TF; node_kernel_h, workgroups of SIXTEEN waves): waves 8..15 are helpers -- helper 8 + k computes the row of fractions of
// slot wave k (where that wave would compute its own) while the slot wave does its loads, link update and flows; they meet once, right
// before the slot wave multiplies its row into the sending flow.  A step of a small batch is as long as its slowest wave's chain: the
// row (~5 us) and the rest (~5 us) side by side instead of in a row.
// CLK: t and vhi were loaded from the device clock by the caller (node_kernel<.., CLK>); the horizon guard on t sits BEHIND the slot
// record's first load (below), so that the clock's scalar load and the record's load are in flight together -- a guard at the top of
// the kernel put the clock in front of the record as one more serial memory latency of every wave.
template <bool PR, bool LP, bool HIST, int MD, bool LU, bool TF, bool HELP = false, bool CLK = false>
__device__ __forceinline__ void node_step(const DevView& v, const int t, const int vhi, const int bx, const int by, double* const pedn_lds) {
  double* const sR = pedn_lds;                          // [8][64] receiving flow of each wave's outgoing link
  double* const sS = pedn_lds + 8 * 64;                 // [8][64] LP only: sending flow of each wave's incoming link
  double* const sPS = pedn_lds + (LP ? 16 : 8) * 64;    // per node m*m tiles of 64 lanes: P[i][j]*s_i, then floor(g_ij)
  const int wave16 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = HELP ? (wave16 & 7) : wave16;
  const int lane = (int)(threadIdx.x & 63);
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  const int r0 = v.sub0 + bx * 64;
  const int r = r0 + lane;
  const int tp = t - 1;
#ifdef PEDN_PHASE_PROFILE
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  PH(0, lane);
#endif
  const SlotRec& W = v.slot_rec[(size_t)by * 8 + wave];  // wave-uniform: scalar loads
  const int node = W.node, slot = W.slot, base = W.base, m = W.m;
  const bool active = node >= 0;
  PH(1, lane + node);
  if (CLK) {
    // a clocked launch replayed beyond the horizon: uniform over the grid.  (node >= -1 always: the term only makes the guard depend on
    // the record, or the compiler sinks the record's load behind the branch and the two loads are serial again; an asm barrier instead
    // costs the per-replica-parameter instantiations 6-16 vector spills)
    if (t >= v.T1 + (node < -1 ? 1 : 0)) return;
    if (bx == 0 && by == 0 && threadIdx.x == 0) v.clock[1] = t;   // the step of the launch behind this one (see node_clock)
  }
  uint32_t fl = 0;
  double s_i = 0.0, r_i = 0.0, qo = 0.0, qi = 0.0, co_prev = 0.0, ci_prev = 0.0;
  int lin = 0, lout = 0, kind = 0;
  double tfr[MD - 1];

  if (HELP && wave16 >= 8) {   // wave-uniform
    if (active && W.kind == 1 && W.dyn == 1 && W.trow >= 0) {
      const int* rw = v.trow_words + (size_t)W.trow * PEDN_TROW_WORDS;
      const int tw0 = rw[lane], tw1 = rw[64 + lane];
      turn_frac_body<PR, true, HIST, true>(v, t, 0u, pedn_lds + v.tf_lds_off + wave * PEDN_TF_INL_LDS, W.trow, r0, tw0, tw1);
    }
    __syncthreads();   // the meeting with the slot waves (below); the helpers' work is done
    // ... but they stay for the workgroup's two other barriers, so that EVERY wave of the workgroup passes the same three (a barrier
    // that some waves skip by retiring works on this hardware -- s_barrier counts the waves that are still alive -- and is outside
    // the HIP barrier model)
    __syncthreads();
    __syncthreads();
    return;
  }

  if (active) {
    kind = W.kind;
    lin = W.lin;
    lout = W.lout;
    const int turn0 = W.turn0 + slot * (m - 1);
    // turning fractions of row `slot` (SlotRec.dyn): imposed / default ones from tf or the replica-uniform shortcut tf_u; those
    // of a dynamic node (path_finder.py:591-715) from the buffer turn_frac_kernel filled for this step, or -- every product of
    // the row constant -- from the host's per-step table.  Part of the load batch below.
    const double* tfrow = W.dyn == 1 ? v.tfd[t & 1] : (W.dyn == 2 ? v.turn_tab_r : v.tf);
    const double* tfu = W.dyn == 2 ? v.turn_tab + (size_t)t * v.n_turns : v.tf_u;
    const bool tf_shared = W.dyn == 2 ? v.pod_pr == 0 : (W.dyn == 0 && tfu[turn0] == tfu[turn0]);  // tf_u: NaN = per-replica rows
    // TF: the wave's own row of fractions (ONE call site: the function is large), handed over in the wave's LDS rows
    double* const my_tf = pedn_lds + (TF ? v.tf_lds_off + wave * PEDN_TF_INL_LDS : 0);
    const bool inl_row = TF && kind == 1 && W.dyn == 1 && W.trow >= 0;
    // ---- the physical slot's parameters and its ONE batch of independent loads (see SlotIn): straight-line and unconditional so that
    // the compiler issues them back to back (a load skipped by a branch costs a wait at the join); the few values of an `early` step are
    // unused.  TF: issued BEFORE the wave computes its own row of fractions, so that the row's loads and arithmetic run under them.
    LinkP Pin = W.Pin, Pout = W.Pout;
    bool early = false;
    int tm1 = 0, t_sw = 0;
    SlotIn x;
    double lu_ia = 0.0, lu_oa = 0.0, lu_ib = 0.0, lu_ob = 0.0, lu_npa = 0.0, lu_npb = 0.0;
    float lu_pa = 0.0f, lu_pb = 0.0f, lu_rs = 0.0f, lu_old = 0.0f;
    const bool lu_win = tp >= v.W;
    auto load_batch = [&]() {
      Pin = lane_params<PR>(v, W.Pin, lin, r);
      Pout = lane_params<PR>(v, W.Pout, lout, r);
      early = tp < Pin.fft;  // link.py:267-269: sending flow is 0 until the first pedestrians can arrive
      uint32_t flw = 0;
      tm1 = wrap_idx(tp - 1, v.T1, flw);
      fl |= flw;
      t_sw = tp + 1 - Pout.tau_sw > 0 ? tp + 1 - Pout.tau_sw : 0;
      if (!LU) {
        x.n_in = rowp(v.f32[G_N], R32(G_N, tp), lin, L, RS, r0)[lane];
        x.n_out = rowp(v.f32[G_N], R32(G_N, tp), lout, L, RS, r0)[lane];
        // density[t'] of a plain link is num_pedestrians[t'] / float32(length * width) (link.py:136): recomputed from n_in with the
        // link update's own division instead of being read back; a separator's density depends on its width at that time
        x.k_in = Pin.sep ? rowp(v.f32[G_K], R32(G_K, tp), lin, L, RS, r0)[lane] : 0.0f;
        x.att_in = rowp(v.f32[G_ATT], R32(G_ATT, tp), lin, L, RS, r0)[lane];
      } else {
        // inputs of the link update of step t' for the corridor (lin, lout): all of them written by the launches before this one
        lu_ia = rowp(v.f64[F_IN], R64(F_IN, tp), lin, Lall, RS, r0)[lane];
        lu_oa = rowp(v.f64[F_OUT], R64(F_OUT, tp), lin, Lall, RS, r0)[lane];
        lu_ib = rowp(v.f64[F_IN], R64(F_IN, tp), lout, Lall, RS, r0)[lane];
        lu_ob = rowp(v.f64[F_OUT], R64(F_OUT, tp), lout, Lall, RS, r0)[lane];
        lu_pa = rowp(v.f32[G_N], R32(G_N, tp - 1), lin, L, RS, r0)[lane];
        lu_pb = rowp(v.f32[G_N], R32(G_N, tp - 1), lout, L, RS, r0)[lane];
        lu_rs = v.rsum[(size_t)lin * RS + r];
        lu_old = lu_win ? rowp(v.f32[G_TT], R32(G_TT, tp - v.W), lin, L, RS, r0)[lane] : 0.0f;
        if (Pin.sep) lu_npa = v.sepnp[(size_t)lin * RS + r];
        if (Pout.sep) lu_npb = v.sepnp[(size_t)lout * RS + r];
      }
      x.co_in = rowp(v.f64[F_CO], R64(F_CO, tp), lin, Lall, RS, r0)[lane];
      x.s_prev = rowp(v.f64[F_S], R64(F_S, tm1), lin, L, RS, r0)[lane];
      x.co_sw = rowp(v.f64[F_CO], R64(F_CO, t_sw), lout, Lall, RS, r0)[lane];
      x.ci_out = rowp(v.f64[F_CI], R64(F_CI, tp), lout, Lall, RS, r0)[lane];
      x.r_prev = rowp(v.f64[F_R], R64(F_R, tm1), lout, L, RS, r0)[lane];
      const double fu = v.front_u[lin], bu = v.back_u[lout];
      x.front_in = fu == fu ? fu : v.front[(size_t)lin * RS + r];
      x.back_out = bu == bu ? bu : v.back[(size_t)lout * RS + r];
      x.sepw_in = Pin.sep ? v.sepw[(size_t)lin * RS + r] : 0.0;
      x.sepw_out = Pout.sep ? v.sepw[(size_t)lout * RS + r] : 0.0;
    };
    int tw0 = 0, tw1 = 0;
    if (inl_row && !HELP) {   // the row's record: one vector load per half, in flight with the batch below
      const int* rw = v.trow_words + (size_t)W.trow * PEDN_TROW_WORDS;
      tw0 = rw[lane]; tw1 = rw[64 + lane];
    }
    if (TF && lin < L) load_batch();
    if (inl_row && !HELP) turn_frac_body<PR, true, HIST, true>(v, t, 0u, my_tf, W.trow, r0, tw0, tw1);
    if (lin >= L) {  // virtual pair: origin demand in, unlimited sink out (node.py:176,186)
      s_i = v.demand[((size_t)W.demand_row * v.T1 + tp) * RS + r];
      co_prev = rowp(v.f64[F_CO], R64(F_CO, tp), lin, Lall, RS, r0)[lane];
      ci_prev = rowp(v.f64[F_CI], R64(F_CI, tp), lout, Lall, RS, r0)[lane];
      if (kind == 1) {
        if (inl_row) {
#pragma unroll
          for (int jj = 0; jj < MD - 1; ++jj)
            if (!HELP && jj < m - 1) tfr[jj] = my_tf[(PEDN_TF_INL_ROWS + jj) * 64 + lane];
        } else {
#pragma unroll
          for (int jj = 0; jj < MD - 1; ++jj)
            if (jj < m - 1) tfr[jj] = tf_shared ? tfu[turn0 + jj] : tfrow[(size_t)(turn0 + jj) * RS + r];
        }
      }
      r_i = 1e6;
    } else {
      if (!TF) load_batch();
      x.co_sw = t_sw > vhi ? 0.0 : x.co_sw;   // as for ci_look below
      const double lu_gate = x.back_out;
      if (v.rl_actions != nullptr && W.act >= 0 && r < v.R) {
        // ActionApplier for a gater (rl/builders.py:313-352: clip_gater_action_value + back_gate_width setter, link.py:121-126),
        // done by the one wave that consumes the width: back gate of its outgoing link = front gate of its incoming link
        double a = v.rl_actions[(size_t)r * v.rl_A + W.act];
        if (a == a) {  // NaN: this agent was given no action
          const double cur = x.back_out;
          if (fabs(a - cur) > v.rl_max_delta_gate) a = cur + clip_d(a - cur, -v.rl_max_delta_gate, v.rl_max_delta_gate);
          a = clip_d(a, 0.0, Pout.width);
          x.back_out = x.front_in = a;
          v.back[(size_t)lout * RS + r] = a;
          v.front[(size_t)lin * RS + r] = a;
        }
      }
      if (kind == 1) {
        if (inl_row) {
#pragma unroll
          for (int jj = 0; jj < MD - 1; ++jj)
            if (!HELP && jj < m - 1) tfr[jj] = my_tf[(PEDN_TF_INL_ROWS + jj) * 64 + lane];
        } else {
#pragma unroll
          for (int jj = 0; jj < MD - 1; ++jj)
            if (jj < m - 1) tfr[jj] = tf_shared ? tfu[turn0 + jj] : tfrow[(size_t)(turn0 + jj) * RS + r];
        }
      }
      if (LU) {
        // Network.update_link_states(t') for the incoming link (link.py:133-188); the speed noise needs none of the loads
        const double nz = speed_noise(v, Pin, lin, tp, r);
        const float na = (float)((double)lu_pa + (lu_ia - lu_oa)), nb = (float)((double)lu_pb + (lu_ib - lu_ob));  // link.py:133-135
        // density = num_pedestrians / float32(length * width) (link.py:136): a plain link's area is the record's area32 -- bit for bit the
        // same cast, and length / width need not stay in scalar registers; a separator divides by its current width, in binary64 when
        // that width is held as np.float64 (PEDN_W_SEP_NUMPY), see link_update_one
        float ka = na / Pin.area32, kb = nb / Pout.area32;
        if (Pin.sep) ka = lu_npa != 0.0 ? (float)((double)na / (Pin.length * x.sepw_in)) : na / (float)(Pin.length * x.sepw_in);
        if (Pout.sep) kb = lu_npb != 0.0 ? (float)((double)nb / (Pout.length * x.sepw_out)) : nb / (float)(Pout.length * x.sepw_out);
        const SpeedOut so = speed_calc(v, Pin, lin, tp, r, ka, kb, lu_rs, lu_old, nz);
        rowp(v.f32[G_N], R32(G_N, tp), lin, L, RS, r0)[lane] = na;
        rowp(v.f32[G_K], R32(G_K, tp), lin, L, RS, r0)[lane] = ka;
        rowp(v.f32[G_V], R32(G_V, tp), lin, L, RS, r0)[lane] = so.spd;
        rowp(v.f32[G_TT], R32(G_TT, tp), lin, L, RS, r0)[lane] = so.tt;
        rowp(v.f32[G_LF], R32(G_LF, tp), lin, L, RS, r0)[lane] = so.lf;
        if (lu_win) rowp(v.f32[G_ATT], R32(G_ATT, tp), lin, L, RS, r0)[lane] = so.att;
        v.rsum[(size_t)lin * RS + r] = so.rs;
        // recorded width of the OUTGOING link (link.py:188 / :451-452): its back gate as loaded above, before any action of this step
        const double go = Pout.sep ? x.sepw_out : lu_gate;
        if (go != Pout.width || v.hist) rowp(v.f64[F_GATE], R64(F_GATE, tp), lout, L, RS, r0)[lane] = go;
        x.n_in = na; x.n_out = nb; x.k_in = ka; x.att_in = so.att;
      }
      if (!LU && !Pin.sep) x.k_in = x.n_in / Pin.area32;
      co_prev = x.co_in;   // cumulative_outflow[t-1] of the incoming link, reused by update_links below
      ci_prev = x.ci_out;  // cumulative_inflow[t-1] of the outgoing link
      PH(2, x.n_in + x.k_in + x.att_in + (float)(x.co_in + x.s_prev + x.co_sw + x.ci_out + x.r_prev + x.front_in + x.back_out));
      // the sending flow's look-back cumulative_inflow[t' + 1 - tau] is issued here and the receiving side's binomial (a Philox call,
      // independent of it) is drawn while it is in flight
      int idx_s = tp + 1 - __float2int_rn(x.att_in / (float)v.dt);  // link.py:260,274
      if (idx_s < 0) idx_s = 0;
      if (idx_s > tp + 1) idx_s = tp + 1;   // a zero / negative / garbage avg_travel_time must not take the load past the rows written so far
      double ci_look = v.f64[F_CI][at(R64(F_CI, idx_s), lin, Lall, RS, r)];
      // (lazy reset: a zero look-back -- the PEDN_F_SAME_STEP case -- reads the row of THIS step, which an ordinary reset left at 0)
      ci_look = idx_s > vhi ? 0.0 : ci_look;
      const double rp = recv_reverse_peds(v, Pout, lout, tp, r, x, fl);
      s_i = early ? 0.0 : send_flow<HIST>(v, Pin, lin, tp, r, x, ci_look, vhi, fl);
      PH(3, s_i);
      rowp(v.f64[F_S], R64(F_S, tp), lin, L, RS, r0)[lane] = s_i;  // link.py:268,367
      if (s_i < 0.0) fl |= PEDN_F_NEG_FLOW;
      r_i = recv_flow(v, Pout, lout, tp, r, x, s_i, rp, fl);
      PH(4, r_i);
      rowp(v.f64[F_R], R64(F_R, tp), lout, L, RS, r0)[lane] = r_i;  // node.py:206
    }
    if (s_i < 0.0 || r_i < 0.0) fl |= PEDN_F_NEG_FLOW;

    if (HELP) {   // the helper wave's row is in this wave's LDS rows once both have been here
      __syncthreads();
      if (inl_row) {
#pragma unroll
        for (int jj = 0; jj < MD - 1; ++jj)
          if (jj < m - 1) tfr[jj] = my_tf[(PEDN_TF_INL_ROWS + jj) * 64 + lane];
      }
    }
    if (kind == 1) {
      // P[i][j] * s_i  (node.py:285)
#pragma unroll
      for (int jj = 0; jj < MD - 1; ++jj) {
        if (jj < m - 1) {
          const int j = jj < slot ? jj : jj + 1;
          sPS[(size_t)(base + slot * m + j) * 64 + lane] = LP ? tfr[jj] : tfr[jj] * s_i;
        }
      }
    } else {
      sPS[(size_t)(base + slot) * 64 + lane] = s_i;
    }
    sR[wave * 64 + lane] = r_i;
    if (LP) sS[wave * 64 + lane] = s_i;
  } else if (HELP) {
    __syncthreads();   // an idle slot wave meets the helpers too
  }
  PH(5, r_i);
  __syncthreads();
  PH(6, lane);

  if (LP) {
    if (active && kind == 1 && slot == 0) {  // one wave solves the node's programme for its 64 replicas
      const size_t w = (size_t)W.lp * (size_t)(RS / 64) + (size_t)(r0 >> 6);
      if (!lp_solve(v.lp_ws + w * v.lp_stride, v.lp_basis + w * v.lp_bstride, m, &sS[wave * 64], &sR[wave * 64], &sPS[(size_t)base * 64], lane))
        fl |= PEDN_F_LP;
    }
  } else if (active && kind == 1) {
    // column `slot`: D_j = sum_i P[i][j] s_i (i ascending), g_ij = floor(min(P s, r_j * (P s / D_j)))  (node.py:286-298)
    double D = 0.0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < MD; ++k) {
      if (k < m && k != slot) {
        double x = sPS[(size_t)(base + k * m + slot) * 64 + lane];
        D = first ? x : D + x;
        first = false;
      }
    }
    const double Ds = D != 0.0 ? D : 1e-5;
#pragma unroll
    for (int k = 0; k < MD; ++k) {
      if (k < m && k != slot) {
        double a = sPS[(size_t)(base + k * m + slot) * 64 + lane];
        double b = r_i * (a / Ds);
        double g = floor(b < a ? b : a);
        sPS[(size_t)(base + k * m + slot) * 64 + lane] = g;
        qi += g;
      }
    }
  }
  PH(7, qi);
  __syncthreads();
  PH(8, lane);

  if (active) {
    if (kind == 1) {
#pragma unroll
      for (int j = 0; j < MD; ++j)
        if (j < m && j != slot) qo += sPS[(size_t)(base + slot * m + j) * 64 + lane];
      if (LP) {  // q = A_ub @ floor(x): the column sums were not formed by a column pass
#pragma unroll
        for (int k = 0; k < MD; ++k)
          if (k < m && k != slot) qi += sPS[(size_t)(base + k * m + slot) * 64 + lane];
      }
      if (!(qo > 0.0)) qo = 0.0;  // np.maximum(0, flows), node.py:299
      if (!(qi > 0.0)) qi = 0.0;
    } else {  // OneToOneNode.solve (node.py:230-242), not floored
      const int other = 1 - slot;
      double s_o = sPS[(size_t)(base + other) * 64 + lane];
      double r_o = sR[(wave - slot + other) * 64 + lane];
      qo = s_i < r_o ? s_i : r_o;
      qi = s_o < r_i ? s_o : r_i;
      if (qo < 0.0 || qi < 0.0) fl |= PEDN_F_NEG_FLOW;
    }
    // Node.update_links (node.py:146-162; link.py:19-25)
    double* const p_out = rowp(v.f64[F_OUT], R64(F_OUT, t), lin, Lall, RS, r0) + lane;
    double* const p_in = rowp(v.f64[F_IN], R64(F_IN, t), lout, Lall, RS, r0) + lane;
    rowp(v.f64[F_CO], R64(F_CO, t), lin, Lall, RS, r0)[lane] = co_prev + qo;
    rowp(v.f64[F_CI], R64(F_CI, t), lout, Lall, RS, r0)[lane] = ci_prev + qi;
    if (fl) atomicOr(&v.flags[r], fl);
    *p_out = qo;
    *p_in = qi;
  }
#ifdef PEDN_PHASE_PROFILE
  PH(9, qo + qi);
  if (active && lane == 0) {
    for (int i = 1; i < 10; ++i) if (ph[i] == 0) ph[i] = ph[i - 1];
    const size_t w = (((size_t)by * (size_t)(v.subRS / 64) + bx) * 8 + wave) % PEDN_PHASE_WAVES;
    for (int i = 1; i < 10; ++i) g_phase[w * 12 + i] += ph[i] - ph[i - 1];
    g_phase[w * 12 + 10] += 1ull;
    g_phase[w * 12 + 11] += ph[9] - ph[0];
  }
#endif
}

// Register budget of node_kernel = waves per SIMD its allocation aims at: 8 with shared link parameters and at most 6 corridors per
// node (59..63 VGPRs, no vector spill); 6 with per-replica parameters (28 more live vector registers: 0 spills instead of 2..8), for
// a node of 7 or 8 corridors (the instantiation unrolled for 8: 4..8 vector spills at 8 waves, none at 6) and for the node LP (4..14
// vector spills at 8); 2 where the slot waves compute their own rows of turning fractions (one workgroup per CU).  The other budgets
// were built and measured slower on every model (profiles/EXPERIMENTS.md #15, #24, #44).
// CLK (the clocked env step, below): 6 -- the step index and valid_hi are values loaded from memory that stay live, and at 8 waves the
// allocator answers with 20 vector spills; the batched RL step never fills the wave places of 8 anyway (544 workgroups at 2048 envs).
template <bool PR, bool LP, int MD, bool TF, bool CLK = false>
constexpr int node_kernel_waves() {
#ifdef PEDN_PHASE_PROFILE
  return TF ? 2 : 6;   // the profiling build's stamps need registers of their own
#else
  return TF ? 2 : (LP || PR || MD > 6 || CLK) ? 6 : 8;
#endif
}

// The step index of a launch: the host's argument, or -- CLK, the device clock (pedn_rl_clock_begin / pedn_rl_step_clocked: an env step
// whose launches have CONSTANT arguments, so that a captured graph can be replayed) -- read from v.clock:
//   clock[0]  step the next node_kernel runs      clock[1]  step the launch behind it runs (written by node_kernel)
//   clock[2]  DevView.valid_hi for that node_kernel (lazy reset: rows above it hold the previous episode's values)
// node_kernel(t) hands t to its second launch; that launch's first workgroup advances clock[0] / clock[2] when every read of them is
// over (stream order: no launch reads a word that the launch running next to it writes).
// The clocked launches are instantiations of their own (CLK): a step index that is a kernel argument can be re-read from the argument
// segment wherever it is needed, one that was loaded from memory has to be held in scalar registers, and node_kernel has none to spare.
struct StepClock { int t, vhi; };
__device__ __forceinline__ StepClock node_clock(const DevView& v) {
  StepClock c;
  c.t = v.clock[0];   // wave-uniform: two scalar loads beside the slot record's (node_step<.., CLK> checks t and writes clock[1])
  c.vhi = v.clock[2];
  return c;
}

template <bool PR, bool LP, bool HIST, int MD = PEDN_MAX_DEGREE, bool LU = false, bool TF = false, bool CLK = false>
__global__ __launch_bounds__(512, (node_kernel_waves<PR, LP, MD, TF, CLK>())) void node_kernel(DevView v, int t) {
  // dynamic LDS, sized by the host for the fullest block (pedn_create: node_lds): a block of nodes of degree 3..4 needs 24 of
  // the 64 tiles a single degree-8 node would
  // blockIdx.x = replica group (fastest in dispatch order): blocks launched together touch neighbouring 512-byte chunks
  // of the same history rows
  extern __shared__ double pedn_lds[];
  if (CLK) {
    const StepClock c = node_clock(v);
    node_step<PR, LP, HIST, MD, LU, TF, false, true>(v, c.t, c.vhi, (int)blockIdx.x, (int)blockIdx.y, pedn_lds);
  } else {
    node_step<PR, LP, HIST, MD, LU, TF>(v, t, v.valid_hi, (int)blockIdx.x, (int)blockIdx.y, pedn_lds);
  }
}

// the single-launch plan with helper waves (node_step<.., HELP>): sixteen waves per workgroup, 128 vector registers each
template <bool PR, bool HIST, int MD>
__global__ __launch_bounds__(1024) void node_kernel_h(DevView v, int t) {
  extern __shared__ double pedn_lds[];
  node_step<PR, false, HIST, MD, true, true, true>(v, t, v.valid_hi, (int)blockIdx.x, (int)blockIdx.y, pedn_lds);
}

__device__ __forceinline__ double2 ld2(const double* p, size_t i) { return *reinterpret_cast<const double2*>(p + i); }
__device__ __forceinline__ float2 ld2(const float* p, size_t i) { return *reinterpret_cast<const float2*>(p + i); }
__device__ __forceinline__ void st2(double* p, size_t i, double a, double b) { *reinterpret_cast<double2*>(p + i) = make_double2(a, b); }
__device__ __forceinline__ void st2(float* p, size_t i, float a, float b) { *reinterpret_cast<float2*>(p + i) = make_float2(a, b); }

// Network.update_link_states (network.py:257-264).  One lane = both directions of one corridor for two adjacent replicas of a
// segment of 128 replicas: every history access is a 16-byte (f64) or 8-byte (f32) vector access, i.e. 1 KiB / 512 B contiguous
// per wave instruction.  Used inside link_turn_kernel, where the turning-fraction waves compete for the wave slots (half the
// waves of link_pr_body).  (Two segments per lane -- four replicas -- were measured slower, profiles/r03_link_kernel_variants.txt.)
template <bool HIST>
__device__ __forceinline__ void link_body(const DevView& v, int t, size_t gid) {
  constexpr int NS = 1;
  const int RS = v.RS, L = v.L, Lall = v.Lall, H = v.subRS / (2 * NS);  // lanes per corridor
  int p = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)H));
  const int lh = (int)(gid % (size_t)H);
  const int r0 = v.sub0 + (lh / 64) * (128 * NS) + (lh % 64) * 2;  // first replica of segment 0; segment s starts 128 s further
  if (p >= v.n_pairs_corr) return;
  const CorrRec& C = v.corr_rec[p];  // wave-uniform
  // the rows of both directions are addressed from p alone where the model allows it (DevView.pairs_adj): their loads are then in
  // flight while the record -- a dependent, cache-cold fetch at the start of a launch -- arrives
  const int a = v.pairs_adj ? 2 * p : C.a, b = v.pairs_adj ? 2 * p + 1 : C.b;
  const LinkP& Pa = C.Pa;
  const LinkP& Pb = C.Pb;
  const bool win = t >= v.W;
  const double bua = v.back_u[a], bub = v.back_u[b];
  // ---- loads of all segments
  double2 ina[NS], outa[NS], inb[NS], outb[NS], wa[NS], wb[NS], fa[NS], fb[NS], ga[NS], gb[NS];
  float2 pa[NS], pb[NS], rsa[NS], rsb[NS], oa[NS], ob[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = r0 + 128 * s;
    ina[s] = ld2(v.f64[F_IN], at(R64(F_IN, t), a, Lall, RS, r)); outa[s] = ld2(v.f64[F_OUT], at(R64(F_OUT, t), a, Lall, RS, r));
    inb[s] = ld2(v.f64[F_IN], at(R64(F_IN, t), b, Lall, RS, r)); outb[s] = ld2(v.f64[F_OUT], at(R64(F_OUT, t), b, Lall, RS, r));
    pa[s] = ld2(v.f32[G_N], at(R32(G_N, t - 1), a, L, RS, r)); pb[s] = ld2(v.f32[G_N], at(R32(G_N, t - 1), b, L, RS, r));
    rsa[s] = ld2(v.rsum, (size_t)a * RS + r); rsb[s] = ld2(v.rsum, (size_t)b * RS + r);
    oa[s] = win ? ld2(v.f32[G_TT], at(R32(G_TT, t - v.W), a, L, RS, r)) : make_float2(0.f, 0.f);
    ob[s] = win ? ld2(v.f32[G_TT], at(R32(G_TT, t - v.W), b, L, RS, r)) : make_float2(0.f, 0.f);
    wa[s] = make_double2(Pa.width, Pa.width); wb[s] = make_double2(Pb.width, Pb.width);
    fa[s] = make_double2(0, 0); fb[s] = make_double2(0, 0);
    if (Pa.sep) { wa[s] = ld2(v.sepw, (size_t)a * RS + r); fa[s] = ld2(v.sepnp, (size_t)a * RS + r); }
    if (Pb.sep) { wb[s] = ld2(v.sepw, (size_t)b * RS + r); fb[s] = ld2(v.sepnp, (size_t)b * RS + r); }
    ga[s] = Pa.sep ? wa[s] : make_double2(bua, bua);  // recorded width (link.py:188 / :451-452)
    gb[s] = Pb.sep ? wb[s] : make_double2(bub, bub);
    if (!Pa.sep && !(bua == bua)) ga[s] = ld2(v.back, (size_t)a * RS + r);
    if (!Pb.sep && !(bub == bub)) gb[s] = ld2(v.back, (size_t)b * RS + r);
  }
  // ---- the noise draws need none of the loads
  double nza[NS][2], nzb[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nza[s][j] = speed_noise(v, Pa, a, t, r0 + 128 * s + j);
      nzb[s][j] = speed_noise(v, Pb, b, t, r0 + 128 * s + j);
    }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = r0 + 128 * s;
    // ---- per replica arithmetic (link.py:133-136, then update_speeds)
    float na[2], nb[2], ka[2], kb[2];
    SpeedOut sa[2], sb[2];
    const double dina[2] = {ina[s].x - outa[s].x, ina[s].y - outa[s].y}, dinb[2] = {inb[s].x - outb[s].x, inb[s].y - outb[s].y};
    const float pav[2] = {pa[s].x, pa[s].y}, pbv[2] = {pb[s].x, pb[s].y};
    const double wav[2] = {wa[s].x, wa[s].y}, wbv[2] = {wb[s].x, wb[s].y}, fav[2] = {fa[s].x, fa[s].y}, fbv[2] = {fb[s].x, fb[s].y};
    const float rsav[2] = {rsa[s].x, rsa[s].y}, rsbv[2] = {rsb[s].x, rsb[s].y}, oav[2] = {oa[s].x, oa[s].y}, obv[2] = {ob[s].x, ob[s].y};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      na[j] = (float)((double)pav[j] + dina[j]);
      nb[j] = (float)((double)pbv[j] + dinb[j]);
      // a separator width held as np.float64 turns the division into binary64 (PEDN_W_SEP_NUMPY)
      ka[j] = (Pa.sep && fav[j] != 0.0) ? (float)((double)na[j] / (Pa.length * wav[j])) : na[j] / (float)(Pa.length * wav[j]);
      kb[j] = (Pb.sep && fbv[j] != 0.0) ? (float)((double)nb[j] / (Pb.length * wbv[j])) : nb[j] / (float)(Pb.length * wbv[j]);
      sa[j] = speed_calc(v, Pa, a, t, r + j, ka[j], kb[j], rsav[j], oav[j], nza[s][j]);
      sb[j] = speed_calc(v, Pb, b, t, r + j, kb[j], ka[j], rsbv[j], obv[j], nzb[s][j]);
    }
    // ---- stores: all at the end (issued as their values appear, between the two speed computations, the launch is 0.5 us slower)
    st2(v.f32[G_N], at(R32(G_N, t), a, L, RS, r), na[0], na[1]);
    st2(v.f32[G_N], at(R32(G_N, t), b, L, RS, r), nb[0], nb[1]);
    st2(v.f32[G_K], at(R32(G_K, t), a, L, RS, r), ka[0], ka[1]);
    st2(v.f32[G_K], at(R32(G_K, t), b, L, RS, r), kb[0], kb[1]);
    st2(v.f32[G_V], at(R32(G_V, t), a, L, RS, r), sa[0].spd, sa[1].spd);
    st2(v.f32[G_V], at(R32(G_V, t), b, L, RS, r), sb[0].spd, sb[1].spd);
    st2(v.f32[G_TT], at(R32(G_TT, t), a, L, RS, r), sa[0].tt, sa[1].tt);
    st2(v.f32[G_TT], at(R32(G_TT, t), b, L, RS, r), sb[0].tt, sb[1].tt);
    st2(v.f32[G_LF], at(R32(G_LF, t), a, L, RS, r), sa[0].lf, sa[1].lf);
    st2(v.f32[G_LF], at(R32(G_LF, t), b, L, RS, r), sb[0].lf, sb[1].lf);
    if (win) {
      st2(v.f32[G_ATT], at(R32(G_ATT, t), a, L, RS, r), sa[0].att, sa[1].att);
      st2(v.f32[G_ATT], at(R32(G_ATT, t), b, L, RS, r), sb[0].att, sb[1].att);
    }
    st2(v.rsum, (size_t)a * RS + r, sa[0].rs, sa[1].rs);
    st2(v.rsum, (size_t)b * RS + r, sb[0].rs, sb[1].rs);
    // the record is initialised to `width` (link.py:56), so an unchanged gate needs no store
    if (ga[s].x != Pa.width || ga[s].y != Pa.width || v.hist) st2(v.f64[F_GATE], at(R64(F_GATE, t), a, L, RS, r), ga[s].x, ga[s].y);
    if (gb[s].x != Pb.width || gb[s].y != Pb.width || v.hist) st2(v.f64[F_GATE], at(R64(F_GATE, t), b, L, RS, r), gb[s].x, gb[s].y);
  }
}

// The link update of both directions of one corridor for ONE replica, given inflow[t] - outflow[t] (da, db) and
// num_pedestrians[t-1] (pa, pb) of the two directions; everything else it needs it loads itself.  (Round 2 also called it from
// inside node_kernel -- the later of a corridor's two end waves updated the corridor, PEDN_FUSE_LINK=1.  Parity-green and never
// faster with write-through hand-off (profiles/r02_last_arriver.txt); with the hand-off the memory model asks for -- agent-scope
// release / acquire fences = L2 write-back and invalidate per wave -- 18 x slower (profiles/r03_last_arriver.txt).  Removed.)
template <bool HIST, bool EARLY_NOISE>
__device__ __forceinline__ void link_update_one(const DevView& v, const LinkP& Pa, const LinkP& Pb, int a, int b, int t, int r,
                                                double ina, double outa, double inb, double outb, float pa, float pb) {
  const int RS = v.RS, L = v.L;
  const bool win = t >= v.W;
  // EARLY_NOISE: the draws run under the caller's loads (inside link_turn_kernel: -0.4 us per launch; the stand-alone launch over
  // shared parameters is 0.3 us faster with the draws where the speed is computed -- profiles/r03_noise_under_loads.txt)
  double nza = 0.0, nzb = 0.0;
  if (EARLY_NOISE) { nza = speed_noise(v, Pa, a, t, r); nzb = speed_noise(v, Pb, b, t, r); }
  const float na = (float)((double)pa + (ina - outa)), nb = (float)((double)pb + (inb - outb));
  const double wa = Pa.sep ? v.sepw[(size_t)a * RS + r] : Pa.width, wb = Pb.sep ? v.sepw[(size_t)b * RS + r] : Pb.width;
  const float ka = (Pa.sep && v.sepnp[(size_t)a * RS + r] != 0.0) ? (float)((double)na / (Pa.length * wa)) : na / (float)(Pa.length * wa);
  const float kb = (Pb.sep && v.sepnp[(size_t)b * RS + r] != 0.0) ? (float)((double)nb / (Pb.length * wb)) : nb / (float)(Pb.length * wb);
  const float oa = win ? v.f32[G_TT][at(R32(G_TT, t - v.W), a, L, RS, r)] : 0.0f, ob = win ? v.f32[G_TT][at(R32(G_TT, t - v.W), b, L, RS, r)] : 0.0f;
  if (!EARLY_NOISE) nza = speed_noise(v, Pa, a, t, r);
  const SpeedOut sa = speed_calc(v, Pa, a, t, r, ka, kb, v.rsum[(size_t)a * RS + r], oa, nza);
  if (!EARLY_NOISE) nzb = speed_noise(v, Pb, b, t, r);
  const SpeedOut sb = speed_calc(v, Pb, b, t, r, kb, ka, v.rsum[(size_t)b * RS + r], ob, nzb);
  // recorded width (link.py:188 / :451-452): the separator width, or the back gate -- one scalar load when every replica shares it
  const double bua = v.back_u[a], bub = v.back_u[b];
  const double ga = Pa.sep ? wa : (bua == bua ? bua : v.back[(size_t)a * RS + r]), gb = Pb.sep ? wb : (bub == bub ? bub : v.back[(size_t)b * RS + r]);
  v.f32[G_N][at(R32(G_N, t), a, L, RS, r)] = na; v.f32[G_N][at(R32(G_N, t), b, L, RS, r)] = nb;
  v.f32[G_K][at(R32(G_K, t), a, L, RS, r)] = ka; v.f32[G_K][at(R32(G_K, t), b, L, RS, r)] = kb;
  v.f32[G_V][at(R32(G_V, t), a, L, RS, r)] = sa.spd; v.f32[G_V][at(R32(G_V, t), b, L, RS, r)] = sb.spd;
  v.f32[G_TT][at(R32(G_TT, t), a, L, RS, r)] = sa.tt; v.f32[G_TT][at(R32(G_TT, t), b, L, RS, r)] = sb.tt;
  v.f32[G_LF][at(R32(G_LF, t), a, L, RS, r)] = sa.lf; v.f32[G_LF][at(R32(G_LF, t), b, L, RS, r)] = sb.lf;
  if (win) { v.f32[G_ATT][at(R32(G_ATT, t), a, L, RS, r)] = sa.att; v.f32[G_ATT][at(R32(G_ATT, t), b, L, RS, r)] = sb.att; }
  v.rsum[(size_t)a * RS + r] = sa.rs; v.rsum[(size_t)b * RS + r] = sb.rs;
  if (ga != Pa.width || v.hist) v.f64[F_GATE][at(R64(F_GATE, t), a, L, RS, r)] = ga;
  if (gb != Pb.width || v.hist) v.f64[F_GATE][at(R64(F_GATE, t), b, L, RS, r)] = gb;
}

// Same update with ONE replica per lane: with per-replica link parameters (PR: they live in vector registers), and as the launch of
// its own over shared parameters (link_kernel_1r: twice the waves of link_body at ~45 instead of 69 vector registers, 8-byte accesses).
template <bool PR, bool HIST>
__device__ __forceinline__ void link_pr_body(const DevView& v, int t, size_t gid) {
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  int p = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)v.subRS));
  const int r = v.sub0 + (int)(gid % (size_t)v.subRS);
  if (p >= v.n_pairs_corr) return;
  const CorrRec& C = v.corr_rec[p];
  const int a = v.pairs_adj ? 2 * p : C.a, b = v.pairs_adj ? 2 * p + 1 : C.b;   // see link_body
  const LinkP Pa = lane_params<PR>(v, C.Pa, a, r), Pb = lane_params<PR>(v, C.Pb, b, r);
  link_update_one<HIST, PR>(v, Pa, Pb, a, b, t, r, v.f64[F_IN][at(R64(F_IN, t), a, Lall, RS, r)], v.f64[F_OUT][at(R64(F_OUT, t), a, Lall, RS, r)],
                        v.f64[F_IN][at(R64(F_IN, t), b, Lall, RS, r)], v.f64[F_OUT][at(R64(F_OUT, t), b, Lall, RS, r)],
                        v.f32[G_N][at(R32(G_N, t - 1), a, L, RS, r)], v.f32[G_N][at(R32(G_N, t - 1), b, L, RS, r)]);
}

template <bool PR, bool HIST>
__global__ __launch_bounds__(256) void link_kernel_1r(DevView v, int t) { link_pr_body<PR, HIST>(v, t, (size_t)blockIdx.x * blockDim.x + threadIdx.x); }

// ---- batched RL glue (rl/builders.py, rl/pz_pednet_env.py:548-581) --------------------------------------------------
struct RlView {
  const int32_t *agent_type, *agent_link_ptr, *agent_links, *agent_act_off, *agent_obs_off;
  const int32_t *slot_agent, *slot_idx;  // action slot -> (agent, position in the agent's link list)
  double* actions;                        // [R][A]
  float *obs, *rew;                       // [R][O], [R][n_agents]
  int32_t n_agents, A, O, obs_mode, normalize, reward_mode, fpl;
  double max_delta_sep, max_delta_gate, min_sep;
};


// ActionApplier (builders.py:281-352): one lane per (action slot, replica)
__global__ void rl_apply_kernel(DevView v, RlView q) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)v.RS));  // RS is a multiple of 128: uniform per wave
  const int r = (int)(gid % (size_t)v.RS);
  if (a >= q.A || r >= v.R) return;
  const int ag = q.slot_agent[a], i = q.slot_idx[a];
  const int l = q.agent_links[q.agent_link_ptr[ag] + i];
  const LinkP P = v.lp[l];
  double x = q.actions[(size_t)r * q.A + a];
  if (x != x) return;  // NaN: this agent was given no action
  if (q.agent_type[ag] == 0) {  // separator: clip_separator_action_value + Separator.separator_width setter (link.py:462-478)
    const double cur = v.sepw[(size_t)l * v.RS + r];
    if (fabs(x - cur) > q.max_delta_sep) x = cur + clip_d(x - cur, -q.max_delta_sep, q.max_delta_sep);
    x = clip_d(x, q.min_sep, P.width - q.min_sep);
    const double other = P.width - x;
    v.sepnp[(size_t)l * v.RS + r] = 1.0; v.sepnp[(size_t)P.rev * v.RS + r] = 1.0;  // np.clip returns np.float64
    v.sepw[(size_t)l * v.RS + r] = x; v.front[(size_t)l * v.RS + r] = x; v.back[(size_t)l * v.RS + r] = x;
    v.sepw[(size_t)P.rev * v.RS + r] = other; v.front[(size_t)P.rev * v.RS + r] = other; v.back[(size_t)P.rev * v.RS + r] = other;
  } else {  // gater: clip_gater_action_value + back_gate_width setter (link.py:121-126)
    const double cur = v.back[(size_t)l * v.RS + r];
    if (fabs(x - cur) > q.max_delta_gate) x = cur + clip_d(x - cur, -q.max_delta_gate, q.max_delta_gate);
    x = clip_d(x, 0.0, P.width);
    v.back[(size_t)l * v.RS + r] = x;
    v.front[(size_t)P.rev * v.RS + r] = x;
  }
}

// ObservationBuilder.build_observation (builders.py:68-238) and PedNetParallelEnv._compute_rewards (pz_pednet_env.py:548-581)
// for every env.  Block of 4 waves = (agent, 64 replicas); wave w takes the agent's links w and w + 4: their observation
// features, and travel-time / density terms into LDS; wave 0 then folds the reward in link order (float32, like the
// reference).  FUSED: the block runs inside link_turn_kernel next to the link update of the same step, which has not stored
// travel time, speed and density of step t yet: they are recomputed for the agent's links with the link update's own
// arithmetic (speed_calc with the same Philox key), so the parts of that launch stay independent.
// lds: 3 * PEDN_MAX_DEGREE * 64 floats of the workgroup's LDS
#define PEDN_OBS_LDS_FLOATS (3 * PEDN_MAX_DEGREE * 64)
template <bool FUSED, bool HIST>
__device__ __forceinline__ void rl_observe_body(const DevView& v, const RlView& q, int t, int accumulate, unsigned block, float* lds) {
  float (*const sT)[64] = reinterpret_cast<float (*)[64]>(lds);
  float (*const sD)[64] = reinterpret_cast<float (*)[64]>(lds + PEDN_MAX_DEGREE * 64);
  float (*const sKc)[64] = reinterpret_cast<float (*)[64]>(lds + 2 * PEDN_MAX_DEGREE * 64);
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const unsigned rgroups = (unsigned)(v.subRS / 64);
  const int ag = (int)(block / rgroups);
  const int r = v.sub0 + (int)(block % rgroups) * 64 + lane;
  if (ag >= q.n_agents) return;  // block-uniform
  const int la = q.agent_link_ptr[ag], n = q.agent_link_ptr[ag + 1] - la;
  const int type = q.agent_type[ag];
  const bool live = r < v.R;
  float* o = q.obs + (size_t)(live ? r : 0) * q.O + q.agent_obs_off[ag];
  if (type == 0) {  // separator agent, builders.py:87-117
    if (wv == 0 && live) {
      const int f = q.agent_links[la], b = q.agent_links[la + 1];
      float x[4] = {(float)v.f64[F_IN][at(R64(F_IN, t), f, Lall, RS, r)], (float)v.f64[F_OUT][at(R64(F_OUT, t), f, Lall, RS, r)],
                    (float)v.f64[F_IN][at(R64(F_IN, t), b, Lall, RS, r)], (float)v.f64[F_OUT][at(R64(F_OUT, t), b, Lall, RS, r)]};
      for (int k = 0; k < 4; ++k) {
        if (q.normalize && (q.obs_mode == 1 || q.obs_mode == 2)) x[k] = x[k] / 20.0f;  // builders.py:183-188
        o[k] = x[k];
      }
    }
  } else {  // gater agent, builders.py:119-177
#pragma unroll 1
    for (int w = wv; w < n; w += 4) {
      const int l = q.agent_links[la + w];
      const LinkP P = v.pr ? lane_params<true>(v, v.lp[l], l, r) : v.lp[l];
      const double in_ld = v.f64[F_IN][at(R64(F_IN, t), l, Lall, RS, r)], out_ld = v.f64[F_OUT][at(R64(F_OUT, t), l, Lall, RS, r)];
      const double in_rd = v.f64[F_IN][at(R64(F_IN, t), P.rev, Lall, RS, r)], out_rd = v.f64[F_OUT][at(R64(F_OUT, t), P.rev, Lall, RS, r)];
      const float in_l = (float)in_ld, out_l = (float)out_ld, in_r = (float)in_rd, out_r = (float)out_rd;
      float tt_l, tt_r, spd, dens;
      if (!FUSED) {
        tt_l = v.f32[G_TT][at(R32(G_TT, t), l, L, RS, r)];
        tt_r = v.f32[G_TT][at(R32(G_TT, t), P.rev, L, RS, r)];
        spd = q.obs_mode == 5 ? v.f32[G_V][at(R32(G_V, t), l, L, RS, r)] : 0.0f;
        dens = dens_at<HIST>(v, P, l, t, r);
      } else {  // link.py:133-136 + update_speeds, as in link_body
        const LinkP Pr = v.pr ? lane_params<true>(v, v.lp[P.rev], P.rev, r) : v.lp[P.rev];
        const float na = (float)((double)v.f32[G_N][at(R32(G_N, t - 1), l, L, RS, r)] + (in_ld - out_ld));
        const float nb = (float)((double)v.f32[G_N][at(R32(G_N, t - 1), P.rev, L, RS, r)] + (in_rd - out_rd));
        const double wa = P.sep ? v.sepw[(size_t)l * RS + r] : P.width, wb = Pr.sep ? v.sepw[(size_t)P.rev * RS + r] : Pr.width;
        const float ka = (P.sep && v.sepnp[(size_t)l * RS + r] != 0.0) ? (float)((double)na / (P.length * wa)) : na / (float)(P.length * wa);
        const float kb = (Pr.sep && v.sepnp[(size_t)P.rev * RS + r] != 0.0) ? (float)((double)nb / (Pr.length * wb)) : nb / (float)(Pr.length * wb);
        const SpeedOut sa = speed_calc(v, P, l, t, r, ka, kb, 0.0f, 0.0f, speed_noise(v, P, l, t, r)), sb = speed_calc(v, Pr, P.rev, t, r, kb, ka, 0.0f, 0.0f, speed_noise(v, Pr, P.rev, t, r));
        tt_l = sa.tt;
        tt_r = sb.tt;
        spd = sa.spd;
        dens = P.sep ? ka : (na + nb) / (float)(P.length * P.width);  // Link.get_density / Separator.get_density at t
      }
      const float gate = (float)v.back[(size_t)l * RS + r];
      if (live) {
        float* oi = o + w * q.fpl;
        switch (q.obs_mode) {
          case 1: oi[0] = in_l; oi[1] = out_r; oi[2] = gate; break;
          case 2: oi[0] = in_l; oi[1] = out_r; oi[2] = dens; oi[3] = gate; break;
          case 3: oi[0] = in_l; oi[1] = out_l; oi[2] = in_r; oi[3] = out_r; oi[4] = gate; break;
          case 4: oi[0] = dens / (float)P.kj; oi[1] = gate; break;
          default: oi[0] = in_l; oi[1] = out_l; oi[2] = in_r; oi[3] = out_r; oi[4] = spd; oi[5] = dens; oi[6] = gate;
        }
        if (q.normalize) {  // builders.py:204-238, applied literally
          if (q.obs_mode == 1 || q.obs_mode == 2) { oi[0] = oi[0] / 20.0f; oi[1] = oi[1] / 20.0f; }
          else if (q.obs_mode == 3) { oi[0] = oi[0] / 6.0f; oi[1] = oi[1] / 20.0f; oi[2] = oi[2] / 20.0f; }
        }
      }
      sT[w][lane] = tt_l + tt_r;  // T_ell + T_ell_reverse, float32 (pz_pednet_env.py:566)
      sD[w][lane] = dens;
      sKc[w][lane] = (float)P.kc;
    }
  }
  __syncthreads();
  if (wv == 0 && live) {
    float reward = 0.0f;
    bool rewarded = false;
    if (type == 1) {  // reward terms in link order, float32 throughout (pz_pednet_env.py:557-577)
      float lr = 0.0f, dens_sum = 0.0f;
      for (int i = 0; i < n; ++i) {
        const float d = sD[i][lane];
        lr = (i == 0) ? 0.0f - sT[i][lane] : lr - sT[i][lane];
        if (d > 4.0f) lr = lr - 10.0f * (d - sKc[i][lane]);
        dens_sum = (i == 0) ? d : dens_sum + d;
      }
      if (n > 1) {  // np.mean of float32: sequential float32 sum / n
        const float avg = dens_sum / (float)n;
        float dsum = 0.0f;
        for (int i = 0; i < n; ++i) { const float d = fabsf(sD[i][lane] - avg); dsum = (i == 0) ? d : dsum + d; }
        lr = lr - 10.0f * (dsum / (float)n);
      }
      reward = lr;
      rewarded = true;
    }
    if (q.reward_mode == 0 && ag != 0) rewarded = false;  // the reference returns after the first agent (:581)
    float* rw = q.rew + (size_t)r * q.n_agents + ag;
    const float add = rewarded ? reward : 0.0f;
    *rw = (accumulate && rewarded) ? *rw + add : (accumulate ? *rw : add);
  }
}

template <bool HIST>
__global__ __launch_bounds__(256) void rl_observe_kernel(DevView v, RlView q, int t, int accumulate) {
  __shared__ float lds[PEDN_OBS_LDS_FLOATS];
  rl_observe_body<false, HIST>(v, q, t, accumulate, blockIdx.x, lds);
}

// ONE launch after node_kernel(t) for everything that only reads what node_kernel(t) and earlier launches wrote, in dispatch order:
//   blocks [0, n_tp_heavy)                      turning fractions of step t+1, the rows with long chains of softmax groups
//   the next n_link_blocks blocks               the link update of step t
//   the next n_tp_blocks - n_tp_heavy blocks    turning fractions of step t+1, the short rows
//   the remaining blocks                        observations and rewards of step t for the batched RL env (q.n_agents > 0)
// The parts are independent (the second and fourth re-derive what the others are about to store), so they run side by side;
// as separate launches each of them cost 5-9 us, most of it the fixed cost of a launch.  Longest first: a row with eight softmax
// groups is ~11 us of dependent binary64 arithmetic, a link-update workgroup lasts 5-6 us, a row with one or two groups 4-5 us.  The
// launch is more workgroups than the machine holds at this kernel's register budget (4 per CU), so what is dispatched last starts
// when the first workgroups retire: with every turning-fraction workgroup in front, a third of the link update started only after
// the short rows had finished and ended long after the long rows (delft x 1024: 20.8 us for 12 us of critical path).
// (OBS = false: the instantiation ordinary stepping uses carries neither the LDS nor the registers of the third part)
template <bool PR, bool OBS, bool HIST, bool CLK = false>
__global__ __launch_bounds__(256, 4) void link_turn_kernel(DevView v, int t, unsigned n_link_blocks, unsigned n_tp_blocks, unsigned n_tp_heavy, RlView q,
                                                           int accumulate) {
  // one LDS buffer for whichever part this workgroup is (the observation part needs 6 KB of the turning fractions' 35.5 KB)
  __shared__ double lds[PEDN_TF_LDS_DOUBLES];
  static_assert(sizeof(double) * PEDN_TF_LDS_DOUBLES >= sizeof(float) * PEDN_OBS_LDS_FLOATS, "observation rows must fit");
  // role of this workgroup (one call site per part: each is inlined once)
  const unsigned b = blockIdx.x;
  if (CLK) {   // device clock (see node_clock): this launch runs the step node_kernel left in clock[1]
    t = v.clock[1];
    if (t >= v.T1) return;   // replayed beyond the horizon: node_kernel did nothing either, the clock stays
    if (b == 0 && threadIdx.x == 0) {   // ... and its first workgroup advances the clock for the next node_kernel
      const int vh = v.clock[2];
      v.clock[0] = t + 1;
      v.clock[2] = (vh != 0x7fffffff && t > vh) ? t : vh;   // launch_step's bookkeeping: rows <= t are written once this step's launches are
    }
  }
  const bool is_tp = b < n_tp_heavy || (b >= n_tp_heavy + n_link_blocks && b < n_tp_blocks + n_link_blocks);
#ifdef PEDN_PHASE_PROFILE
  if (threadIdx.x == 0 && b < PEDN_LT_BLOCKS) {
    g_lt_time[b * 4] = is_tp ? (b < n_tp_heavy ? 0 : 2) : (b < n_tp_heavy + n_link_blocks ? 1 : 3);
    g_lt_time[b * 4 + 1] = wall_clock64();
  }
#endif
  if (is_tp) {
    // (behind the last step of the horizon there is no step t + 1: the host launches no such workgroups, a clocked launch has them idle)
    if (t + 1 < v.T1) turn_frac_body<PR, true, HIST>(v, t + 1, b < n_tp_heavy ? b : b - n_link_blocks, lds);
  } else if (b < n_tp_heavy + n_link_blocks) {
    const size_t gid = (size_t)(b - n_tp_heavy) * blockDim.x + threadIdx.x;
    if (PR) link_pr_body<true, HIST>(v, t, gid);
    else link_body<HIST>(v, t, gid);
  } else if (OBS) {
    rl_observe_body<true, HIST>(v, q, t, accumulate, b - n_link_blocks - n_tp_blocks, reinterpret_cast<float*>(lds));
  }
#ifdef PEDN_PHASE_PROFILE
  if ((threadIdx.x & 63) == 0 && b < PEDN_LT_BLOCKS) atomicMax(&g_lt_time[b * 4 + 2], (unsigned long long)wall_clock64());
#endif
}

// ---- state initialisation / host <-> device helpers ---------------------------------------------------------
// rows [row0, row0 + n_rows) of the fields selected by `what` back to their initial values (bit 0: the seven f64 / six f32 link fields
// incl. the running sum when row 0 is among them; bit 1: only avg_travel_time; bit 2: only the gate record; bit 3: everything but the
// gate record and the rows of avg_travel_time below the window -- what a lazy reset restored already).  Rows a field does not have (a
// shorter ring) are skipped.
__global__ void init_state_kernel(DevView v, int row0, int n_rows, int what) {
  const int RS = v.RS, L = v.L;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)n_rows * L * RS;
  if (gid >= total) return;
  int r = (int)(gid % RS);
  int l = (int)((gid / RS) % L);
  int t = row0 + (int)(gid / ((size_t)RS * L));  // history row (full-record mode: the time index; recent-history mode: a ring slot)
  const size_t i = ((size_t)t * L + l) * RS + r;
  const LinkP P = v.pr ? lane_params<true>(v, v.lp[l], l, r) : v.lp[l];
  auto has64 = [&](int f) { return t <= v.m64[f]; };  // rows of a field = mask + 1 (a ring), or all of max_rows
  auto has32 = [&](int g) { return t <= v.m32[g]; };
  const bool rest = (what & 8) != 0, all = (what & 1) != 0 || rest;
  const bool sr = all || (what & 16) != 0;   // bit 4: only sending / receiving flow
  if (sr && has64(F_S)) v.f64[F_S][i] = -1.0;
  if (sr && has64(F_R)) v.f64[F_R][i] = -1.0;
  if (((all && !rest) || (what & 4)) && has64(F_GATE)) v.f64[F_GATE][i] = P.width;  // link.py:56
  if (all && has32(G_TT)) v.f32[G_TT][i] = t == 0 ? P.tt0 : 0.0f;
  // link.py:91: avg_travel_time[t] = travel_time[0] for t < W; the link update only writes it from t = W on, so the slots
  // of a ring (fewer rows than W) all start there
  if (((all && !(rest && t < v.W)) || (what & 2)) && has32(G_ATT)) v.f32[G_ATT][i] = (t < v.W || v.hist) ? P.tt0 : 0.0f;
  if (all && has32(G_N)) v.f32[G_N][i] = 0.0f;
  if (all && has32(G_K)) v.f32[G_K][i] = 0.0f;
  if (all && has32(G_V)) v.f32[G_V][i] = 0.0f;
  if (all && has32(G_LF)) v.f32[G_LF][i] = 0.0f;
  if (all && t == 0) v.rsum[(size_t)l * RS + r] = P.tt0;  // link.py:84
}

// rows above valid_hi (lazy reset: neither written nor cleared since the reset) are answered with the field's initial value `init`
template <typename T>
__global__ void gather_kernel(const T* src, T* dst, int t0, int nt, int c0, int nc, int r0, int nr, int cols, int RS, int mask, int valid_hi, T init) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nt * nc * nr;
  if (gid >= total) return;
  int r = (int)(gid % nr);
  int c = (int)((gid / nr) % nc);
  int t = (int)(gid / ((size_t)nr * nc));
  dst[gid] = t0 + t > valid_hi ? init : src[((size_t)((t0 + t) & mask) * cols + (c0 + c)) * RS + (r0 + r)];
}

// dst[(row0 + i) * RS + r] = src[i * src_stride + (per_replica ? r : 0)] for r in [r0, r1)
__global__ void scatter_rows_kernel(double* dst, const double* src, int n_rows, size_t row0, size_t row_stride, int RS, int r0,
                                    int r1, int per_replica, int src_stride) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int nr = r1 - r0;
  if (gid >= (size_t)n_rows * nr) return;
  int i = (int)(gid / nr);
  int r = r0 + (int)(gid % nr);
  dst[(row0 + (size_t)i * row_stride) * RS + r] = src[(size_t)i * src_stride + (per_replica ? r : 0)];
}

// demand[row][t][r] = src[r][t] for t < n (0 beyond, and for the padding replicas): pedn_set_demand_matrix
__global__ void demand_matrix_kernel(double* dst, const double* src, size_t row, int n, int T1, int R, int RS) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)T1 * RS) return;
  const int t = (int)(gid / RS), r = (int)(gid % RS);
  dst[(row * T1 + t) * RS + r] = (r < R && t < n) ? src[(size_t)r * n + t] : 0.0;
}

// demand[row][t][replicas[k]] = src[k][t] for t < n (0 beyond): pedn_set_demand_rows
__global__ void demand_rows_kernel(double* dst, const double* src, const int32_t* replicas, int n_rep, size_t row, int n, int T1, int RS) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)T1 * n_rep) return;
  const int t = (int)(gid / n_rep), k = (int)(gid % n_rep);
  dst[(row * T1 + t) * RS + replicas[k]] = t < n ? src[(size_t)k * n + t] : 0.0;
}

// pedn_draw_demand: one lane per (time index, replica) of one origin's demand row
__global__ void draw_demand_kernel(double* dst, size_t row, int T1, int R, int RS, uint32_t k0, uint32_t k1, uint32_t replica_offset,
                                   uint32_t node, const int32_t* pattern, const double* base, const double* peak,
                                   const int32_t* spike_start, const int32_t* spike_len, const double* spike_height) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)T1 * RS) return;
  const int t = (int)(gid / RS), r = (int)(gid % RS), T = T1 - 1;
  double val = 0.0;
  if (r < R) {
    const int pat = pattern[r];
    if (pat == 1) val = base[r];
    else if (t < T) {
      const double w = (double)T / 20.0, a = ((double)t - (double)T / 4.0), b = ((double)t - 3.0 * (double)T / 4.0);
      // (pedn_exp, not the device library's exp: the series is pinned bit for bit by a CPU restatement, oracle/rand_contract.py)
      const double lam = base[r] + peak[r] * pedn_exp(-(a * a) / (2.0 * w * w)) + peak[r] * pedn_exp(-(b * b) / (2.0 * w * w));
      // Poisson(lam) by inversion: sequential search from 0 with one 53-bit uniform (lam stays below ~100 here)
      uint32_t c[4] = {(uint32_t)t, node, 0x50u, replica_offset + (uint32_t)r};
      philox4x32_10(c, k0, k1);
      const double u = (double)((((uint64_t)c[0] << 32) | c[1]) >> 11) * 0x1p-53;
      double p = pedn_exp(-lam), cdf = p;
      int k = 0;
      while (u > cdf && k < 1000) {
        ++k;
        p *= lam / (double)k;
        cdf += p;
      }
      val = (double)k;
      if (pat == 2 && t >= spike_start[r] && t < spike_start[r] + spike_len[r]) val += spike_height[r];
    }
  }
  dst[(row * T1 + t) * RS + r] = val;
}

// ---- per-replica scenarios: packing, and the randomisers of env_loader.py:183-259,363-424 on the device ----------------------
// (travel_time[0], free_flow_tau, tau_shockwave) of a link from its parameters with the expressions of link.py:58-63,83-86,380 and
// the roundings of scenarios.py: derive_statics_arrays (np.float32 / unit_time is a float32 division; round = rint, half to even)
__device__ __forceinline__ LinkPR make_link_pr(double kc, double kj, double vf, double length, double dt) {
  LinkPR q;
  q.kc = kc; q.kj = kj; q.vf = vf;
  const double shockwave_speed = (vf * kc) / (kj - kc);
  const double a = length / vf, b = length / 0.05;
  q.tt0 = (float)(a < b ? a : b);
  const int fft = __float2int_rn(q.tt0 / (float)dt), tsw = (int)rint(length / (shockwave_speed * dt));
  q.fft = (int16_t)(fft > 32767 ? 32767 : fft);
  q.tau_sw = (int16_t)(tsw > 32767 ? 32767 : (tsw < 0 ? 0 : tsw));
  return q;
}

// pedn_set_link_params: six host matrices [L][R] -> records [L][RS]
__global__ void pack_link_params_kernel(LinkPR* dst, const double* kc, const double* kj, const double* vf, const int32_t* fft,
                                        const int32_t* tau_sw, const float* tt0, int L, int R, int RS) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)L * RS) return;
  const int l = (int)(gid / RS), r = (int)(gid % RS);
  LinkPR q;
  if (r < R) {
    const size_t i = (size_t)l * R + r;
    q.kc = kc[i]; q.kj = kj[i]; q.vf = vf[i]; q.tt0 = tt0[i]; q.fft = (int16_t)fft[i]; q.tau_sw = (int16_t)tau_sw[i];
  } else {  // padding lanes hold valid numbers and stay idle: free_flow_tau far in the future
    q.kc = 1.0; q.kj = 2.0; q.vf = 1.0; q.tt0 = 1.0f; q.fft = 32767; q.tau_sw = 1;
  }
  dst[gid] = q;
}
// pedn_get_link_params: records -> whichever of the six host-layout matrices [L][R] are asked for
__global__ void unpack_link_params_kernel(const LinkPR* src, double* kc, double* kj, double* vf, int32_t* fft, int32_t* tau_sw, float* tt0,
                                          int L, int R, int RS) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)L * R) return;
  const int l = (int)(gid / R), r = (int)(gid % R);
  const LinkPR q = src[(size_t)l * RS + r];
  if (kc) kc[gid] = q.kc;
  if (kj) kj[gid] = q.kj;
  if (vf) vf[gid] = q.vf;
  if (fft) fft[gid] = q.fft;
  if (tau_sw) tau_sw[gid] = q.tau_sw;
  if (tt0) tt0[gid] = q.tt0;
}

__device__ __forceinline__ double u01(uint32_t w) { return (double)w * 0x1p-32; }   // [0, 1)

// generate_random_link_params (env_loader.py:363-424) for every replica: one lane per replica walks the corridors and picks exactly
// k = int(P * fraction) of them without replacement by selection sampling (corridor p is taken with probability
// (k - taken) / (P - p): every k-subset is equally likely, like np.random.choice(..., replace=False)); a chosen corridor gets, each
// with probability 1/2, k_critical / k_jam scaled by U(0.6, 1.2) with the floors max(0.5, .) / max(2 k_c, .), and free_flow_speed
// scaled by U(0.6, 0.9).  Both links of the corridor are scaled from their OWN base parameters (v.lp).  One Philox call per
// (corridor, replica), keyed by the global replica id: the scenarios do not depend on how the ensemble is sharded.
__global__ void rand_links_kernel(DevView v, LinkPR* dst, int k, uint32_t k0, uint32_t k1, int* max_tau_sw) {
  const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= v.RS) return;
  const int P = v.n_pairs_corr;
  int taken = 0, tmax = 0;
  for (int p = 0; p < P; ++p) {
    uint32_t c[4] = {(uint32_t)p, 0x61u, 0u, v.replica_offset + (uint32_t)r};
    philox4x32_10(c, k0, k1);
    const bool chosen = (uint64_t)c[0] * (uint64_t)(P - p) < ((uint64_t)(k - taken) << 32);
    taken += chosen ? 1 : 0;
    const bool dens = chosen && (c[1] & 0xffffu) < 0x8000u, spd = chosen && (c[1] >> 16) < 0x8000u;
    const double F = 0.6 + (1.2 - 0.6) * u01(c[2]), G = 0.6 + (0.9 - 0.6) * u01(c[3]);
    const int ab[2] = {v.pairs_adj ? 2 * p : v.corr_rec[p].a, v.pairs_adj ? 2 * p + 1 : v.corr_rec[p].b};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const LinkP& B = v.lp[ab[h]];
      double kc = B.kc, kj = B.kj, vf = B.vf;
      if (dens) {
        kc = fmax(0.5, B.kc * F);
        kj = fmax(kc * 2.0, B.kj * F);
      }
      if (spd) vf = B.vf * G;
      const LinkPR q = make_link_pr(kc, kj, vf, B.length, v.dt);
      tmax = max(tmax, (int)q.tau_sw);
      dst[(size_t)ab[h] * v.RS + r] = q;
    }
  }
  if (max_tau_sw && r < v.R) atomicMax(max_tau_sw, tmax);
}

// generate_random_od_flows (env_loader.py:224-259): U(1, 10) per (OD pair, replica), constant over the episode
__global__ void rand_od_kernel(double* od_w_r, int n_od, int RS, uint32_t k0, uint32_t k1, uint32_t replica_offset) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n_od * RS) return;
  const int od = (int)(gid / RS), r = (int)(gid % RS);
  uint32_t c[4] = {(uint32_t)od, 0x63u, 0u, replica_offset + (uint32_t)r};
  philox4x32_10(c, k0, k1);
  od_w_r[gid] = 1.0 + (10.0 - 1.0) * u01(c[0]);
}

// P(od | up) per replica from time-constant OD weights (path_finder.py:599-615), the arithmetic of tabulate_pair_pod column by column:
//   pod_tot_kernel     tot[u][r] = sum of the upstream's OD weights in table order
//   pod_pair_kernel    pair_pod_r[q][r] = w / tot (uniform when the sum is 0)
//   pod_turn_kernel    turn_tab_r[turn][r] = sum of the products of a turn whose probabilities are all the constant 1 (h_turn_mode)
//   pod_finish_kernel  check_fractions (path_finder.py:691-715) of the rows tabulated this way: finish_tabulated_rows per replica
__global__ void pod_tot_kernel(const double* od_w_r, const int32_t* up_od_ptr, const int32_t* upod_od, int n_up, int RS, double* tot) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n_up * RS) return;
  const int u = (int)(gid / RS), r = (int)(gid % RS);
  double t = 0.0;
  for (int q = up_od_ptr[u]; q < up_od_ptr[u + 1]; ++q) t += od_w_r[(size_t)upod_od[q] * RS + r];
  tot[gid] = t;
}
__global__ void pod_pair_kernel(const double* od_w_r, const int32_t* up_od_ptr, const int32_t* upod_od, const int32_t* upod_up,
                                const int32_t* pair_upod, const double* tot, int n_pair, int RS, double* pair_pod_r) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n_pair * RS) return;
  const int q = (int)(gid / RS), r = (int)(gid % RS);
  const int x = pair_upod[q], u = upod_up[x], n = up_od_ptr[u + 1] - up_od_ptr[u];
  const double t = tot[(size_t)u * RS + r];
  pair_pod_r[gid] = t > 0.0 ? od_w_r[(size_t)upod_od[x] * RS + r] / t : (n > 0 ? 1.0 / (double)n : 0.0);
}
__global__ void pod_turn_kernel(const double* pair_pod_r, const int32_t* turn_pair_ptr, const int32_t* turn_mode, int n_turns, int RS,
                                double* turn_tab_r) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n_turns * RS) return;
  const int tn = (int)(gid / RS), r = (int)(gid % RS);
  double acc = 0.0;
  if (turn_mode[tn])
    for (int q = turn_pair_ptr[tn]; q < turn_pair_ptr[tn + 1]; ++q) acc += 1.0 * pair_pod_r[(size_t)q * RS + r];
  turn_tab_r[gid] = acc;
}
__global__ void pod_finish_kernel(double* turn_tab_r, const int32_t* tab_rows, int n_rows, int RS) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n_rows * RS) return;
  const int i = (int)(gid / RS), r = (int)(gid % RS);
  const int ta = tab_rows[2 * i], n = tab_rows[2 * i + 1];
  double rowsum = 0.0;
  for (int jj = 0; jj < n; ++jj) {
    const double f = turn_tab_r[(size_t)(ta + jj) * RS + r];
    rowsum = (jj == 0) ? f : rowsum + f;
  }
  if (fabs(rowsum - 1) > 1e-3)
    for (int jj = 0; jj < n; ++jj) {
      const double f = turn_tab_r[(size_t)(ta + jj) * RS + r];
      turn_tab_r[(size_t)(ta + jj) * RS + r] = rowsum > 1e-6 ? f / rowsum : 1.0 / (double)n;
    }
}

// generate_random_demand_params (env_loader.py:183-222) + the series of od_manager.py:92-155 for EVERY origin and replica in one
// launch: blockIdx.y = origin.  Pattern (uniform over gaussian_peaks / constant / sudden_demand), base U(2, 10), peak max(U(10, 30),
// base + 5), spike length in [10, 20), start in [0, max(1, T - length)), height in [20, 50) are drawn per (origin, replica); the Poisson
// series as in draw_demand_kernel.
__global__ void rand_demand_kernel(double* dst, const int32_t* rows, const int32_t* nodes, int T1, int R, int RS, uint32_t k0, uint32_t k1,
                                   uint32_t replica_offset) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)T1 * RS) return;
  const int t = (int)(gid / RS), r = (int)(gid % RS), T = T1 - 1;
  const uint32_t node = (uint32_t)nodes[blockIdx.y];
  const size_t row = (size_t)rows[blockIdx.y];
  double val = 0.0;
  if (r < R) {
    uint32_t a[4] = {node, 0x51u, 0u, replica_offset + (uint32_t)r}, b[4] = {node, 0x52u, 0u, replica_offset + (uint32_t)r};
    philox4x32_10(a, k0, k1);
    philox4x32_10(b, k0, k1);
    const int pat = (int)(((uint64_t)a[0] * 3u) >> 32);
    const double base = 2.0 + 8.0 * u01(a[1]);
    double peak = 10.0 + 20.0 * u01(a[2]);
    if (peak < base + 5.0) peak = base + 5.0;
    const int len = 10 + (int)(((uint64_t)a[3] * 10u) >> 32);
    const int span = T - len > 1 ? T - len : 1;
    const int start = (int)(((uint64_t)b[0] * (uint64_t)span) >> 32);
    const double height = (double)(20 + (int)(((uint64_t)b[1] * 30u) >> 32));
    if (pat == 1) val = base;
    else if (t < T) {
      const double w = (double)T / 20.0, x = ((double)t - (double)T / 4.0), y = ((double)t - 3.0 * (double)T / 4.0);
      const double lam = base + peak * pedn_exp(-(x * x) / (2.0 * w * w)) + peak * pedn_exp(-(y * y) / (2.0 * w * w));
      uint32_t c[4] = {(uint32_t)t, node, 0x50u, replica_offset + (uint32_t)r};
      philox4x32_10(c, k0, k1);
      const double u = (double)((((uint64_t)c[0] << 32) | c[1]) >> 11) * 0x1p-53;
      double p = pedn_exp(-lam), cdf = p;
      int k = 0;
      while (u > cdf && k < 1000) {
        ++k;
        p *= lam / (double)k;
        cdf += p;
      }
      val = (double)k;
      if (pat == 2 && t >= start && t < start + len) val += height;
    }
  }
  dst[(row * T1 + t) * RS + r] = val;
}

// pedn_set_width: ONE value into the replicas [r0, r0 + n) of one link's width row, and the link's replica-uniform shortcut entry beside it
// -- the values travel as kernel arguments: no staging copy, no host memory that has to outlive the call, nothing to wait for (a host-side
// controller sets a width after every step)
__global__ void set_width_kernel(double* row, int r0, int n, double value, double* uniform_entry, double uniform_value) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) row[r0 + i] = value;
  if (i == 0 && uniform_entry != nullptr) *uniform_entry = uniform_value;
}

// pedn_rl_clock_begin: the device clock (node_clock) set to step t
__global__ void set_clock_kernel(int32_t* clock, int t, int valid_hi) {
  if (threadIdx.x == 0) { clock[0] = t; clock[1] = t; clock[2] = valid_hi; clock[3] = 0; }
}
// busy for `ticks` of the constant 100 MHz clock (pedn_hip.hip: probe_overlap); every wave reaches the exit
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}

__global__ void device_math_kernel(int op, int n, const double* a, const double* b, uint32_t k0, uint32_t k1, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op == 8) {  // streaming calibration with 16-byte accesses: lane i < n/2 handles elements 2i, 2i+1 (n even)
    if (2 * i + 1 < n) {
      const double2 x2 = reinterpret_cast<const double2*>(a)[i], y2 = reinterpret_cast<const double2*>(b)[i];
      reinterpret_cast<double2*>(out)[i] = make_double2(x2.x + y2.x, x2.y + y2.y);
    }
    return;
  }
  if (op == 9) {  // scattered-chunk calibration: chunks of k0 doubles visited in a scrambled order (n, k0 powers of two)
    const uint32_t chunk = k0, nch = (uint32_t)n / chunk, c = (uint32_t)i / chunk;
    const uint32_t pc = (c * 2654435761u + 12345u) & (nch - 1);  // odd multiplier: a permutation of the chunk ids
    const size_t j = (size_t)pc * chunk + (uint32_t)i % chunk;
    out[j] = a[j] + b[j];
    return;
  }
  double x = a[i], y = b ? b[i] : 0.0;
  RngKey key{k0, k1, (uint32_t)i, 7u, 11u, 0u};
  switch (op) {
    case 0: out[i] = (double)pedn_powf((float)x, (float)y); break;
    case 1: out[i] = pedn_exp(x); break;
    case 2: out[i] = sqrt(x); break;
    case 3: out[i] = x / y; break;
    case 4: out[i] = (double)((float)x / (float)y); break;
    case 5: out[i] = rng_binomial((long long)x, y, key, 0); break;
    case 6: key.site = 3u; out[i] = x * rng_z(key); break;
    case 7: out[i] = x + y; break;  // streaming calibration: 16 B read + 8 B written per lane, 8-byte accesses
    default: out[i] = 0.0;
  }
}
