// pedn_hip.hip -- MI355X (gfx950) engine for PedNStream's network_loading hot path: HIP kernels + C-ABI (include/pedn.h).
//
// Data layout in HBM: every history field is one array [T+1][columns][RS] with the replica index
// fastest (RS = replicas rounded up to a multiple of 128).  A wavefront therefore owns 64 consecutive replicas of ONE
// link or node: topology and link parameters are wave-uniform (scalar loads), every history access of a wave is
// one coalesced 256/512-byte row segment, and the data-dependent look-backs (cumulative_inflow[t+1-tau],
// inflow[t-tau-k]) gather between rows of the same column.
//
// Kernels per step t (reference network.py:266-287):
//   turn_prob_kernel  one lane per (softmax group, replica): P(down | up, od)      path_finder.py:561-589
//   node_kernel       one wave per (node slot, 64 replicas), one block per bin of nodes with <= 8 slots in total:
//                     sending flow of the slot's incoming link, receiving flow of its outgoing link, dynamic turning
//                     fractions, the node's flow distribution through LDS, cumulative counts
//                                                                                     node.py:164-221,230-242,272-300; link.py:216-416
//   link_kernel       one lane per (corridor = link pair, replica): pedestrians, density, fundamental diagram,
//                     travel time and its moving average                            link.py:133-188; functions.py:112-134
//
// Compile with -ffp-contract=off: results must match the reference bit for bit (cumulative counts) and the
// arithmetic below spells out every binary32 / binary64 rounding point of numpy's scalar semantics.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/pedn.h"
#include "pedn_math.hpp"

using namespace pedn;

// ------------------------------------------------------------------------------------------------- device view
enum { F_IN = 0, F_OUT, F_CI, F_CO, F_S, F_R, F_GATE };
enum { G_TT = 0, G_ATT, G_N, G_K, G_V, G_LF };

struct LinkP {  // static per-link parameters, wave-uniform on the device
  double length, width, vf, kc, kj, gamma, act, bi, noise;
  float tt0;
  int32_t rev, sep, fd, tau_sw, fft;
  int32_t pad[2];
};

struct SlotRec {  // one wave of node_kernel: a (node, slot) with everything static it needs, fetched by ONE scalar load burst
  int32_t node, slot, base, m, kind, dyn, lin, lout, turn0, demand_row, pad0, pad1;
  LinkP Pin, Pout;  // parameters of the incoming / outgoing link of the slot (unused for a virtual pair)
};

struct CorrRec {  // one lane group of link_kernel: both directions of a corridor
  int32_t a, b, pad0, pad1;
  LinkP Pa, Pb;
};

struct EntS {  // one downstream entry of a softmax group (update_node_turn_probs, path_finder.py:561-589)
  int32_t link, rev, sep;  // outgoing link (-1: virtual), its reverse, separator flag
  float area32;            // float32(length * width) of a plain link
  double vf, kc;           // for the capacity fallback back_gate * v_f * k_c * dt (:576)
  double dist_term;        // alpha * distance / (sum of the group's distances + 1e-6), static (:582)
};

struct DevView {
  double* f64[7];
  float* f32[6];
  float* rsum;
  double *front, *back, *sepw, *sepnp, *tf, *demand, *ent_p;
  // replica-uniform shortcuts (NaN = the value differs between replicas, read the per-replica row instead): a value every
  // replica shares is one scalar load per wave instead of 8 bytes per lane
  const double *front_u, *back_u;  // [L]
  const double* tf_u;              // [n_turns]
  // per-replica scenario parameters (randomised ensembles / RL resets, env_loader.py:363-424); used when pr != 0
  const double *kc_r, *kj_r, *vf_r;   // [L][RS] k_critical, k_jam, free_flow_speed
  const int32_t *fft_r, *tausw_r;     // [L][RS] free_flow_tau, shock-wave look-back
  const float* tt0_r;                 // [L][RS] travel_time[0]
  const double *pair_pod_r, *turn_tab_r;  // [n_pair][RS], [n_turns][RS]: P(od | up) with per-replica, time-constant OD weights
  int32_t pr, pod_pr;
  const double* od_w;
  uint32_t* flags;
  const LinkP* lp;
  const SlotRec* slot_rec;
  const CorrRec* corr_rec;
  const int32_t *node_kind, *node_slot_ptr, *node_turn_ptr, *node_demand_row, *node_dyn, *slot_in, *slot_out;
  const int32_t *grp_ent_ptr, *grp_allphys, *ent_pair, *turn_pair_ptr;
  const struct EntS* ents;      // static per-entry data of the softmax groups
  const int32_t* grp_multi;     // groups with more than one downstream entry (single-entry groups have P = 1 exactly)
  const int32_t* pair_const;    // [n_pair] 1: the product's probability is the constant 1
  const int32_t* turn_mode;     // [n_turns] 1: every product of the turn is constant -> fraction tabulated per step on the host
  const double* turn_tab;       // [T+1][n_turns] tabulated raw fractions of such turns
  const double* pair_pod;  // [T+1][n_pair] P(od | up) of the pair's upstream group, replica independent
  int32_t L, Lall, T1, RS, R, W, n_grp, n_multi, n_pairs_corr, n_pair, n_turns;
  double dt, pf_temp, pf_alpha, pf_beta, pf_omega, pf_eps;
  uint32_t k0, k1, replica_offset;
  int32_t meanfield;
};

__device__ __forceinline__ size_t at(int t, int col, int cols, int RS, int r) {
  return ((size_t)t * (size_t)cols + (size_t)col) * (size_t)RS + (size_t)r;
}
__device__ __forceinline__ float clip01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
// Link parameters as seen by one lane: the shared record, or (PR) the replica's own k_critical / k_jam / free-flow speed
// and the quantities derived from them on the host.
template <bool PR>
__device__ __forceinline__ LinkP lane_params(const DevView& v, const LinkP& P, int l, int r) {
  LinkP Q = P;
  if (PR) {
    const size_t i = (size_t)l * v.RS + r;
    Q.kc = v.kc_r[i]; Q.kj = v.kj_r[i]; Q.vf = v.vf_r[i];
    Q.fft = v.fft_r[i]; Q.tau_sw = v.tausw_r[i]; Q.tt0 = v.tt0_r[i];
  }
  return Q;
}

__device__ __forceinline__ int wrap_idx(int i, int T1, uint32_t& fl) {
  if (i < 0) i += T1;
  if (i < 0 || i >= T1) { fl |= PEDN_F_INDEX; return 0; }
  return i;
}

// Link.get_density (link.py:190-197) / Separator.get_density (:427-428) at history index t
__device__ __forceinline__ float dens_at(const DevView& v, const LinkP& P, int l, int t, int r) {
  if (P.sep) return v.f32[G_K][at(t, l, v.L, v.RS, r)];
  float n = v.f32[G_N][at(t, l, v.L, v.RS, r)] + v.f32[G_N][at(t, P.rev, v.L, v.RS, r)];
  return n / (float)(P.length * P.width);
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
__device__ double send_flow(const DevView& v, const LinkP& P, int l, int tp, int r, const SlotIn& x, uint32_t& fl) {
  const int RS = v.RS, T1 = v.T1;
  const float nself = x.n_in, nrev = x.n_out, kk = x.k_in, att = x.att_in;
  double aw = P.sep ? x.sepw_in : P.width;
  float dens = P.sep ? kk : (nself + nrev) / (float)(P.length * aw);
  int tau = __float2int_rn(att / (float)v.dt);  // link.py:260
  if (tau <= 0) fl |= PEDN_F_SAME_STEP;
  int idx = tp + 1 - tau;
  if (idx < 0) idx = 0;
  float cf = clip01((kk - (float)P.kc) / (float)(P.kj - P.kc));  // link.py:282
  double ff = v.f64[F_CI][at(idx, l, v.Lall, RS, r)] - x.co_in;  // the one data-dependent look-back of the common path
  if (!(ff > 0.0)) ff = 0.0;
  double bnd = (double)(cf * nself) + (double)(1.0f - cf) * ff;  // link.py:284-288
  double smax = x.front_in * P.kc * P.vf * v.dt;                 // link.py:296
  double s = smax < bnd ? smax : bnd;
  double orig = s;
  RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)tp, 0u};
  if (s > 0.0) {
    float rf = clip01(dens / (float)P.kj);                          // link.py:315
    float p = 0.7f + (float)(0.85 - 0.7) * pedn_powf(rf, 0.8f);     // link.py:317
    bool diffusion_used = false;
    if (dens <= (float)P.kc) {  // link.py:323
      float F = 1.0f / (1.0f + (float)P.gamma * att);
      float G = 1.0f - F;
      const double* in = v.f64[F_IN];
      double i0 = in[at(wrap_idx(tp - tau, T1, fl), l, v.Lall, RS, r)], i1 = in[at(wrap_idx(tp - tau - 1, T1, fl), l, v.Lall, RS, r)];
      double i2 = in[at(wrap_idx(tp - tau - 2, T1, fl), l, v.Lall, RS, r)], i3 = in[at(wrap_idx(tp - tau - 3, T1, fl), l, v.Lall, RS, r)];
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
    if (!diffusion_used) {  // link.py:336-338,342-344
      key.site = 0u;
      s = rng_binomial((long long)floor(s), (double)p, key, v.meanfield);
    }
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
__device__ double recv_flow(const DevView& v, const LinkP& P, int l, int tp, int r, const SlotIn& x, double s_rev, uint32_t& fl) {
  const float nrev = x.n_in;
  double aw = P.sep ? x.sepw_out : P.width;
  double kjA = P.kj * (P.length * aw);
  int tsw = P.tau_sw;
  double b;
  if (P.sep) {
    if (tp + 1 - tsw < 0) b = kjA;
    else {
      if (tsw <= 0) fl |= PEDN_F_SAME_STEP;
      b = x.co_sw + kjA - x.ci_out;
    }
  } else {
    if (nrev < 0.0f) fl |= PEDN_F_NEG_BINOM;
    RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)tp, 2u};
    double rp = rng_binomial((long long)nrev, 0.9, key, v.meanfield);  // link.py:381-382
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

// ------------------------------------------------------------------------------------------------- kernels
// P(down | up, od) for every softmax group with more than one downstream (update_node_turn_probs, path_finder.py:561-589).
// A group with a single downstream has P = e/e = 1 exactly; those are constants and never recomputed.
template <bool PR>
__global__ __launch_bounds__(256, 8) void turn_prob_kernel(DevView v, int t) {
  const int RS = v.RS;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int gi = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)RS));
  int r = (int)(gid % (size_t)RS);
  if (gi >= v.n_multi) return;
  const int g = v.grp_multi[gi];
  uint32_t fl = 0;
  const int a = v.grp_ent_ptr[g], n = v.grp_ent_ptr[g + 1] - a;
  const int allphys = v.grp_allphys[g];
  const int t2 = wrap_idx(t - 2, v.T1, fl);
  double cap[PEDN_MAX_DEGREE - 1];
  float kf[PEDN_MAX_DEGREE - 1];
  double sumc = 0.0;
#pragma unroll
  for (int e = 0; e < PEDN_MAX_DEGREE - 1; ++e) {
    if (e < n) {
      const EntS E = v.ents[a + e];
      if (E.link >= 0) {
        if (E.sep) kf[e] = v.f32[G_K][at(t - 1, E.link, v.L, RS, r)];  // Separator.get_density, link.py:427-428
        else kf[e] = (v.f32[G_N][at(t - 1, E.link, v.L, RS, r)] + v.f32[G_N][at(t - 1, E.rev, v.L, RS, r)]) / E.area32;
        double c = v.f64[F_R][at(t2, E.link, v.L, RS, r)];
        if (!(c >= 0.0)) {
          const double vf = PR ? v.vf_r[(size_t)E.link * RS + r] : E.vf, kc = PR ? v.kc_r[(size_t)E.link * RS + r] : E.kc;
          c = v.back[(size_t)E.link * RS + r] * vf * kc * v.dt;  // :575-576
        }
        cap[e] = c;
      } else {
        kf[e] = 0.0f;
        cap[e] = 100.0;  // :577-579
      }
      sumc = (e == 0) ? cap[e] : sumc + cap[e];
    }
  }
  double ex[PEDN_MAX_DEGREE - 1];
  double esum = 0.0;
#pragma unroll
  for (int e = 0; e < PEDN_MAX_DEGREE - 1; ++e) {
    if (e < n) {
      double nd;
      if (allphys) {  // float32 array branch of :581,583
        float x = kf[e] - 2.0f;
        if (!(x > 0.0f)) x = 0.0f;
        nd = (double)((float)v.pf_beta * (x / 8.0f));
      } else {
        double x = (double)kf[e] - 2.0;
        if (!(x > 0.0)) x = 0.0;
        nd = v.pf_beta * (x / 8.0);
      }
      double u = v.ents[a + e].dist_term + nd - (v.pf_omega * cap[e]) / (sumc + 1e-6) + v.pf_eps;
      ex[e] = pedn_exp(-v.pf_temp * u);
      esum = (e == 0) ? ex[e] : esum + ex[e];
    }
  }
#pragma unroll
  for (int e = 0; e < PEDN_MAX_DEGREE - 1; ++e) {
    if (e < n) {
      const int q = v.ent_pair[a + e];  // slot of the (turn, od) product that consumes this probability
      if (q >= 0) v.ent_p[(size_t)q * RS + r] = ex[e] / esum;
    }
  }
  if (fl) atomicOr(&v.flags[r], fl);
}

// One block = 8 waves = a bin of nodes whose slot counts add up to <= 8; one wave per (node slot, 64 replicas).
template <bool PR>
__global__ __launch_bounds__(512, 6) void node_kernel(DevView v, int t) {
  __shared__ double sPS[64 * 64];  // per node m*m tiles of 64 lanes: P[i][j]*s_i, then floor(g_ij)
  __shared__ double sR[8 * 64];    // receiving flow of each wave's outgoing link
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  // blockIdx.x = replica group (fastest in dispatch order): blocks launched together touch neighbouring 512-byte chunks
  // of the same history rows
  const int r = (int)blockIdx.x * 64 + lane;
  const int tp = t - 1;
  const SlotRec& W = v.slot_rec[(size_t)blockIdx.y * 8 + wave];  // wave-uniform: scalar loads
  const int node = W.node, slot = W.slot, base = W.base, m = W.m;
  const bool active = node >= 0;
  uint32_t fl = 0;
  double s_i = 0.0, r_i = 0.0, qo = 0.0, qi = 0.0, co_prev = 0.0, ci_prev = 0.0;
  int lin = 0, lout = 0, kind = 0;
  double tfr[PEDN_MAX_DEGREE - 1];

  if (active) {
    kind = W.kind;
    lin = W.lin;
    lout = W.lout;
    const int turn0 = W.turn0 + slot * (m - 1);
    const bool static_tf = kind == 1 && !W.dyn;
    if (lin >= L) {  // virtual pair: origin demand in, unlimited sink out (node.py:176,186)
      s_i = v.demand[((size_t)W.demand_row * v.T1 + tp) * RS + r];
      co_prev = v.f64[F_CO][at(tp, lin, Lall, RS, r)];
      ci_prev = v.f64[F_CI][at(tp, lout, Lall, RS, r)];
      if (static_tf) {
        const bool shared = v.tf_u[turn0] == v.tf_u[turn0];  // not NaN
#pragma unroll
        for (int jj = 0; jj < PEDN_MAX_DEGREE - 1; ++jj)
          if (jj < m - 1) tfr[jj] = shared ? v.tf_u[turn0 + jj] : v.tf[(size_t)(turn0 + jj) * RS + r];
      }
      r_i = 1e6;
    } else {
      const LinkP Pin = lane_params<PR>(v, W.Pin, lin, r);
      const LinkP Pout = lane_params<PR>(v, W.Pout, lout, r);
      const bool early = tp < Pin.fft;  // link.py:267-269: sending flow is 0 until the first pedestrians can arrive
      uint32_t flw = 0;
      const int tm1 = wrap_idx(tp - 1, v.T1, flw);
      fl |= flw;
      // ---- one batch of independent loads (see SlotIn): straight-line and unconditional so that the compiler issues them
      // back to back (a load skipped by a branch costs a wait at the join); the few values of an `early` step are unused
      SlotIn x;
      const int t_sw = tp + 1 - Pout.tau_sw > 0 ? tp + 1 - Pout.tau_sw : 0;
      x.n_in = v.f32[G_N][at(tp, lin, L, RS, r)];
      x.n_out = v.f32[G_N][at(tp, lout, L, RS, r)];
      x.k_in = v.f32[G_K][at(tp, lin, L, RS, r)];
      x.att_in = v.f32[G_ATT][at(tp, lin, L, RS, r)];
      x.co_in = v.f64[F_CO][at(tp, lin, Lall, RS, r)];
      x.s_prev = v.f64[F_S][at(tm1, lin, L, RS, r)];
      x.co_sw = v.f64[F_CO][at(t_sw, lout, Lall, RS, r)];
      x.ci_out = v.f64[F_CI][at(tp, lout, Lall, RS, r)];
      x.r_prev = v.f64[F_R][at(tm1, lout, L, RS, r)];
      const double fu = v.front_u[lin], bu = v.back_u[lout];
      x.front_in = fu == fu ? fu : v.front[(size_t)lin * RS + r];
      x.back_out = bu == bu ? bu : v.back[(size_t)lout * RS + r];
      x.sepw_in = Pin.sep ? v.sepw[(size_t)lin * RS + r] : 0.0;
      x.sepw_out = Pout.sep ? v.sepw[(size_t)lout * RS + r] : 0.0;
      if (static_tf) {
        const bool shared = v.tf_u[turn0] == v.tf_u[turn0];  // not NaN
#pragma unroll
        for (int jj = 0; jj < PEDN_MAX_DEGREE - 1; ++jj)
          if (jj < m - 1) tfr[jj] = shared ? v.tf_u[turn0 + jj] : v.tf[(size_t)(turn0 + jj) * RS + r];
      }
      co_prev = x.co_in;   // cumulative_outflow[t-1] of the incoming link, reused by update_links below
      ci_prev = x.ci_out;  // cumulative_inflow[t-1] of the outgoing link
      s_i = early ? 0.0 : send_flow(v, Pin, lin, tp, r, x, fl);
      v.f64[F_S][at(tp, lin, L, RS, r)] = s_i;  // link.py:268,367
      if (s_i < 0.0) fl |= PEDN_F_NEG_FLOW;
      r_i = recv_flow(v, Pout, lout, tp, r, x, s_i, fl);
      v.f64[F_R][at(tp, lout, L, RS, r)] = r_i;  // node.py:206
    }
    if (s_i < 0.0 || r_i < 0.0) fl |= PEDN_F_NEG_FLOW;

    if (kind == 1) {
      // turning fractions of row `slot`: static, or recomputed from the route-choice tables (path_finder.py:591-715)
      if (W.dyn) {
        // tf[turn] = sum over the turn's (od) products P(down | up, od) * P(od | up)   (path_finder.py:668-686).
        // The products of one row are contiguous: probabilities were stored in pair order by turn_prob_kernel and
        // P(od | up) (replica independent, :599-615) was tabulated per pair and step on the host.
        const double* pod = v.pair_pod + (size_t)t * v.n_pair;
        const bool ppr = v.pod_pr != 0;  // per-replica OD weights: tables indexed [product][replica] instead of [step][product]
        double rowsum = 0.0;
#pragma unroll
        for (int jj = 0; jj < PEDN_MAX_DEGREE - 1; ++jj) {
          if (jj < m - 1) {
            double acc = 0.0;
            const int q1 = v.turn_mode[turn0 + jj] ? 0 : v.turn_pair_ptr[turn0 + jj + 1];
            if (v.turn_mode[turn0 + jj]) acc = ppr ? v.turn_tab_r[(size_t)(turn0 + jj) * RS + r] : v.turn_tab[(size_t)t * v.n_turns + turn0 + jj];
            for (int q = v.turn_pair_ptr[turn0 + jj]; q < q1; q += 4) {
              // loads first (independent), then the strictly sequential sum the reference performs
              double e0 = v.pair_const[q] ? 1.0 : v.ent_p[(size_t)q * RS + r];
              double e1 = q + 1 < q1 ? (v.pair_const[q + 1] ? 1.0 : v.ent_p[(size_t)(q + 1) * RS + r]) : 0.0;
              double e2 = q + 2 < q1 ? (v.pair_const[q + 2] ? 1.0 : v.ent_p[(size_t)(q + 2) * RS + r]) : 0.0;
              double e3 = q + 3 < q1 ? (v.pair_const[q + 3] ? 1.0 : v.ent_p[(size_t)(q + 3) * RS + r]) : 0.0;
              acc += e0 * (ppr ? v.pair_pod_r[(size_t)q * RS + r] : pod[q]);
              if (q + 1 < q1) acc += e1 * (ppr ? v.pair_pod_r[(size_t)(q + 1) * RS + r] : pod[q + 1]);
              if (q + 2 < q1) acc += e2 * (ppr ? v.pair_pod_r[(size_t)(q + 2) * RS + r] : pod[q + 2]);
              if (q + 3 < q1) acc += e3 * (ppr ? v.pair_pod_r[(size_t)(q + 3) * RS + r] : pod[q + 3]);
            }
            tfr[jj] = acc;
            rowsum = (jj == 0) ? acc : rowsum + acc;
          }
        }
        const bool renorm = fabs(rowsum - 1) > 1e-3;  // check_fractions, :700-714
#pragma unroll
        for (int jj = 0; jj < PEDN_MAX_DEGREE - 1; ++jj) {
          if (jj < m - 1) {
            if (renorm) tfr[jj] = rowsum > 1e-6 ? tfr[jj] / rowsum : 1.0 / (double)(m - 1);
            v.tf[(size_t)(turn0 + jj) * RS + r] = tfr[jj];
          }
        }
      }
      // P[i][j] * s_i  (node.py:285)
#pragma unroll
      for (int jj = 0; jj < PEDN_MAX_DEGREE - 1; ++jj) {
        if (jj < m - 1) {
          const int j = jj < slot ? jj : jj + 1;
          sPS[(size_t)(base + slot * m + j) * 64 + lane] = tfr[jj] * s_i;
        }
      }
    } else {
      sPS[(size_t)(base + slot) * 64 + lane] = s_i;
    }
    sR[wave * 64 + lane] = r_i;
  }
  __syncthreads();

  if (active && kind == 1) {
    // column `slot`: D_j = sum_i P[i][j] s_i (i ascending), g_ij = floor(min(P s, r_j * (P s / D_j)))  (node.py:286-298)
    double D = 0.0;
    bool first = true;
#pragma unroll
    for (int k = 0; k < PEDN_MAX_DEGREE; ++k) {
      if (k < m && k != slot) {
        double x = sPS[(size_t)(base + k * m + slot) * 64 + lane];
        D = first ? x : D + x;
        first = false;
      }
    }
    const double Ds = D != 0.0 ? D : 1e-5;
#pragma unroll
    for (int k = 0; k < PEDN_MAX_DEGREE; ++k) {
      if (k < m && k != slot) {
        double a = sPS[(size_t)(base + k * m + slot) * 64 + lane];
        double b = r_i * (a / Ds);
        double g = floor(b < a ? b : a);
        sPS[(size_t)(base + k * m + slot) * 64 + lane] = g;
        qi += g;
      }
    }
  }
  __syncthreads();

  if (active) {
    if (kind == 1) {
#pragma unroll
      for (int j = 0; j < PEDN_MAX_DEGREE; ++j)
        if (j < m && j != slot) qo += sPS[(size_t)(base + slot * m + j) * 64 + lane];
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
    v.f64[F_OUT][at(t, lin, Lall, RS, r)] = qo;
    v.f64[F_CO][at(t, lin, Lall, RS, r)] = co_prev + qo;
    v.f64[F_IN][at(t, lout, Lall, RS, r)] = qi;
    v.f64[F_CI][at(t, lout, Lall, RS, r)] = ci_prev + qi;
    if (fl) atomicOr(&v.flags[r], fl);
  }
}

// BiDirectionalFd.__call__ + the travel-time part of Link.update_speeds for one direction and one replica; pure arithmetic
struct SpeedOut { float spd, tt, lf, att, rs; };

__device__ __forceinline__ SpeedOut speed_calc(const DevView& v, const LinkP& P, int l, int t, int r, float ks, float ko,
                                               float rsum_prev, float tt_old) {
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
  if (P.noise > 0.0) {  // functions.py:132-133
    double nz = 0.0;
    if (!v.meanfield) {
      RngKey key{v.k0, v.k1, v.replica_offset + (uint32_t)r, (uint32_t)l, (uint32_t)t, 3u};
      nz = P.noise * rng_z(key);
    }
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

__device__ __forceinline__ double2 ld2(const double* p, size_t i) { return *reinterpret_cast<const double2*>(p + i); }
__device__ __forceinline__ float2 ld2(const float* p, size_t i) { return *reinterpret_cast<const float2*>(p + i); }
__device__ __forceinline__ void st2(double* p, size_t i, double a, double b) { *reinterpret_cast<double2*>(p + i) = make_double2(a, b); }
__device__ __forceinline__ void st2(float* p, size_t i, float a, float b) { *reinterpret_cast<float2*>(p + i) = make_float2(a, b); }

// Network.update_link_states (network.py:257-264).  One lane = both directions of one corridor for TWO adjacent replicas:
// every history access is a 16-byte (f64) or 8-byte (f32) vector access, i.e. 1 KiB / 512 B per wave instruction.
__global__ __launch_bounds__(256) void link_kernel(DevView v, int t) {
  const int RS = v.RS, L = v.L, Lall = v.Lall, H = RS / 2;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int p = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)H));
  const int r = 2 * (int)(gid % (size_t)H);
  if (p >= v.n_pairs_corr) return;
  const CorrRec& C = v.corr_rec[p];  // wave-uniform
  const int a = C.a, b = C.b;
  const LinkP& Pa = C.Pa;
  const LinkP& Pb = C.Pb;
  const bool win = t >= v.W;
  // ---- loads
  const double2 ina = ld2(v.f64[F_IN], at(t, a, Lall, RS, r)), outa = ld2(v.f64[F_OUT], at(t, a, Lall, RS, r));
  const double2 inb = ld2(v.f64[F_IN], at(t, b, Lall, RS, r)), outb = ld2(v.f64[F_OUT], at(t, b, Lall, RS, r));
  const float2 pa = ld2(v.f32[G_N], at(t - 1, a, L, RS, r)), pb = ld2(v.f32[G_N], at(t - 1, b, L, RS, r));
  const float2 rsa = ld2(v.rsum, (size_t)a * RS + r), rsb = ld2(v.rsum, (size_t)b * RS + r);
  const float2 oa = win ? ld2(v.f32[G_TT], at(t - v.W, a, L, RS, r)) : make_float2(0.f, 0.f);
  const float2 ob = win ? ld2(v.f32[G_TT], at(t - v.W, b, L, RS, r)) : make_float2(0.f, 0.f);
  const double bua = v.back_u[a], bub = v.back_u[b];
  double2 wa = make_double2(Pa.width, Pa.width), wb = make_double2(Pb.width, Pb.width), fa = make_double2(0, 0), fb = make_double2(0, 0);
  if (Pa.sep) { wa = ld2(v.sepw, (size_t)a * RS + r); fa = ld2(v.sepnp, (size_t)a * RS + r); }
  if (Pb.sep) { wb = ld2(v.sepw, (size_t)b * RS + r); fb = ld2(v.sepnp, (size_t)b * RS + r); }
  double2 ga = Pa.sep ? wa : make_double2(bua, bua), gb = Pb.sep ? wb : make_double2(bub, bub);  // recorded width (link.py:188 / :451-452)
  if (!Pa.sep && !(bua == bua)) ga = ld2(v.back, (size_t)a * RS + r);
  if (!Pb.sep && !(bub == bub)) gb = ld2(v.back, (size_t)b * RS + r);
  // ---- per replica arithmetic (link.py:133-136, then update_speeds)
  float na[2], nb[2], ka[2], kb[2];
  SpeedOut sa[2], sb[2];
  const double dina[2] = {ina.x - outa.x, ina.y - outa.y}, dinb[2] = {inb.x - outb.x, inb.y - outb.y};
  const float pav[2] = {pa.x, pa.y}, pbv[2] = {pb.x, pb.y};
  const double wav[2] = {wa.x, wa.y}, wbv[2] = {wb.x, wb.y}, fav[2] = {fa.x, fa.y}, fbv[2] = {fb.x, fb.y};
  const float rsav[2] = {rsa.x, rsa.y}, rsbv[2] = {rsb.x, rsb.y}, oav[2] = {oa.x, oa.y}, obv[2] = {ob.x, ob.y};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    na[j] = (float)((double)pav[j] + dina[j]);
    nb[j] = (float)((double)pbv[j] + dinb[j]);
    // a separator width held as np.float64 turns the division into binary64 (PEDN_W_SEP_NUMPY)
    ka[j] = (Pa.sep && fav[j] != 0.0) ? (float)((double)na[j] / (Pa.length * wav[j])) : na[j] / (float)(Pa.length * wav[j]);
    kb[j] = (Pb.sep && fbv[j] != 0.0) ? (float)((double)nb[j] / (Pb.length * wbv[j])) : nb[j] / (float)(Pb.length * wbv[j]);
    sa[j] = speed_calc(v, Pa, a, t, r + j, ka[j], kb[j], rsav[j], oav[j]);
    sb[j] = speed_calc(v, Pb, b, t, r + j, kb[j], ka[j], rsbv[j], obv[j]);
  }
  // ---- stores
  st2(v.f32[G_N], at(t, a, L, RS, r), na[0], na[1]);
  st2(v.f32[G_N], at(t, b, L, RS, r), nb[0], nb[1]);
  st2(v.f32[G_K], at(t, a, L, RS, r), ka[0], ka[1]);
  st2(v.f32[G_K], at(t, b, L, RS, r), kb[0], kb[1]);
  st2(v.f32[G_V], at(t, a, L, RS, r), sa[0].spd, sa[1].spd);
  st2(v.f32[G_V], at(t, b, L, RS, r), sb[0].spd, sb[1].spd);
  st2(v.f32[G_TT], at(t, a, L, RS, r), sa[0].tt, sa[1].tt);
  st2(v.f32[G_TT], at(t, b, L, RS, r), sb[0].tt, sb[1].tt);
  st2(v.f32[G_LF], at(t, a, L, RS, r), sa[0].lf, sa[1].lf);
  st2(v.f32[G_LF], at(t, b, L, RS, r), sb[0].lf, sb[1].lf);
  if (win) {
    st2(v.f32[G_ATT], at(t, a, L, RS, r), sa[0].att, sa[1].att);
    st2(v.f32[G_ATT], at(t, b, L, RS, r), sb[0].att, sb[1].att);
  }
  st2(v.rsum, (size_t)a * RS + r, sa[0].rs, sa[1].rs);
  st2(v.rsum, (size_t)b * RS + r, sb[0].rs, sb[1].rs);
  // the record is initialised to `width` (link.py:56), so an unchanged gate needs no store
  if (ga.x != Pa.width || ga.y != Pa.width) st2(v.f64[F_GATE], at(t, a, L, RS, r), ga.x, ga.y);
  if (gb.x != Pb.width || gb.y != Pb.width) st2(v.f64[F_GATE], at(t, b, L, RS, r), gb.x, gb.y);
}

// Same update with per-replica link parameters: one replica per lane (the parameters live in vector registers).
__global__ __launch_bounds__(256) void link_kernel_pr(DevView v, int t) {
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int p = __builtin_amdgcn_readfirstlane((int)(gid / (size_t)RS));
  const int r = (int)(gid % (size_t)RS);
  if (p >= v.n_pairs_corr) return;
  const CorrRec& C = v.corr_rec[p];
  const int a = C.a, b = C.b;
  const LinkP Pa = lane_params<true>(v, C.Pa, a, r), Pb = lane_params<true>(v, C.Pb, b, r);
  const bool win = t >= v.W;
  const double da = v.f64[F_IN][at(t, a, Lall, RS, r)] - v.f64[F_OUT][at(t, a, Lall, RS, r)];
  const double db = v.f64[F_IN][at(t, b, Lall, RS, r)] - v.f64[F_OUT][at(t, b, Lall, RS, r)];
  const float na = (float)((double)v.f32[G_N][at(t - 1, a, L, RS, r)] + da), nb = (float)((double)v.f32[G_N][at(t - 1, b, L, RS, r)] + db);
  const double wa = Pa.sep ? v.sepw[(size_t)a * RS + r] : Pa.width, wb = Pb.sep ? v.sepw[(size_t)b * RS + r] : Pb.width;
  const float ka = (Pa.sep && v.sepnp[(size_t)a * RS + r] != 0.0) ? (float)((double)na / (Pa.length * wa)) : na / (float)(Pa.length * wa);
  const float kb = (Pb.sep && v.sepnp[(size_t)b * RS + r] != 0.0) ? (float)((double)nb / (Pb.length * wb)) : nb / (float)(Pb.length * wb);
  const float oa = win ? v.f32[G_TT][at(t - v.W, a, L, RS, r)] : 0.0f, ob = win ? v.f32[G_TT][at(t - v.W, b, L, RS, r)] : 0.0f;
  const SpeedOut sa = speed_calc(v, Pa, a, t, r, ka, kb, v.rsum[(size_t)a * RS + r], oa);
  const SpeedOut sb = speed_calc(v, Pb, b, t, r, kb, ka, v.rsum[(size_t)b * RS + r], ob);
  const double ga = Pa.sep ? wa : v.back[(size_t)a * RS + r], gb = Pb.sep ? wb : v.back[(size_t)b * RS + r];
  v.f32[G_N][at(t, a, L, RS, r)] = na; v.f32[G_N][at(t, b, L, RS, r)] = nb;
  v.f32[G_K][at(t, a, L, RS, r)] = ka; v.f32[G_K][at(t, b, L, RS, r)] = kb;
  v.f32[G_V][at(t, a, L, RS, r)] = sa.spd; v.f32[G_V][at(t, b, L, RS, r)] = sb.spd;
  v.f32[G_TT][at(t, a, L, RS, r)] = sa.tt; v.f32[G_TT][at(t, b, L, RS, r)] = sb.tt;
  v.f32[G_LF][at(t, a, L, RS, r)] = sa.lf; v.f32[G_LF][at(t, b, L, RS, r)] = sb.lf;
  if (win) { v.f32[G_ATT][at(t, a, L, RS, r)] = sa.att; v.f32[G_ATT][at(t, b, L, RS, r)] = sb.att; }
  v.rsum[(size_t)a * RS + r] = sa.rs; v.rsum[(size_t)b * RS + r] = sb.rs;
  if (ga != Pa.width) v.f64[F_GATE][at(t, a, L, RS, r)] = ga;
  if (gb != Pb.width) v.f64[F_GATE][at(t, b, L, RS, r)] = gb;
}

// ---- batched RL glue (rl/builders.py, rl/pz_pednet_env.py:548-581) --------------------------------------------------
struct RlView {
  const int32_t *agent_type, *agent_link_ptr, *agent_links, *agent_act_off, *agent_obs_off;
  const int32_t *slot_agent, *slot_idx;  // action slot -> (agent, position in the agent's link list)
  double* actions;                        // [R][A]
  float *obs, *rew;                       // [R][O], [R][n_agents]
  int32_t n_agents, A, O, obs_mode, normalize, reward_mode, fpl;
  double max_delta_sep, max_delta_gate, min_sep;
};

__device__ __forceinline__ double clip_d(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }  // np.clip

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

// ObservationBuilder + _compute_rewards.  Block = (agent, 64 replicas); wave w = the agent's w-th controlled link: its
// features go straight to the observation row, its reward terms to LDS; wave 0 then folds them in the reference's order.
__global__ __launch_bounds__(512) void rl_observe_kernel(DevView v, RlView q, int t, int accumulate) {
  __shared__ float sT[PEDN_MAX_DEGREE][64], sD[PEDN_MAX_DEGREE][64], sKc[PEDN_MAX_DEGREE][64];
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int r = (int)blockIdx.x * 64 + lane;
  const int ag = (int)blockIdx.y;
  const int RS = v.RS, L = v.L, Lall = v.Lall;
  const int la = q.agent_link_ptr[ag], n = q.agent_link_ptr[ag + 1] - la;
  const int type = q.agent_type[ag];
  const bool live = r < v.R;
  float* o = q.obs + (size_t)(live ? r : 0) * q.O + q.agent_obs_off[ag];
  if (type == 0) {  // separator agent, builders.py:87-117
    if (w == 0 && live) {
      const int f = q.agent_links[la], b = q.agent_links[la + 1];
      float x[4] = {(float)v.f64[F_IN][at(t, f, Lall, RS, r)], (float)v.f64[F_OUT][at(t, f, Lall, RS, r)],
                    (float)v.f64[F_IN][at(t, b, Lall, RS, r)], (float)v.f64[F_OUT][at(t, b, Lall, RS, r)]};
      for (int k = 0; k < 4; ++k) {
        if (q.normalize && (q.obs_mode == 1 || q.obs_mode == 2)) x[k] = x[k] / 20.0f;  // builders.py:183-188
        o[k] = x[k];
      }
    }
  } else if (w < n) {  // gater agent, link w: builders.py:119-177
    const int l = q.agent_links[la + w];
    const LinkP P = v.pr ? lane_params<true>(v, v.lp[l], l, r) : v.lp[l];
    const float in_l = (float)v.f64[F_IN][at(t, l, Lall, RS, r)], out_l = (float)v.f64[F_OUT][at(t, l, Lall, RS, r)];
    const float in_r = (float)v.f64[F_IN][at(t, P.rev, Lall, RS, r)], out_r = (float)v.f64[F_OUT][at(t, P.rev, Lall, RS, r)];
    const float tt_l = v.f32[G_TT][at(t, l, L, RS, r)], tt_r = v.f32[G_TT][at(t, P.rev, L, RS, r)];
    const float spd = q.obs_mode == 5 ? v.f32[G_V][at(t, l, L, RS, r)] : 0.0f;
    const float dens = dens_at(v, P, l, t, r);
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
  __syncthreads();
  if (w == 0 && live) {
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

// ---- state initialisation / host <-> device helpers ---------------------------------------------------------
__global__ void init_state_kernel(DevView v) {
  const int RS = v.RS, L = v.L, T1 = v.T1;
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)T1 * L * RS;
  if (gid >= total) return;
  int r = (int)(gid % RS);
  int l = (int)((gid / RS) % L);
  int t = (int)(gid / ((size_t)RS * L));
  const LinkP P = v.pr ? lane_params<true>(v, v.lp[l], l, r) : v.lp[l];
  v.f64[F_S][gid] = -1.0;
  v.f64[F_R][gid] = -1.0;
  v.f64[F_GATE][gid] = P.width;  // link.py:56
  v.f32[G_TT][gid] = t == 0 ? P.tt0 : 0.0f;
  v.f32[G_ATT][gid] = t < v.W ? P.tt0 : 0.0f;  // link.py:91
  v.f32[G_N][gid] = 0.0f;
  v.f32[G_K][gid] = 0.0f;
  v.f32[G_V][gid] = 0.0f;
  v.f32[G_LF][gid] = 0.0f;
  if (t == 0) v.rsum[(size_t)l * RS + r] = P.tt0;  // link.py:84
}

template <typename T>
__global__ void gather_kernel(const T* src, T* dst, int t0, int nt, int c0, int nc, int r0, int nr, int cols, int RS) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nt * nc * nr;
  if (gid >= total) return;
  int r = (int)(gid % nr);
  int c = (int)((gid / nr) % nc);
  int t = (int)(gid / ((size_t)nr * nc));
  dst[gid] = src[((size_t)(t0 + t) * cols + (c0 + c)) * RS + (r0 + r)];
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

__global__ void device_math_kernel(int op, int n, const double* a, const double* b, uint32_t k0, uint32_t k1, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
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

// ------------------------------------------------------------------------------------------------- host side
static thread_local std::string g_last_error;

struct pedn_sim {
  DevView v{};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int device = 0;
  int n_nodes = 0, n_turns = 0, n_demand = 0, n_od = 0, n_blocks = 0, n_ent = 0;
  std::vector<int32_t> node_turn_ptr, node_demand_row;
  std::vector<int32_t> h_up_od_ptr, h_upod_od, h_pair_upod;  // route-choice tables needed to re-tabulate P(od | up)
  std::vector<double> h_od_w;
  std::vector<int32_t> h_turn_pair_ptr, h_pair_const, h_turn_mode;
  double* d_pair_pod = nullptr;
  double* d_turn_tab = nullptr;
  RlView rl{};
  bool rl_ready = false;
  std::vector<double> h_front_u, h_back_u, h_tf_u;
  double *d_front_u = nullptr, *d_back_u = nullptr, *d_tf_u = nullptr;
  std::vector<int32_t> h_node_dyn;
  std::vector<char> h_rl_link;
  double *d_kc_r = nullptr, *d_kj_r = nullptr, *d_vf_r = nullptr, *d_pair_pod_r = nullptr, *d_turn_tab_r = nullptr;
  int32_t *d_fft_r = nullptr, *d_tausw_r = nullptr;
  float* d_tt0_r = nullptr;  // links whose widths the RL action kernel writes per replica: never uniform
  int n_pair = 0, n_up = 0;
  std::vector<void*> allocs;
  void* stage = nullptr;
  size_t stage_bytes = 0;
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

static int ensure_stage(pedn_sim* s, size_t bytes) {
  if (bytes <= s->stage_bytes) return PEDN_OK;
  if (s->stage) HIP_TRY(s, hipFree(s->stage));
  s->stage = nullptr;
  s->stage_bytes = 0;
  size_t want = std::max<size_t>(bytes, 1 << 20);
  HIP_TRY(s, hipMalloc(&s->stage, want));
  s->stage_bytes = want;
  return PEDN_OK;
}

static int reset_state(pedn_sim* s) {
  DevView& v = s->v;
  size_t n_all = (size_t)v.T1 * v.Lall * v.RS, n_l = (size_t)v.T1 * v.L * v.RS;
  for (int f = 0; f < 4; ++f) HIP_TRY(s, hipMemsetAsync(v.f64[f], 0, n_all * sizeof(double), s->stream));
  HIP_TRY(s, hipMemsetAsync(v.flags, 0, (size_t)v.RS * sizeof(uint32_t), s->stream));
  if (v.L > 0) {
    unsigned blocks = (unsigned)((n_l + 255) / 256);
    hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s->stream, v);
    HIP_TRY(s, hipGetLastError());
  }
  return PEDN_OK;
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

extern "C" {

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

  HIP_TRY(nullptr, hipSetDevice(device));
  pedn_sim* s = new pedn_sim();
  s->device = device;
  DevView& v = s->v;
  v.L = L;
  v.Lall = L + m->n_vlinks;
  v.T1 = m->T + 1;
  v.R = n_replicas;
  v.RS = (n_replicas + 127) / 128 * 128;  // a link_kernel wave covers 128 replicas (2 per lane), a node_kernel wave 64
  v.W = m->window;
  v.dt = m->dt;
  v.pf_temp = m->pf_temp; v.pf_alpha = m->pf_alpha; v.pf_beta = m->pf_beta; v.pf_omega = m->pf_omega; v.pf_eps = m->pf_eps;
  v.k0 = (uint32_t)seed; v.k1 = (uint32_t)(seed >> 32);
  v.replica_offset = (uint32_t)replica_offset;
  v.meanfield = rng_mode == PEDN_RNG_MEANFIELD;
  v.n_grp = m->n_grp;
  s->n_nodes = N; s->n_turns = m->n_turns; s->n_demand = m->n_demand; s->n_od = m->n_od; s->n_ent = m->n_ent;
  s->node_turn_ptr.assign(m->node_turn_ptr, m->node_turn_ptr + N + 1);
  s->node_demand_row.assign(m->node_demand_row, m->node_demand_row + N);

#define TRY(expr) do { int _rc = (expr); if (_rc != PEDN_OK) { std::string keep = g_last_error; pedn_destroy(s); g_last_error = keep; return _rc; } } while (0)
  {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return fail(nullptr, PEDN_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    hipEventCreate(&s->ev0);
    hipEventCreate(&s->ev1);
  }
  // ---- memory budget
  {
    size_t RS = v.RS, T1 = v.T1;
    size_t need = 4 * T1 * v.Lall * RS * 8 + 3 * T1 * (size_t)L * RS * 8 + 6 * T1 * (size_t)L * RS * 4 + (size_t)m->n_demand * T1 * RS * 8 +
                  (size_t)(m->n_turns + m->n_ent + 3 * L) * RS * 8;
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
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
    P.tau_sw = m->link_tau_sw[l]; P.fft = m->link_fft[l]; P.pad[0] = P.pad[1] = 0;
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
    TRY(upload(s, ent_pair.data(), ent_pair.size(), &v.ent_pair));
    // exp(x)/exp(x) == 1 exactly as long as exp(x) is finite and non-zero; |x| <= temp * (|alpha| + |beta|*k_max/8 + |omega| + |eps|)
    const double xmax = fabs(m->pf_temp) * (fabs(m->pf_alpha) + fabs(m->pf_beta) * 16.0 + fabs(m->pf_omega) + fabs(m->pf_eps));
    const bool shortcut = xmax < 600.0;
    std::vector<EntS> ents(std::max(m->n_ent, 1));
    std::vector<int32_t> multi, pconst(std::max(m->n_pair, 1), 0);
    for (int g = 0; g < m->n_grp; ++g) {
      const int a = m->grp_ent_ptr[g], b = m->grp_ent_ptr[g + 1];
      double sumd = 0.0;
      for (int e = a; e < b; ++e) sumd = (e == a) ? m->ent_dist[e] : sumd + m->ent_dist[e];
      for (int e = a; e < b; ++e) {
        EntS& E = ents[e];
        E.link = m->ent_link[e];
        E.rev = E.sep = 0; E.area32 = 1.0f; E.vf = E.kc = 0.0;
        if (E.link >= 0) {
          if (E.link >= L) { pedn_destroy(s); return fail(nullptr, PEDN_E_ARG, "ent_link out of range"); }
          E.rev = m->link_rev[E.link]; E.sep = m->link_sep[E.link];
          E.area32 = (float)(m->link_length[E.link] * m->link_width[E.link]);
          E.vf = m->link_vf[E.link]; E.kc = m->link_kc[E.link];
        }
        E.dist_term = (m->pf_alpha * m->ent_dist[e]) / (sumd + 1e-6);
      }
      if (b - a == 1 && shortcut) { if (ent_pair[a] >= 0) pconst[ent_pair[a]] = 1; }
      else if (b - a >= 1) multi.push_back(g);
    }
    v.n_multi = (int)multi.size();
    TRY(upload(s, ents.data(), ents.size(), &v.ents));
    TRY(upload(s, multi.data(), multi.size(), &v.grp_multi));
    TRY(upload(s, pconst.data(), pconst.size(), &v.pair_const));
    s->h_pair_const = pconst;
    s->h_turn_pair_ptr.assign(m->turn_pair_ptr, m->turn_pair_ptr + m->n_turns + 1);
    s->h_turn_mode.assign(std::max(m->n_turns, 1), 0);
    for (int tn = 0; tn < m->n_turns; ++tn) {
      bool all_const = true;   // a turn without products is the constant 0
      for (int q = m->turn_pair_ptr[tn]; q < m->turn_pair_ptr[tn + 1]; ++q) all_const = all_const && pconst[q];
      s->h_turn_mode[tn] = all_const ? 1 : 0;
    }
    TRY(upload(s, s->h_turn_mode.data(), s->h_turn_mode.size(), &v.turn_mode));
    v.n_turns = m->n_turns;
  }
  TRY(upload(s, m->grp_ent_ptr, m->n_grp + 1, &v.grp_ent_ptr));
  TRY(upload(s, m->grp_allphys, m->n_grp, &v.grp_allphys));
  TRY(upload(s, m->turn_pair_ptr, m->n_turns + 1, &v.turn_pair_ptr));
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
    TRY(upload(s, cr.data(), cr.size(), &v.corr_rec));
  }
  {  // bin nodes into blocks of 8 waves (first-fit decreasing on the slot count)
    std::vector<int> order(N);
    for (int n = 0; n < N; ++n) order[n] = n;
    auto deg = [&](int n) { return m->node_slot_ptr[n + 1] - m->node_slot_ptr[n]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return deg(a) > deg(b); });
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
    std::vector<SlotRec> rec(bins.size() * 8, idle);
    for (size_t b = 0; b < bins.size(); ++b) {
      int wave = 0, base = 0;
      for (int n : bins[b]) {
        int d = deg(n);
        for (int k = 0; k < d; ++k) {
          SlotRec& R = rec[b * 8 + wave++];
          R.node = n; R.slot = k; R.base = base; R.m = d;
          R.kind = m->node_kind[n]; R.dyn = m->node_dyn[n];
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
    TRY(upload(s, rec.data(), rec.size(), &v.slot_rec));
  }
  // ---- dynamic state
  {
    size_t RS = v.RS, T1 = v.T1;
    for (int f = 0; f < 4; ++f) TRY(dalloc(s, T1 * v.Lall * RS, &v.f64[f]));
    for (int f = 4; f < 7; ++f) TRY(dalloc(s, T1 * (size_t)L * RS, &v.f64[f]));
    for (int f = 0; f < 6; ++f) TRY(dalloc(s, T1 * (size_t)L * RS, &v.f32[f]));
    TRY(dalloc(s, (size_t)L * RS, &v.rsum));
    TRY(dalloc(s, (size_t)L * RS, &v.front));
    TRY(dalloc(s, (size_t)L * RS, &v.back));
    TRY(dalloc(s, (size_t)L * RS, &v.sepw));
    TRY(dalloc(s, (size_t)L * RS, &v.sepnp));
    HIP_TRY(s, hipMemset(v.sepnp, 0, std::max<size_t>((size_t)L * RS, 1) * sizeof(double)));
    TRY(dalloc(s, (size_t)m->n_turns * RS, &v.tf));
    TRY(dalloc(s, (size_t)m->n_demand * T1 * RS, &v.demand));
    TRY(dalloc(s, (size_t)std::max(m->n_pair, m->n_ent) * RS, &v.ent_p));
    TRY(dalloc(s, (size_t)m->n_pair * T1, &s->d_pair_pod));
    v.pair_pod = s->d_pair_pod;
    TRY(dalloc(s, (size_t)std::max(m->n_turns, 1) * T1, &s->d_turn_tab));
    v.turn_tab = s->d_turn_tab;
    HIP_TRY(s, hipMemset(s->d_turn_tab, 0, (size_t)std::max(m->n_turns, 1) * T1 * sizeof(double)));
    TRY(dalloc(s, RS, &v.flags));
  }
  // initial widths, turning fractions, demand (broadcast to every replica)
  {
    const double* w0[3] = {m->front_gate0, m->back_gate0, m->sep_width0};
    double* dst[3] = {v.front, v.back, v.sepw};
    for (int k = 0; k < 3; ++k) {
      if (L == 0) break;
      TRY(ensure_stage(s, (size_t)L * 8));
      HIP_TRY(s, hipMemcpyAsync(s->stage, w0[k], (size_t)L * 8, hipMemcpyHostToDevice, s->stream));
      size_t n = (size_t)L * v.RS;
      hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst[k], (const double*)s->stage, L,
                         (size_t)0, (size_t)1, v.RS, 0, v.RS, 0, 1);
      HIP_TRY(s, hipStreamSynchronize(s->stream));
    }
    if (m->n_turns) {
      TRY(ensure_stage(s, (size_t)m->n_turns * 8));
      HIP_TRY(s, hipMemcpyAsync(s->stage, m->tf_init, (size_t)m->n_turns * 8, hipMemcpyHostToDevice, s->stream));
      size_t n = (size_t)m->n_turns * v.RS;
      hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v.tf, (const double*)s->stage,
                         m->n_turns, (size_t)0, (size_t)1, v.RS, 0, v.RS, 0, 1);
      HIP_TRY(s, hipStreamSynchronize(s->stream));
    }
    if (m->n_demand) {
      size_t rows = (size_t)m->n_demand * v.T1;
      TRY(ensure_stage(s, rows * 8));
      HIP_TRY(s, hipMemcpyAsync(s->stage, m->demand, rows * 8, hipMemcpyHostToDevice, s->stream));
      size_t n = rows * v.RS;
      hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v.demand, (const double*)s->stage,
                         (int)rows, (size_t)0, (size_t)1, v.RS, 0, v.RS, 0, 1);
      HIP_TRY(s, hipStreamSynchronize(s->stream));
    }
    if (m->n_ent) HIP_TRY(s, hipMemsetAsync(v.ent_p, 0, (size_t)std::max(m->n_pair, m->n_ent) * v.RS * 8, s->stream));
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
#undef TRY
  *out = s;
  return PEDN_OK;
}

int pedn_destroy(pedn_sim* s) {
  if (!s) return PEDN_OK;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  for (void* p : s->allocs) hipFree(p);
  if (s->stage) hipFree(s->stage);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
  return PEDN_OK;
}

int pedn_reset(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  return reset_state(s);
}

// values (host) -> rows of a [rows][RS] device array, one replica or all
static int push_rows(pedn_sim* s, double* dst, const double* values, int n_rows, size_t row0, size_t row_stride, int replica) {
  if (n_rows <= 0) return PEDN_OK;
  DevView& v = s->v;
  if (replica != PEDN_ALL && (replica < 0 || replica >= v.R)) return fail(s, PEDN_E_ARG, "replica out of range");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));  // the staging buffer may still be read by an earlier scatter
  int rc = ensure_stage(s, (size_t)n_rows * 8);
  if (rc != PEDN_OK) return rc;
  HIP_TRY(s, hipMemcpyAsync(s->stage, values, (size_t)n_rows * 8, hipMemcpyHostToDevice, s->stream));
  int r0 = replica == PEDN_ALL ? 0 : replica, r1 = replica == PEDN_ALL ? v.RS : replica + 1;
  size_t n = (size_t)n_rows * (r1 - r0);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst, (const double*)s->stage, n_rows,
                     row0, row_stride, v.RS, r0, r1, 0, 1);
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_set_demand(pedn_sim* s, int32_t node, int32_t replica, const double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  int row = s->node_demand_row[node];
  if (row < 0) return fail(s, PEDN_E_ARG, "node has no virtual (origin/destination) link");
  std::vector<double> full(s->v.T1, 0.0);
  for (int i = 0; i < n && i < s->v.T1; ++i) full[i] = values[i];
  return push_rows(s, s->v.demand, full.data(), s->v.T1, (size_t)row * s->v.T1, 1, replica);
}

int pedn_set_od_weights(pedn_sim* s, int32_t od, const double* values, int32_t n) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (od < 0 || od >= s->n_od || n != s->v.T1) return fail(s, PEDN_E_ARG, "od index or length out of range");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy((void*)(s->v.od_w + (size_t)od * s->v.T1), values, (size_t)n * 8, hipMemcpyHostToDevice));
  std::copy(values, values + n, s->h_od_w.begin() + (size_t)od * s->v.T1);
  return tabulate_pair_pod(s);
}

int pedn_set_turning_fractions(pedn_sim* s, int32_t node, int32_t replica, const double* tf, int32_t n) {
  if (!s || !tf) return fail(s, PEDN_E_ARG, "null argument");
  if (node < 0 || node >= s->n_nodes) return fail(s, PEDN_E_ARG, "node out of range");
  int a = s->node_turn_ptr[node], b = s->node_turn_ptr[node + 1];
  if (n != b - a) return fail(s, PEDN_E_ARG, "turning-fraction count does not match m(m-1)");
  int rc = push_rows(s, s->v.tf, tf, n, (size_t)a, 1, replica);
  if (rc != PEDN_OK) return rc;
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
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy2D(tf, 8, s->v.tf + (size_t)a * s->v.RS + replica, (size_t)s->v.RS * 8, 8, n, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

int pedn_set_width(pedn_sim* s, int32_t which, int32_t link, int32_t replica, double value) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (link < 0 || link >= s->v.L || which < 0 || which > 3) return fail(s, PEDN_E_ARG, "link or selector out of range");
  double* dst = which == PEDN_W_FRONT ? s->v.front : which == PEDN_W_BACK ? s->v.back : which == PEDN_W_SEP ? s->v.sepw : s->v.sepnp;
  int rc = push_rows(s, dst, &value, 1, (size_t)link, 1, replica);
  if (rc != PEDN_OK || which > PEDN_W_BACK) return rc;
  (which == PEDN_W_FRONT ? s->h_front_u : s->h_back_u)[link] = replica == PEDN_ALL ? value : __builtin_nan("");
  return push_uniform(s);
}

int pedn_set_widths(pedn_sim* s, int32_t which, const double* values) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (which < 0 || which > 3) return fail(s, PEDN_E_ARG, "selector out of range");
  DevView& v = s->v;
  if (v.L == 0) return PEDN_OK;
  double* dst = which == PEDN_W_FRONT ? v.front : which == PEDN_W_BACK ? v.back : which == PEDN_W_SEP ? v.sepw : v.sepnp;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  size_t bytes = (size_t)v.L * v.R * 8;
  int rc = ensure_stage(s, bytes);
  if (rc != PEDN_OK) return rc;
  HIP_TRY(s, hipMemcpyAsync(s->stage, values, bytes, hipMemcpyHostToDevice, s->stream));
  size_t n = (size_t)v.L * v.R;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, dst, (const double*)s->stage, v.L,
                     (size_t)0, (size_t)1, v.RS, 0, v.R, 1, v.R);
  HIP_TRY(s, hipGetLastError());
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

int pedn_get_widths(pedn_sim* s, int32_t which, double* values) {
  if (!s || !values) return fail(s, PEDN_E_ARG, "null argument");
  if (which < 0 || which > 3) return fail(s, PEDN_E_ARG, "selector out of range");
  DevView& v = s->v;
  if (v.L == 0) return PEDN_OK;
  const double* src = which == PEDN_W_FRONT ? v.front : which == PEDN_W_BACK ? v.back : which == PEDN_W_SEP ? v.sepw : v.sepnp;
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  HIP_TRY(s, hipMemcpy2D(values, (size_t)v.R * 8, src, (size_t)v.RS * 8, (size_t)v.R * 8, v.L, hipMemcpyDeviceToHost));
  return PEDN_OK;
}

static int launch_step(pedn_sim* s, int t) {
  DevView& v = s->v;
  const unsigned rgroups = (unsigned)(v.RS / 64);
  if (v.n_multi > 0) {
    size_t n = (size_t)v.n_multi * v.RS;
    if (v.pr) hipLaunchKernelGGL(turn_prob_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, t);
    else hipLaunchKernelGGL(turn_prob_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, t);
  }
  if (v.pr) hipLaunchKernelGGL(node_kernel<true>, dim3(rgroups, (unsigned)s->n_blocks), dim3(512), 0, s->stream, v, t);
  else hipLaunchKernelGGL(node_kernel<false>, dim3(rgroups, (unsigned)s->n_blocks), dim3(512), 0, s->stream, v, t);
  if (v.n_pairs_corr > 0) {
    size_t n = (size_t)v.n_pairs_corr * (v.pr ? v.RS : v.RS / 2);
    if (v.pr) hipLaunchKernelGGL(link_kernel_pr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, t);
    else hipLaunchKernelGGL(link_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, t);
  }
  return PEDN_OK;
}

int pedn_step(pedn_sim* s, int32_t t) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (t < 1 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  launch_step(s, t);
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_profile_step(pedn_sim* s, int32_t t, float ms[3]) {
  if (!s || !ms) return fail(s, PEDN_E_ARG, "null argument");
  if (t < 1 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  DevView& v = s->v;
  // hipExtLaunchKernelGGL start/stop events carry the dispatch's own begin/end timestamps (what rocprofv3 reports),
  // not the enqueue-to-completion interval an ordinary hipEventRecord bracket would measure.
  hipEvent_t ev[6];
  for (int i = 0; i < 6; ++i) HIP_TRY(s, hipEventCreate(&ev[i]));
  const unsigned rgroups = (unsigned)(v.RS / 64);
  if (v.n_multi > 0) {
    size_t n = (size_t)v.n_multi * v.RS;
    if (v.pr) hipExtLaunchKernelGGL(turn_prob_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, ev[0], ev[1], 0, v, t);
    else hipExtLaunchKernelGGL(turn_prob_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, ev[0], ev[1], 0, v, t);
  }
  if (v.pr) hipExtLaunchKernelGGL(node_kernel<true>, dim3(rgroups, (unsigned)s->n_blocks), dim3(512), 0, s->stream, ev[2], ev[3], 0, v, t);
  else hipExtLaunchKernelGGL(node_kernel<false>, dim3(rgroups, (unsigned)s->n_blocks), dim3(512), 0, s->stream, ev[2], ev[3], 0, v, t);
  if (v.n_pairs_corr > 0) {
    size_t n = (size_t)v.n_pairs_corr * (v.pr ? v.RS : v.RS / 2);
    if (v.pr) hipExtLaunchKernelGGL(link_kernel_pr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, ev[4], ev[5], 0, v, t);
    else hipExtLaunchKernelGGL(link_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, ev[4], ev[5], 0, v, t);
  }
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  ms[0] = ms[1] = ms[2] = 0.0f;
  if (v.n_multi > 0) HIP_TRY(s, hipEventElapsedTime(&ms[0], ev[0], ev[1]));
  HIP_TRY(s, hipEventElapsedTime(&ms[1], ev[2], ev[3]));
  if (v.n_pairs_corr > 0) HIP_TRY(s, hipEventElapsedTime(&ms[2], ev[4], ev[5]));
  for (int i = 0; i < 6; ++i) hipEventDestroy(ev[i]);
  return PEDN_OK;
}

int pedn_run(pedn_sim* s, int32_t t0, int32_t t1) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (t0 < 1 || t1 > s->v.T1 || t0 > t1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  HIP_TRY(s, hipSetDevice(s->device));
  for (int t = t0; t < t1; ++t) launch_step(s, t);
  HIP_TRY(s, hipGetLastError());
  return PEDN_OK;
}

int pedn_synchronize(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PEDN_OK;
}

int pedn_error_flags(pedn_sim* s, uint32_t* flags) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
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
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  size_t n = (size_t)(t1 - t0) * (c1 - c0) * (r1 - r0);
  size_t esz = field < 7 ? 8 : 4;
  int rc = ensure_stage(s, n * esz);
  if (rc != PEDN_OK) return rc;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (field < 7)
    hipLaunchKernelGGL(gather_kernel<double>, dim3(blocks), dim3(256), 0, s->stream, (const double*)v.f64[field], (double*)s->stage, t0,
                       t1 - t0, c0, c1 - c0, r0, r1 - r0, cols, v.RS);
  else
    hipLaunchKernelGGL(gather_kernel<float>, dim3(blocks), dim3(256), 0, s->stream, (const float*)v.f32[field - 7], (float*)s->stage, t0,
                       t1 - t0, c0, c1 - c0, r0, r1 - r0, cols, v.RS);
  HIP_TRY(s, hipGetLastError());
  HIP_TRY(s, hipMemcpyAsync(out, s->stage, n * esz, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PEDN_OK;
}

void* pedn_device_ptr(pedn_sim* s, int32_t field, int64_t* columns, int64_t* replica_stride) {
  if (!s || field < 0 || field >= PEDN_N_FIELDS) return nullptr;
  if (columns) *columns = field < 4 ? s->v.Lall : s->v.L;
  if (replica_stride) *replica_stride = s->v.RS;
  return field < 7 ? (void*)s->v.f64[field] : (void*)s->v.f32[field - 7];
}

void* pedn_stream(pedn_sim* s) { return s ? (void*)s->stream : nullptr; }

int pedn_timer_begin(pedn_sim* s) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipEventRecord(s->ev0, s->stream));
  return PEDN_OK;
}

int pedn_timer_end(pedn_sim* s, float* ms) {
  if (!s || !ms) return fail(s, PEDN_E_ARG, "null argument");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipEventRecord(s->ev1, s->stream));
  HIP_TRY(s, hipEventSynchronize(s->ev1));
  HIP_TRY(s, hipEventElapsedTime(ms, s->ev0, s->ev1));
  return PEDN_OK;
}

int pedn_set_link_params(pedn_sim* s, const double* kc, const double* kj, const double* vf, const int32_t* fft, const int32_t* tau_sw,
                         const float* tt0) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
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
  }
  int rc;
  const size_t n = (size_t)v.L * v.RS;
  if (!s->d_kc_r) {
    if ((rc = dalloc(s, n, &s->d_kc_r)) || (rc = dalloc(s, n, &s->d_kj_r)) || (rc = dalloc(s, n, &s->d_vf_r)) ||
        (rc = dalloc(s, n, &s->d_fft_r)) || (rc = dalloc(s, n, &s->d_tausw_r)) || (rc = dalloc(s, n, &s->d_tt0_r))) return rc;
    // padding lanes (replica >= R) must hold valid numbers too: start from a safe fill
    std::vector<double> one(n, 1.0), two(n, 2.0);
    HIP_TRY(s, hipMemcpy(s->d_kc_r, one.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(s, hipMemcpy(s->d_kj_r, two.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(s, hipMemcpy(s->d_vf_r, one.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<int32_t> big(n, 1 << 20);  // free_flow_tau far in the future: padding lanes stay idle
    HIP_TRY(s, hipMemcpy(s->d_fft_r, big.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> onei(n, 1);
    HIP_TRY(s, hipMemcpy(s->d_tausw_r, onei.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> onef(n, 1.0f);
    HIP_TRY(s, hipMemcpy(s->d_tt0_r, onef.data(), n * 4, hipMemcpyHostToDevice));
  }
  if ((rc = push_matrix(s, s->d_kc_r, kc, v.L)) || (rc = push_matrix(s, s->d_kj_r, kj, v.L)) || (rc = push_matrix(s, s->d_vf_r, vf, v.L)) ||
      (rc = push_matrix(s, s->d_fft_r, fft, v.L)) || (rc = push_matrix(s, s->d_tausw_r, tau_sw, v.L)) || (rc = push_matrix(s, s->d_tt0_r, tt0, v.L)))
    return rc;
  v.kc_r = s->d_kc_r; v.kj_r = s->d_kj_r; v.vf_r = s->d_vf_r; v.fft_r = s->d_fft_r; v.tausw_r = s->d_tausw_r; v.tt0_r = s->d_tt0_r;
  v.pr = 1;
  return PEDN_OK;
}

int pedn_set_od_weights_per_replica(pedn_sim* s, const double* w) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  HIP_TRY(s, hipSetDevice(s->device));
  HIP_TRY(s, hipStreamSynchronize(s->stream));
  DevView& v = s->v;
  if (!w) {
    v.pod_pr = 0;
    return PEDN_OK;
  }
  const int np = s->n_pair, nt = s->n_turns, R = v.R;
  if (np == 0) return PEDN_OK;
  // P(od | up) per replica (path_finder.py:599-615), same arithmetic as tabulate_pair_pod, one column per replica
  std::vector<double> pod((size_t)np * R), tab((size_t)std::max(nt, 1) * R, 0.0), upod(s->h_upod_od.size());
  for (int r = 0; r < R; ++r) {
    for (int u = 0; u < s->n_up; ++u) {
      const int a = s->h_up_od_ptr[u], b = s->h_up_od_ptr[u + 1];
      double tot = 0.0;
      for (int q = a; q < b; ++q) tot += w[(size_t)s->h_upod_od[q] * R + r];
      for (int q = a; q < b; ++q) upod[q] = tot > 0.0 ? w[(size_t)s->h_upod_od[q] * R + r] / tot : (b - a > 0 ? 1.0 / (double)(b - a) : 0.0);
    }
    for (int q = 0; q < np; ++q) pod[(size_t)q * R + r] = upod[s->h_pair_upod[q]];
    for (int tn = 0; tn < nt; ++tn) {
      if (!s->h_turn_mode[tn]) continue;
      double acc = 0.0;
      for (int q = s->h_turn_pair_ptr[tn]; q < s->h_turn_pair_ptr[tn + 1]; ++q) acc += 1.0 * pod[(size_t)q * R + r];
      tab[(size_t)tn * R + r] = acc;
    }
  }
  int rc;
  if (!s->d_pair_pod_r) {
    if ((rc = dalloc(s, (size_t)np * v.RS, &s->d_pair_pod_r)) || (rc = dalloc(s, (size_t)std::max(nt, 1) * v.RS, &s->d_turn_tab_r))) return rc;
    HIP_TRY(s, hipMemset(s->d_pair_pod_r, 0, (size_t)np * v.RS * 8));
    HIP_TRY(s, hipMemset(s->d_turn_tab_r, 0, (size_t)std::max(nt, 1) * v.RS * 8));
  }
  if ((rc = push_matrix(s, s->d_pair_pod_r, pod.data(), np)) || (rc = push_matrix(s, s->d_turn_tab_r, tab.data(), nt))) return rc;
  v.pair_pod_r = s->d_pair_pod_r; v.turn_tab_r = s->d_turn_tab_r;
  v.pod_pr = 1;
  return PEDN_OK;
}

int pedn_rl_configure(pedn_sim* s, const pedn_rl_desc* d, int32_t* n_actions, int32_t* n_obs) {
  if (!s || !d) return fail(s, PEDN_E_ARG, "null argument");
  if (d->n_agents < 1) return fail(s, PEDN_E_ARG, "no agents");
  if (d->obs_mode < 1 || d->obs_mode > 5) return fail(s, PEDN_E_ARG, "obs_mode must be 1..5");
  HIP_TRY(s, hipSetDevice(s->device));
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
  if ((rc = dalloc(s, (size_t)v.R * O, &q.obs)) != PEDN_OK) return rc;
  if ((rc = dalloc(s, (size_t)v.R * d->n_agents, &q.rew)) != PEDN_OK) return rc;
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
  s->rl_ready = true;
  if (n_actions) *n_actions = A;
  if (n_obs) *n_obs = O;
  return PEDN_OK;
}

int pedn_rl_apply_actions(pedn_sim* s, const double* actions, int32_t on_device) {
  if (!s || !actions) return fail(s, PEDN_E_ARG, "null argument");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  HIP_TRY(s, hipSetDevice(s->device));
  DevView& v = s->v;
  RlView& q = s->rl;
  const size_t bytes = (size_t)v.R * q.A * sizeof(double);
  RlView qq = q;
  if (on_device) qq.actions = const_cast<double*>(actions);  // read the caller's rows in place: no staging copy
  else HIP_TRY(s, hipMemcpyAsync(q.actions, actions, bytes, hipMemcpyHostToDevice, s->stream));
  size_t n = (size_t)q.A * v.RS;
  hipLaunchKernelGGL(rl_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, v, qq);
  HIP_TRY(s, hipGetLastError());
  if (!on_device) HIP_TRY(s, hipStreamSynchronize(s->stream));  // the host buffer is borrowed for the call only
  return PEDN_OK;
}

int pedn_rl_observe(pedn_sim* s, int32_t t, int32_t accumulate, float* obs, float* rewards) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (!s->rl_ready) return fail(s, PEDN_E_ARG, "pedn_rl_configure has not been called");
  if (t < 0 || t > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "time step outside 0..T");
  HIP_TRY(s, hipSetDevice(s->device));
  DevView& v = s->v;
  RlView& q = s->rl;
  hipLaunchKernelGGL(rl_observe_kernel, dim3((unsigned)(v.RS / 64), (unsigned)q.n_agents), dim3(512), 0, s->stream, v, q, t, accumulate);
  HIP_TRY(s, hipGetLastError());
  if (obs) HIP_TRY(s, hipMemcpyAsync(obs, q.obs, (size_t)v.R * q.O * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  if (rewards) HIP_TRY(s, hipMemcpyAsync(rewards, q.rew, (size_t)v.R * q.n_agents * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  if (obs || rewards) HIP_TRY(s, hipStreamSynchronize(s->stream));
  return PEDN_OK;
}

int pedn_rl_step(pedn_sim* s, const double* actions, int32_t on_device, int32_t t, int32_t action_gap, float* obs, float* rewards) {
  if (!s) return fail(nullptr, PEDN_E_ARG, "null handle");
  if (action_gap < 1 || t < 1 || t + action_gap - 1 > s->v.T1 - 1) return fail(s, PEDN_E_ARG, "step range outside 1..T");
  int rc = PEDN_OK;
  if (actions && (rc = pedn_rl_apply_actions(s, actions, on_device)) != PEDN_OK) return rc;
  for (int k = 0; k < action_gap; ++k) {
    launch_step(s, t + k);
    const bool last = k == action_gap - 1;
    if ((rc = pedn_rl_observe(s, t + k, k > 0, last ? obs : nullptr, last ? rewards : nullptr)) != PEDN_OK) return rc;
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
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(nullptr, hipMalloc((void**)&da, (size_t)n * 8));
  HIP_TRY(nullptr, hipMalloc((void**)&dout, (size_t)n * 8));
  HIP_TRY(nullptr, hipMemcpy(da, a, (size_t)n * 8, hipMemcpyHostToDevice));
  if (b) {
    HIP_TRY(nullptr, hipMalloc((void**)&db, (size_t)n * 8));
    HIP_TRY(nullptr, hipMemcpy(db, b, (size_t)n * 8, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(device_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, n, (const double*)da, (const double*)db,
                     (uint32_t)seed, (uint32_t)(seed >> 32), dout);
  HIP_TRY(nullptr, hipDeviceSynchronize());
  HIP_TRY(nullptr, hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
  hipFree(da);
  hipFree(dout);
  if (db) hipFree(db);
  return PEDN_OK;
}

}  // extern "C"
