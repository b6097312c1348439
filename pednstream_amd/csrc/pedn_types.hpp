// pedn_types.hpp -- device-visible data structures of the HIP engine: the per-link parameter record, the scalar records
// one wave reads, and the view of all HBM arrays that every kernel receives by value.  See DESIGN.md section 4 for the layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pedn.h"

// ------------------------------------------------------------------------------------------------- device view
enum { F_IN = 0, F_OUT, F_CI, F_CO, F_S, F_R, F_GATE };
enum { G_TT = 0, G_ATT, G_N, G_K, G_V, G_LF };

struct LinkP {  // static per-link parameters, wave-uniform on the device
  double length, width, vf, kc, kj, gamma, act, bi, noise;
  double kjA;                                  // kj * (length * width): jam capacity of an unseparated link (link.py:386)
  float tt0;
  float kc32, kj32, dk32, area32, gamma32;     // (float)kc, (float)kj, (float)(kj - kc), (float)(length * width), (float)gamma:
                                               // the float32 casts the formulas take of wave-uniform doubles, made once on the host
  int32_t rev, sep, fd, tau_sw, fft;
  int32_t pad;
  __host__ __device__ void derive() {
    kjA = kj * (length * width);
    kc32 = (float)kc; kj32 = (float)kj; dk32 = (float)(kj - kc); area32 = (float)(length * width); gamma32 = (float)gamma;
  }
};

// Per-replica scenario parameters of one (link, replica): ONE 32-byte record -- two 16-byte loads per lane -- instead of six arrays
// (randomised ensembles / RL resets, env_loader.py:363-424).  The look-backs fit 16 bits (pedn_set_link_params checks).
struct alignas(16) LinkPR {
  double kc, kj, vf;      // k_critical, k_jam, free_flow_speed
  float tt0;              // travel_time[0] (link.py:83)
  int16_t fft, tau_sw;    // free_flow_tau (link.py:86), shock-wave look-back (link.py:380)
};
static_assert(sizeof(LinkPR) == 32, "LinkPR must stay two 16-byte loads");

struct SlotRec {  // one wave of node_kernel: a (node, slot) with everything static it needs, fetched by ONE scalar load burst
  // dyn: where the slot's row of turning fractions comes from -- 0 tf / tf_u (default or imposed), 1 tfd[t & 1] (turn_frac_kernel),
  // 2 turn_tab[t] / turn_tab_r (every product constant: replica-independent, tabulated and renormalised on the host)
  // act: action slot of the batched RL step that sets this slot's gate (back gate of lout = front gate of lin), -1 none
  // lp: index of the node among those that solve the node LP (assign_flows_type 'optimal'), on the node's slot 0, else -1
  // trow: the slot's row record in trow_words when the row's turning fractions are computed on the device (dyn == 1), else -1
  int32_t node, slot, base, m, kind, dyn, lin, lout, turn0, demand_row, act, lp, trow, pad1;
  LinkP Pin, Pout;  // parameters of the incoming / outgoing link of the slot (unused for a virtual pair)
};

struct CorrRec {  // one lane group of link_kernel: both directions of a corridor
  int32_t a, b, pad0, pad1;
  LinkP Pa, Pb;
};

// ---- route choice (path_finder.py:561-737), organised per row (= incoming slot) of a dynamic node: one wave of
// turn_frac_kernel owns (row, 64 replicas).  Its static data are int32 word records fetched with vector loads (lane k holds
// word k) and broadcast with v_readlane -- see turn_frac_body.
#define PEDN_TF_LDS_ROWS 64    // LDS rows (64 lanes x 8 bytes) of one workgroup = 4 rows of dynamic nodes, shared out by the host
#define PEDN_TROW_WORDS 128
#define PEDN_TF_INL_ROWS 8     // LDS rows a slot wave of node_kernel<.., TF> has for the probabilities of its own row (single-launch plan)
#define PEDN_TF_COOP_GROUPS 8   // rows with more multi-entry groups get a workgroup of their own (see turn_frac_body)
// row record, PEDN_TROW_WORDS words (m = -1: padding):
//   [0] m  [1] first turn of the row  [2] first group (index into tgrp_words / 32)  [3] number of multi-entry groups
//   [4] Q0  [5] Q1: the row's (turn, od) products  [6] n_used outgoing links below  [7] some of them is a separator
//   entries e = 0..4 at 8 + 10 e, e = 5, 6 at 64 + 10 (e - 5): link, reverse link, separator flag, float32(length * width),
//       free-flow speed (2 words), k_critical (2), length (2); entries >= n_used repeat entry 0
//   turns jj = 0..6 at 84 + 3 jj: q0, q1 (absolute product indices), mode (1: every product is the constant 1 -> the raw
//       fraction is tabulated per step on the host)
//   [107] 1: some probability of the row lives in ent_p instead of LDS
//   [108] 1: coop -- the four records of this workgroup are the same row; its groups and turns are shared out over the waves
//   [109] first LDS row of this row inside its workgroup (the inline variant rebases to it)
// group record, 32 words (softmax group (od, up) with more than one downstream, update_node_turn_probs :561-589):
//   [0] n  [1] allphys  [2 + e] entry of the row record the downstream link is (-1: virtual link, density 0, capacity 100)
//   [9 + e] where P(down | up, od) goes: < 0 nowhere, < PEDN_TF_LDS_ROWS that LDS row of the workgroup, else row
//   (value - PEDN_TF_LDS_ROWS) of ent_p   [16 + 2 e] alpha * distance / (sum of the group's distances + 1e-6) (2 words)

struct DevView {
  double* f64[7];
  float* f32[6];
  float* rsum;
  double *front, *back, *sepw, *sepnp, *tf, *demand, *ent_p;
  double* tfd[2];  // [n_turns][RS] x 2: fractions of dynamic nodes; step t reads tfd[t & 1], the fractions of t + 1 are
                   // written into the other buffer while the host can still read those of t
  // replica-uniform shortcuts (NaN = the value differs between replicas, read the per-replica row instead): a value every
  // replica shares is one scalar load per wave instead of 8 bytes per lane
  const double *front_u, *back_u;  // [L]
  const double* tf_u;              // [n_turns]
  // per-replica scenario parameters (randomised ensembles / RL resets, env_loader.py:363-424); used when pr != 0
  const LinkPR* prm;                  // [L][RS] k_critical, k_jam, free_flow_speed, free_flow_tau, shock-wave look-back, travel_time[0]
  const double *pair_pod_r, *turn_tab_r;  // [n_pair][RS], [n_turns][RS]: P(od | up) with per-replica, time-constant OD weights
  int32_t pr, pod_pr;
  const double* od_w;
  uint32_t* flags;
  const LinkP* lp;
  const SlotRec* slot_rec;
  const CorrRec* corr_rec;
  const int32_t *node_kind, *node_slot_ptr, *node_turn_ptr, *node_demand_row, *node_dyn, *slot_in, *slot_out;
  const int32_t* pair_row;      // [n_pair] where the product's P(down | up, od) is: -1 the constant 1 (single-entry group), else as word 9 + e of a group record
  const int32_t* trow_words;    // [n_trow][PEDN_TROW_WORDS] rows of dynamic nodes, heaviest first
  const int32_t* tgrp_words;    // [n_multi + 8][32] groups with more than one downstream entry (single-entry groups have P = 1 exactly)
  const double* turn_tab;       // [T+1][n_turns] tabulated raw fractions of such turns
  const double* pair_pod;  // [T+1][n_pair] P(od | up) of the pair's upstream group, replica independent
  int32_t L, Lall, T1, RS, R, W, n_grp, n_multi, n_trow, n_pairs_corr, n_pair, n_turns;
  double dt, pf_temp, pf_alpha, pf_beta, pf_omega, pf_eps;
  uint32_t k0, k1, replica_offset;
  int32_t meanfield;
  int32_t tf_general;  // diagnostics (PEDN_TF_GENERAL): 1 the any-size softmax path, 2 the general row-sum path of turn_frac_body
  // history rows: time index t lives in row (t & mask).  Full-record mode: every mask is 0x7fffffff (row = t, T+1 rows, the
  // reference's own footprint).  Recent-history mode (hist = 1): inflow and cumulative_inflow keep all rows (the sending
  // flow looks back an unbounded, data-dependent number of steps into them), the others are rings of mask + 1 rows.
  int32_t m64[7], m32[6];
  int32_t hist;
  // Lazy reset (pedn_reset_lazy, full-record mode): rows above valid_hi have neither been written nor cleared since the last reset and
  // hold the previous episode's values.  The only reads that can land there are the wrapped (negative) indices of get_outflow's
  // look-backs (link.py:205-212: Python's inflow[-k] = the untouched tail = 0): answered with the initial value instead of the row.
  // 0x7fffffff after an ordinary reset.
  int32_t valid_hi;
  // Device-resident step clock (pedn_rl_clock_begin / pedn_rl_step_clocked): a launch whose step argument is negative takes its step --
  // and the node kernel its valid_hi -- from here, see node_clock in pedn_kernels.hpp.  [0] node_kernel's step, [1] the step of the
  // launch behind it, [2] valid_hi.
  int32_t* clock;
  int32_t tf_lds_off;   // node_kernel<.., TF>: first double of the waves' private LDS rows for their own turning fractions (PEDN_TF_INL_ROWS each)
  int32_t pairs_adj;  // 1: corridor p is the links (2p, 2p + 1) -- the reference creates the two directions of an edge one after the other --
                      // so the link update forms its row addresses from p alone, without waiting for the corridor's record
  int32_t sub0, subRS;  // replicas [sub0, sub0 + subRS) are this launch's share (the whole batch, or one half of it per stream)
  // gater actions applied inside node_kernel (pedn_rl_step, gater-only agent sets): row-major [R][rl_A] widths, NaN = no action
  const double* rl_actions;
  int32_t rl_A;
  double rl_max_delta_gate;
  // node LP (PEDN_NODE_OPTIMAL): per (LP node, replica group) a tableau of lp_stride doubles x 64 lanes and lp_bstride basis ids
  double* lp_ws;
  int32_t* lp_basis;
  size_t lp_stride, lp_bstride;
};
