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

struct TurnRec {  // one turn of a dynamic node: its (turn, od) products and how its fraction is obtained
  int32_t q0, q1;  // products [q0, q1) in the reference's summation order
  int32_t mode;    // 1: every product is constant -> the raw fraction is tabulated per step on the host
  int32_t pad;
};

struct GrpEnt {  // one downstream entry of a softmax group (update_node_turn_probs, path_finder.py:561-589)
  int32_t link, rev, sep;  // outgoing link (-1: virtual), its reverse, separator flag
  int32_t pair;            // the (turn, od) product that consumes this entry's probability (-1: none)
  float area32;            // float32(length * width) of a plain link
  int32_t pad;
  double vf, kc;           // for the capacity fallback back_gate * v_f * k_c * dt (:576)
  double dist_term;        // alpha * distance / (sum of the group's distances + 1e-6), static (:582)
  double length;           // of the link (density of a separator = pedestrians / (length * separator width))
};

struct GrpRec {  // one softmax group with more than one downstream: everything static a lane of turn_prob_kernel needs,
  int32_t n, allphys, pad0, pad1;  // fetched by one scalar load burst at an address that depends on blockIdx only
  GrpEnt e[PEDN_MAX_DEGREE - 1];
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
  const TurnRec* turn_rec;      // [n_turns]
  const int32_t* pair_row;      // [n_pair] row of ent_p that holds the product's probability; constant products share the row of ones
  const GrpRec* grp_rec;        // [n_multi] groups with more than one downstream entry (single-entry groups have P = 1 exactly)
  const double* turn_tab;       // [T+1][n_turns] tabulated raw fractions of such turns
  const double* pair_pod;  // [T+1][n_pair] P(od | up) of the pair's upstream group, replica independent
  int32_t L, Lall, T1, RS, R, W, n_grp, n_multi, n_pairs_corr, n_pair, n_turns;
  double dt, pf_temp, pf_alpha, pf_beta, pf_omega, pf_eps;
  uint32_t k0, k1, replica_offset;
  int32_t meanfield;
};
