/* pedn.h -- C-ABI of the MI355X engine for PedNStream's per-timestep network_loading hot path.
 *
 * The reference is pure Python and has no FFI; this header is the boundary a maintainer binds with ctypes
 * (see INTEGRATION.md).  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree):
 *
 *   pedn_create                  Network.__init__ / init_nodes_and_links          src/LTM/network.py:56-121,194-248
 *                                Link.__init__ / Separator.__init__ state arrays   src/LTM/link.py:12-17,32-100,420-425
 *   pedn_step / pedn_run         Network.network_loading(t)                        src/LTM/network.py:266-287
 *                                (Node.assign_flows src/LTM/node.py:164-221, Link.cal_sending_flow src/LTM/link.py:216-370,
 *                                 Link.cal_receiving_flow[_with_reverse] :372-416, Network.update_link_states
 *                                 src/LTM/network.py:257-264, PathFinder.calculate_node_turning_fractions
 *                                 src/LTM/path_finder.py:717-737)
 *   pedn_set_demand              node.demand (origin arrays)                       src/LTM/network.py:130-139, node.py:176
 *   pedn_set_od_weights          ODManager.od_flows / get_od_flow                  src/LTM/od_manager.py:22-54
 *   pedn_set_turning_fractions   Network.update_turning_fractions_per_node         src/LTM/network.py:250-255
 *   pedn_get_turning_fractions   node.turning_fractions                            src/LTM/node.py:11
 *   pedn_set_width[s]            Link.front/back_gate_width, Separator.separator_width setters
 *                                                                                   src/LTM/link.py:110-126,462-478
 *   pedn_read                    the per-link history arrays read by callers       src/LTM/link.py:12-17,56,82-97
 *   pedn_error_flags             raise ValueError / Warning sites                  src/LTM/link.py:345-346,365-366; node.py:192-194,218-219,237-238
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a negative PEDN_E_* code on
 * failure, with a message retrievable through pedn_last_error().  Host pointers are borrowed for the duration of the
 * call.  All device work of one handle is enqueued on one HIP stream; pedn_step/pedn_run are asynchronous, the
 * read/flag calls synchronise.  Replicas are independent copies of the scenario stepped together; `replica` arguments
 * accept PEDN_ALL (-1) for "every replica".
 */
#ifndef PEDN_H
#define PEDN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEDN_ABI_VERSION 4
#define PEDN_ALL (-1)
#define PEDN_MAX_DEGREE 8 /* incident corridor slots per node handled by the node kernel */

/* return codes */
#define PEDN_OK 0
#define PEDN_E_ARG (-1)     /* bad argument / inconsistent model description */
#define PEDN_E_DEVICE (-2)  /* HIP runtime error */
#define PEDN_E_MODEL (-3)   /* sticky per-replica model error, see pedn_error_flags */
#define PEDN_E_NOMEM (-4)

/* per-replica sticky error bits (pedn_error_flags) */
#define PEDN_F_NEG_SENDING 1u /* negative sending flow            (link.py:345-346,365-366 ValueError) */
#define PEDN_F_NEG_FLOW 2u    /* negative s / r / q at a node      (node.py:192-194,218-219,237-238 Warning) */
#define PEDN_F_INDEX 4u       /* history index outside [-(T+1), T] (IndexError in the reference) */
#define PEDN_F_NEG_BINOM 8u   /* binomial with n < 0               (numpy ValueError at link.py:382) */
#define PEDN_F_SAME_STEP 16u  /* look-back of 0 steps: the reference result depends on node iteration order */
#define PEDN_F_LP 32u         /* node LP (assign_flows_type 'optimal') did not terminate: `res.success` false, node.py:267 */
/* (bit 64u is reserved: it belonged to the persistent plan of ABI 3, removed as a measured negative) */

/* RNG modes (oracle/rng_contract.py) */
#define PEDN_RNG_PHILOX 0
#define PEDN_RNG_MEANFIELD 1 /* binomial -> floor(n*p), normal -> 0 */

#define PEDN_NODE_CLASSIC 0
#define PEDN_NODE_OPTIMAL 1
#define PEDN_LP_PENALTY 1e-2 /* Node.w, node.py:14 */
#define PEDN_HIST_FULL 0
#define PEDN_HIST_RECENT 1

/* width selectors */
#define PEDN_W_FRONT 0
#define PEDN_W_BACK 1
#define PEDN_W_SEP 2
/* 0/1 per (separator link, replica): the separator width is held as a numpy float64 scalar in the reference (it is after
 * ActionApplier's np.clip, rl/builders.py:295), which makes `num_pedestrians / area` (link.py:136) a binary64 division
 * rounded to float32 instead of a float32 division.  pedn_rl_apply_actions sets it; the plain setters clear it. */
#define PEDN_W_SEP_NUMPY 3

/* history fields; f64 fields 0..6 have n_links + n_vlinks columns for ids 0..3 (virtual links keep only those) */
enum {
  PEDN_FIELD_INFLOW = 0,
  PEDN_FIELD_OUTFLOW = 1,
  PEDN_FIELD_CUM_INFLOW = 2,
  PEDN_FIELD_CUM_OUTFLOW = 3,
  PEDN_FIELD_SENDING = 4,
  PEDN_FIELD_RECEIVING = 5,
  PEDN_FIELD_GATE_REC = 6, /* back_gate_width_data (separator_width_data for separators) */
  PEDN_FIELD_TRAVEL_TIME = 7,
  PEDN_FIELD_AVG_TRAVEL_TIME = 8,
  PEDN_FIELD_NUM_PED = 9,
  PEDN_FIELD_DENSITY = 10,
  PEDN_FIELD_SPEED = 11,
  PEDN_FIELD_LINK_FLOW = 12,
  PEDN_N_FIELDS = 13
};

/* Static, replica-independent description of one scenario (built by pednstream_amd/flatten.py).
 * Node n owns slots node_slot_ptr[n] .. node_slot_ptr[n+1]-1; slot k of a node holds the incoming link
 * slot_in_link[] and the outgoing link slot_out_link[] towards the same neighbour (reverse pair); a link index
 * >= n_links denotes a virtual link (origin/destination side, always slot 0 of its node).
 * Turning fractions of node n are node_turn_ptr[n] .. +m(m-1), source-major, destinations ascending skipping the
 * source's own slot (node.py:274-280). */
typedef struct pedn_model_desc {
  int32_t abi_version;
  int32_t n_nodes, n_links, n_vlinks, n_turns, n_demand, n_od;
  int32_t T;      /* simulation_steps; histories have T+1 entries */
  int32_t window; /* moving-average window W = round(100/dt)  (link.py:89) */
  double dt;      /* unit_time */

  const int32_t* node_kind;       /* [n_nodes] 0 one-to-one, 1 regular                 (network.py:141-167) */
  const int32_t* node_slot_ptr;   /* [n_nodes+1] */
  const int32_t* node_turn_ptr;   /* [n_nodes+1] */
  const int32_t* node_demand_row; /* [n_nodes] row of `demand`, or -1 */
  const int32_t* node_dyn;        /* [n_nodes] 1: turning fractions recomputed every step (path_finder.py:731-737) */
  const int32_t* slot_in_link;    /* [n_slots] */
  const int32_t* slot_out_link;   /* [n_slots] */

  const int32_t* link_rev;    /* [n_links] reverse link */
  const int32_t* link_sep;    /* [n_links] 1: Separator */
  const int32_t* link_fd;     /* [n_links] 0 yperman, 1 greenshields, 2 smulders */
  const int32_t* link_tau_sw; /* [n_links] round(length/(shockwave_speed*dt))  (link.py:380) */
  const int32_t* link_fft;    /* [n_links] free_flow_tau                        (link.py:86) */
  const float* link_tt0;      /* [n_links] travel_time[0]                       (link.py:83) */
  const double *link_length, *link_width, *link_vf, *link_kc, *link_kj, *link_gamma, *link_act, *link_bi, *link_noise;
  const double *front_gate0, *back_gate0, *sep_width0; /* [n_links] initial widths */

  const double* tf_init; /* [n_turns] */
  const double* demand;  /* [n_demand][T+1] */
  const double* od_w;    /* [n_od][T+1] */

  /* route-choice tables (all index ranges are global, CSR per node) */
  double pf_temp, pf_alpha, pf_beta, pf_omega, pf_eps;
  int32_t n_up, n_upod, n_grp, n_ent, n_pair;
  const int32_t* node_up_ptr;   /* [n_nodes+1] -> upstream groups of the node              (up_od_probs, :599-615) */
  const int32_t* up_slot;       /* [n_up]      incoming slot (local index) the upstream group belongs to */
  const int32_t* up_od_ptr;     /* [n_up+1]    -> upod entries of one upstream */
  const int32_t* upod_od;       /* [n_upod]    OD row in od_w */
  const int32_t* node_grp_ptr;  /* [n_nodes+1] -> (od, up) softmax groups of the node       (turns_distances, :563) */
  const int32_t* grp_ent_ptr;   /* [n_grp+1]   -> downstream entries of one group */
  const int32_t* grp_allphys;   /* [n_grp]     1: every downstream is a physical link (f32 density branch, :581) */
  const int32_t* grp_node;      /* [n_grp]     node index the group belongs to */
  const int32_t* ent_link;      /* [n_ent]     outgoing link of the entry, -1 = virtual     (:577-579) */
  const double* ent_dist;       /* [n_ent]     remaining distance */
  const int32_t* turn_pair_ptr; /* [n_turns+1] -> (entry, upod) products summed into one turning fraction (:668-686) */
  const int32_t* pair_ent;      /* [n_pair] */
  const int32_t* pair_upod;     /* [n_pair] */

  /* PEDN_HIST_FULL: all 13 arrays keep their T+1 entries like the reference's (80 B per link, time index and replica).
   * PEDN_HIST_RECENT: only what the recurrence itself looks far back into is kept whole -- inflow and cumulative_inflow
   * (link.py:199-214,284-288: data-dependent look-back); cumulative_outflow keeps max(tau_shockwave) + 2 entries, travel_time
   * the moving-average window + 2, everything else the last 4.  Same numbers step for step (the batched RL environment
   * reads nothing older); pedn_read of an entry that has left its ring fails, and so does one of an entry not written yet whose
   * ring slot still holds an older one (sending_flow / receiving_flow of step t are entries t-1, node.py:206, link.py:367: after
   * step t their newest entry is t-1).  16 B + a few rows instead of 80 B. */
  int32_t history_mode;

  /* PEDN_NODE_CLASSIC: RegularNode.solve('classic') (node.py:272-300).  PEDN_NODE_OPTIMAL: assign_flows_type 'optimal', the
   * linear programme of node.py:249-271 (maximise the total flow minus 0.01 x the deviation from the turning fractions under
   * the sending / receiving constraints) solved per (node, replica) by a dense primal simplex with Bland's rule.  The
   * reference hands the same programme to scipy/HiGHS; the optimum is degenerate, so the two agree in the objective value,
   * not necessarily in the vertex: parity for this mode is objective-level only (unpinned). */
  int32_t node_model;

  /* [n_nodes] or NULL: measured cost of every node's slowest slot wave (any unit; ticks from kernel entry to the node kernel's first
   * block barrier, tools/pack_calibrate.py -> <scenario>/pack_cost.json).  Nodes are binned into the node kernel's blocks of eight
   * waves; a block lasts as long as its slowest wave, so nodes of similar cost should share a block.  Without it the bins are packed
   * by a static estimate (the number of (turn, od) products of a node). */
  const float* node_cost;
} pedn_model_desc;

typedef struct pedn_sim pedn_sim;

int pedn_abi_version(void);
/* message of the last failing call on this thread (handle may be NULL for pedn_create failures) */
const char* pedn_last_error(const pedn_sim* sim);

int pedn_create(const pedn_model_desc* model, int32_t n_replicas, int32_t replica_offset, uint64_t seed,
                int32_t rng_mode, int32_t device, pedn_sim** out);
int pedn_destroy(pedn_sim* sim);

/* values[n] -> demand[node][0..n-1] (rest zero); node is the model's node index, replica may be PEDN_ALL */
int pedn_set_demand(pedn_sim* sim, int32_t node, int32_t replica, const double* values, int32_t n);
/* demand[node][0..n-1] of one replica back to the host (what pedn_set_demand* / pedn_draw_demand left on the device) */
int pedn_get_demand(pedn_sim* sim, int32_t node, int32_t replica, double* values, int32_t n);
/* the same for every replica in one call: values[r * n + i] = demand of replica r at time index i (n <= T+1, rest zero) */
int pedn_set_demand_matrix(pedn_sim* sim, int32_t node, const double* values, int32_t n);
/* the same for a subset: values[k * n + i] = demand of replica replicas[k] at time index i, k < n_rep; one upload */
int pedn_set_demand_rows(pedn_sim* sim, int32_t node, const int32_t* replicas, int32_t n_rep, const double* values, int32_t n);
/* values[T+1] -> OD weight row `od` (shared by all replicas) */
int pedn_set_od_weights(pedn_sim* sim, int32_t od, const double* values, int32_t n);
int pedn_set_turning_fractions(pedn_sim* sim, int32_t node, int32_t replica, const double* tf, int32_t n);
int pedn_get_turning_fractions(pedn_sim* sim, int32_t node, int32_t replica, double* tf, int32_t n);
int pedn_set_width(pedn_sim* sim, int32_t which, int32_t link, int32_t replica, double value);
/* values[n_links][n_replicas] */
int pedn_set_widths(pedn_sim* sim, int32_t which, const double* values);
/* every replica back to the same widths (an episode reset): front[n_links], back[n_links], sep[n_links] broadcast, the
 * "separator width is an np.float64" marks (PEDN_W_SEP_NUMPY) cleared */
int pedn_reset_widths(pedn_sim* sim, const double* front, const double* back, const double* sep);
/* current widths -> values[n_links][n_replicas] (the device may have changed them through pedn_rl_apply_actions) */
int pedn_get_widths(pedn_sim* sim, int32_t which, double* values);

/* one step t in 1..T for every replica; asynchronous.  (The reference's loops run t = 1 .. T-1, network.py:266-287 callers; its
 * arrays have T+1 entries like these, so t = T is a valid step -- the batched RL env takes it.  Behind step T no turning fractions
 * of T+1 are prepared: the per-step tables end at T.) */
int pedn_step(pedn_sim* sim, int32_t t);
/* steps t0 .. t1-1 enqueued back to back, t1 <= T+1; asynchronous.  Same results as t1 - t0 calls of pedn_step, under the launch
 * plan the engine picked for the model (pedn_plan_info): for a model without device-computed turning fractions the slot waves of
 * step t + 1's node kernel perform Network.update_link_states(t) (network.py:257-264) themselves -- ONE launch per step; small batches
 * of models with short dynamic rows step the same way, the slot waves computing their own rows of turning fractions too.  The link
 * update of the last step launched stays pending until the next step or the first call that looks at or changes the state. */
int pedn_run(pedn_sim* sim, int32_t t0, int32_t t1);
int pedn_synchronize(pedn_sim* sim);
/* synchronises; flags[n_replicas] may be NULL; returns the OR over replicas (>= 0) or a negative code */
int pedn_error_flags(pedn_sim* sim, uint32_t* flags);

/* out[(t1-t0)][(link1-link0)][(rep1-rep0)] of the field's element type (f64 for ids 0..6, f32 otherwise) */
int pedn_read(pedn_sim* sim, int32_t field, int32_t t0, int32_t t1, int32_t link0, int32_t link1, int32_t rep0,
              int32_t rep1, void* out);

/* number of time indices field `field` keeps: T+1, or the size of its ring in PEDN_HIST_RECENT mode (time index t lives in
 * row t mod that size) */
int pedn_history_rows(pedn_sim* sim, int32_t field);
/* Everything the engine still owes the histories is enqueued on pedn_stream(): a link update left pending by the last pedn_step /
 * pedn_run (the next step's node kernel would have performed it), chains that are still forked, the rows a lazy reset declared
 * unwritten (cleared now).  Asynchronous.  After it, work ordered behind pedn_stream() sees every row of every field complete. */
int pedn_flush(pedn_sim* sim);
/* zero-copy access for on-device consumers: HBM base pointer of a history field laid out
 * [pedn_history_rows][columns][replica_stride]; columns/replica_stride may be NULL.  The call itself performs pedn_flush, so the
 * view is complete for a consumer that orders itself behind pedn_stream() NOW.  The pointer stays valid for the handle's life, but
 * what it shows is complete only up to the last pedn_flush / pedn_device_ptr / pedn_synchronize: after a later pedn_step / pedn_run
 * (last step's density / speed / travel time / num_pedestrians rows pending) or pedn_reset_lazy (rows above the current step hold the
 * previous episode) a consumer that kept the pointer calls pedn_flush before it orders itself behind the stream again. */
void* pedn_device_ptr(pedn_sim* sim, int32_t field, int64_t* columns, int64_t* replica_stride);
/* hipStream_t the engine launches on (a pure getter: it enqueues nothing -- see pedn_flush) */
void* pedn_stream(pedn_sim* sim);

/* HIP-event timing on the engine's stream: begin records an event, end records + synchronises and returns the
 * elapsed milliseconds in *ms */
int pedn_timer_begin(pedn_sim* sim);
int pedn_timer_end(pedn_sim* sim, float* ms);

/* One step with every kernel bracketed by HIP events on the engine's stream (synchronises).  ms[0] = turning-probability
 * kernel, ms[1] = node kernel, ms[2] = link kernel (0 when a kernel is not launched for this scenario). */
int pedn_profile_step(pedn_sim* sim, int32_t t, float ms[3]);

/* Steps t0 <= t < t1 under the launch plan pedn_run uses for such a range, every launch bracketed by its own dispatch
 * timestamps (synchronises).  ms[0..2] = mean duration of a launch of {stand-alone turning fractions, node kernel, the launch
 * behind it (link update and / or turning fractions of t + 1)}; *chains = 1 or 2: how many chains of launches ran side by
 * side (2: the two halves of the replicas on two streams -- a launch then covers n_replicas / 2 and overlaps the other
 * chain's launches, so per-launch bandwidths of the two chains add up). */
int pedn_profile_run(pedn_sim* sim, int32_t t0, int32_t t1, float ms[3], int32_t* chains);
/* The same range, not averaged: out[5 * row + {0..4}] = {step, chain (0 / 1), kind (0 stand-alone turning fractions, 1 node kernel,
 * 2 the launch behind it), start, end} with start / end in milliseconds after the start of the range's first launch; at most
 * `capacity` rows (3 per step and chain), *n_rows of them written.  Shows how the launches of the two chains overlap. */
int pedn_profile_timeline(pedn_sim* sim, int32_t t0, int32_t t1, float* out, int32_t capacity, int32_t* n_rows, int32_t* chains);

/* launch plan of pedn_run for long ranges: 1 = one chain of launches on the engine's stream, 2 = the halves of the replicas as two
 * chains on two streams (the default from 640 replicas -- the halves are whole 128-replica segments, 640 = 384 + 256; replicas are independent, results are the same).  The two streams are probed
 * to run side by side (pedn_plan_info); when the runtime cannot provide two independent queues the plan falls back to one chain. */
int pedn_set_streams(pedn_sim* sim, int32_t n);
/* The launch plan of pedn_run: info[0] = chains (1 | 2), info[1] = 1 when the link update is performed by the next step's node kernel
 * (one launch per step), info[2] = streams created until one was found that overlaps with the engine's stream (0: not probed yet;
 * the runtime may map two streams onto one hardware queue, which would serialise the chains), info[3] = duration in microseconds of
 * the probe's two concurrent 300 us kernels on the pair kept (~300: they overlap, ~600: they do not); n = entries of info (>= 4);
 * with n >= 5: info[4] = how nodes were packed into the node kernel's workgroups: 0 by degree, 1 by the static load estimate, 2 by
 * the measured cost pedn_model_desc.node_cost. */
int pedn_plan_info(pedn_sim* sim, int32_t* info, int32_t n);

/* reset all histories and dynamic state to t = 0 (widths, turning fractions and demand are kept) */
int pedn_reset(pedn_sim* sim);
/* The same for a caller that resets every episode (rl/pz_pednet_env.py:143-193 rebuilds the Network instead): in full-record mode only
 * what a new episode reads before it writes is restored -- row 0 of every field, avg_travel_time below the moving-average window
 * (link.py:91), the gate record (link.py:56) -- and every other row counts as unwritten until its step runs: pedn_read answers such
 * rows with the field's initial value (what the reference's fresh arrays hold), the kernels do the same for the only reads that can
 * land there (get_outflow's negative indices, link.py:205-212), an out-of-order step or observation clears the rows it skips first,
 * and pedn_device_ptr clears the rest, because a zero-copy consumer may look anywhere.  Recent-history mode: the same as pedn_reset. */
int pedn_reset_lazy(pedn_sim* sim);

/* ---- per-replica scenarios on one topology (reference: NetworkEnvGenerator.create_network with
 * link_params_overrides / od_flows / demand_params_overrides, src/utils/env_loader.py:81-158, as produced by
 * generate_random_link_params :363-424, generate_random_od_flows :224-259, generate_random_demand_params :183-222).
 * Demand per replica is pedn_set_demand(node, replica, ...).  Call pedn_reset afterwards: travel_time[0] depends on them. */
/* matrices [n_links][n_replicas]: k_critical, k_jam, free_flow_speed and the host-derived free_flow_tau (link.py:86),
 * shock-wave look-back (link.py:380) and travel_time[0] (link.py:83).  kc == NULL returns to the shared parameters. */
int pedn_set_link_params(pedn_sim* sim, const double* kc, const double* kj, const double* vf, const int32_t* free_flow_tau,
                         const int32_t* tau_sw, const float* tt0);
/* the per-replica link parameters back to the host, matrices [n_links][n_replicas]; any pointer may be NULL */
int pedn_get_link_params(pedn_sim* sim, double* kc, double* kj, double* vf, int32_t* free_flow_tau, int32_t* tau_sw, float* tt0);
/* w[n_od][n_replicas]: time-constant OD weights per replica (the randomiser's np.full(T+1, weight)); NULL returns to the
 * shared, time-varying weights of pedn_set_od_weights */
int pedn_set_od_weights_per_replica(pedn_sim* sim, const double* w);
int pedn_get_od_weights_per_replica(pedn_sim* sim, double* w);
/* A new scenario for EVERY replica drawn on the device, in place -- what NetworkEnvGenerator.randomize_network draws per env
 * (src/utils/env_loader.py:160-181) without the host: `what` bit 0 link parameters (generate_random_link_params :363-424: exactly
 * int(corridors * link_fraction) corridors per replica without replacement; with probability 1/2 k_critical and k_jam x U(0.6, 1.2)
 * with the floors max(0.5, .) / max(2 k_c, .); with probability 1/2 free_flow_speed x U(0.6, 0.9); look-backs and travel_time[0]
 * re-derived, link.py:58-63,83-86,380), bit 1 OD weights (generate_random_od_flows :224-259: U(1, 10) per OD pair, constant over
 * the episode), bit 2 origin demand (generate_random_demand_params :183-222 + the series of od_manager.py:92-155 as pedn_draw_demand)
 * for the n_origins nodes listed.  Philox4x32-10 keyed by (seed, global replica id): a function of the seed, independent of how the
 * ensemble is sharded; the same distributions as the reference's randomisers, NOT numpy's stream: pinned bit for bit by
 * the CPU restatement oracle/rand_contract.py and distribution-tested against the numpy generator it replaces.
 * Asynchronous; call pedn_reset afterwards. */
int pedn_randomize_scenarios(pedn_sim* sim, uint64_t seed, double link_fraction, int32_t what, const int32_t* origin_nodes,
                             int32_t n_origins);

/* ---- batched RL environment step around the hot path ------------------------------------------------
 * Replaces, for every replica at once, the per-env Python glue of the reference's PettingZoo wrapper:
 *   pedn_rl_apply_actions   ActionApplier.apply_all_actions / clip_*_action_value     rl/builders.py:264-352
 *   pedn_rl_observe         ObservationBuilder.build_observation                       rl/builders.py:68-177 (+ :179-238 normalisation)
 *                           PedNetParallelEnv._compute_rewards                          rl/pz_pednet_env.py:548-581
 * Agents are listed in the reference's order (separators, then gaters; rl/discovery.py:121-123).  An action row holds,
 * agent after agent, one width per separator and one width per outgoing link of a gater; an observation row holds 4
 * features per separator and features_per_link x outdegree per gater (no padding, discovery.py:176-178). */
typedef struct pedn_rl_desc {
  int32_t n_agents;
  const int32_t* agent_type;     /* [n_agents] 0 separator, 1 gater */
  const int32_t* agent_link_ptr; /* [n_agents+1] */
  const int32_t* agent_links;    /* gater: controlled outgoing links; separator: forward link, reverse link */
  int32_t obs_mode;              /* 1..5 = "option1".."option5" (builders.py:47-58) */
  int32_t normalize;             /* builders.py:179-238 */
  int32_t reward_mode;           /* 0: as the reference runs (its `return` sits inside the agent loop, pz_pednet_env.py:581,
                                       so only the first agent is ever rewarded); 1: every gater agent */
  double max_delta_sep, max_delta_gate, min_sep; /* pz_pednet_env.py:84-86 */
} pedn_rl_desc;

/* validates the description and allocates the device buffers; returns the row lengths */
int pedn_rl_configure(pedn_sim* sim, const pedn_rl_desc* desc, int32_t* n_actions, int32_t* n_obs);
/* actions[n_replicas][n_actions] (binary64, widths in metres); on_device != 0: `actions` is a device pointer.  A NaN entry
 * means "no action for this slot": the widths it controls stay exactly as they are (apply_all_actions only touches the agents
 * it is given, rl/builders.py:343-352) */
int pedn_rl_apply_actions(pedn_sim* sim, const double* actions, int32_t on_device);
/* observations and rewards of step t (after pedn_step(t)) into the device buffers; accumulate != 0 adds the rewards to
 * the buffer (float32, like the reference's cumulative_rewards) instead of overwriting.  obs / rewards may be NULL;
 * otherwise they receive host copies [n_replicas][n_obs] / [n_replicas][n_agents] (synchronises). */
int pedn_rl_observe(pedn_sim* sim, int32_t t, int32_t accumulate, float* obs, float* rewards);
/* The observation / reward buffers as the last pedn_rl_step / pedn_rl_observe left them -> host (either pointer may be NULL); waits
 * for the engine's work.  For callers that step SEVERAL engines before they look at any of them: pedn_rl_step(..., NULL, NULL) on every
 * engine, then one fetch each -- the engines' launches overlap instead of alternating with host waits (MultiScenarioVecEnv: one engine
 * per randomised topology, rl/pz_pednet_env.py:143-193 rebuilds the network per reset). */
int pedn_rl_fetch(pedn_sim* sim, float* obs, float* rewards);
/* pedn_rl_step(.., NULL, NULL) on every engine of sims[0 .. n), then pedn_rl_fetch of each, in ONE call: the engines' launches overlap and
 * the per-engine cost of a host-side loop over handles is gone (MultiScenarioVecEnv: 32 engines x 64 envs).  actions / obs / rewards are the
 * engines' rows one after the other (engine k: its n_replicas rows); all engines step the same t with the same agent layout. */
int pedn_rl_step_many(pedn_sim** sims, int32_t n, const double* actions, int32_t t, int32_t action_gap, float* obs, float* rewards);
/* apply -> action_gap x (pedn_step(t+k), observe) in one call (pz_pednet_env.py:195-254).  obs / rewards NULL: asynchronous, the
 * results stay in the device buffers (pedn_rl_device_ptr), complete after pedn_synchronize.  With on_device actions and nothing
 * fetched, large batches step as two chains on two streams that stay forked across calls (the halves of the envs are independent);
 * every call that needs the whole batch joins them first, so callers see the same ordering as on one stream.  on_device = 2: device
 * actions as with 1, and every launch of the call goes to pedn_stream() -- for a caller that orders the call between its own streams
 * and the engine's with events instead of host synchronisation (VecPedNetEnv.step_device(sync=False)). */
int pedn_rl_step(pedn_sim* sim, const double* actions, int32_t on_device, int32_t t, int32_t action_gap, float* obs,
                 float* rewards);
/* ---- env steps with CONSTANT launch arguments: a captured graph of (policy -> env step -> reward bookkeeping) can be replayed ------
 * (rl/pz_pednet_env.py:195-254 is called once per policy step, rl/train_ppo_sb3.py:246 wraps one env; here the step index lives in
 * device memory, so the launches of a step do not change from step to step.)
 *   pedn_rl_clock_begin(t)    everything pending is settled on pedn_stream(), the device clock is set to step t.  The turning fractions
 *                             of step t must be in place already (they are after pedn_rl_step(t - 1)): the first step of an episode
 *                             runs through pedn_rl_step, whose stand-alone fractions read the gate widths behind that step's actions.
 *   pedn_rl_step_clocked      ONE env step (apply -> action_gap x (network_loading, observe)) enqueued on `stream` (a hipStream_t; NULL =
 *                             pedn_stream()) without touching the host's bookkeeping: no allocation, no synchronisation, no event query --
 *                             safe under stream capture.  `actions` is a device pointer [n_replicas][n_actions] (or NULL) read when
 *                             the launch RUNS: a replayed graph reads whatever the buffer holds then.  The step's last launch advances
 *                             the clock; a step enqueued beyond the horizon T does nothing.  Observations / rewards: pedn_rl_device_ptr.
 *                             The caller orders `stream` behind pedn_stream() once after pedn_rl_clock_begin.
 *   pedn_rl_clock_end(&t)     synchronises the DEVICE, reads the clock back (t = the next step to run) and restores the host's
 *                             bookkeeping; every other entry point that steps, reads or changes the state does this implicitly first. */
int pedn_rl_clock_begin(pedn_sim* sim, int32_t t);
int pedn_rl_step_clocked(pedn_sim* sim, const double* actions, int32_t action_gap, void* stream);
int pedn_rl_clock_end(pedn_sim* sim, int32_t* t);
/* 1 while a clocked section is open (pedn_rl_clock_begin .. the next call that ends it), else 0: a caller that replays a captured
 * graph checks this before every replay -- another entry point may have ended the section -- and begins it again if need be */
int pedn_rl_clocked(pedn_sim* sim);
/* A hash of everything the launches of pedn_rl_step_clocked carry by value (device pointers, sizes, whether replicas have their own
 * scenarios, the agent set) and of what selects their kernels and grids.  A captured graph is valid as long as this value is the one
 * it was captured under: e.g. the first reset(options={'randomize': True}) switches the step to the per-replica-parameter kernels. */
uint64_t pedn_rl_clock_signature(pedn_sim* sim);
/* device buffers for zero-copy consumers: 0 actions (f64 [R][n_actions]), 1 observations (f32 [R][n_obs]), 2 rewards (f32 [R][n_agents]) */
void* pedn_rl_device_ptr(pedn_sim* sim, int32_t which);

/* Origin demand drawn on the device for every replica at once -- the arrays DemandGenerator builds on the host
 * (src/LTM/od_manager.py:92-155) for the patterns generate_random_demand_params picks (src/utils/env_loader.py:183-222):
 *   pattern 0 gaussian_peaks: Poisson(base + peak * (bump(T/4) + bump(3T/4))), bump(c)(t) = exp(-(t-c)^2 / (2 (T/20)^2)), t < T
 *   pattern 1 constant:       base for t <= T
 *   pattern 2 sudden_demand:  pattern 0 plus `spike_height` for spike_start <= t < spike_start + spike_len
 * Arrays are [n_replicas] for ONE origin node.  The Poisson draws are keyed (seed, global replica id, node, t) with
 * Philox4x32-10 and inverted by sequential search; this is a generator of its own (same distributions as the reference,
 * not numpy's stream), pinned bit for bit by oracle/rand_contract.py (glibc's exp on both sides). */
int pedn_draw_demand(pedn_sim* sim, int32_t node, uint64_t seed, const int32_t* pattern, const double* base, const double* peak,
                     const int32_t* spike_start, const int32_t* spike_len, const double* spike_height);

/* Diagnostic: evaluate the device-side arithmetic primitives on the GPU so that tests can compare them bit for bit
 * with the oracle.  op 0: powf(a[i], b[i]) -> out (f32 in a/b/out as doubles); 1: exp(a[i]); 2: sqrt(a[i]);
 * 3: a[i]/b[i] (f64); 4: (float)a[i]/(float)b[i]; 5: binomial(n=a[i], p=b[i]) keyed (seed, replica=i, link=7, t=11,
 * site=0); 6: normal(sigma=a[i]) keyed (seed, replica=i, link=7, t=11); 7 / 8: a[i] + b[i] with 8- / 16-byte accesses
 * (streaming launches of known byte count for calibrating the HBM counters and the practical bandwidth ceiling); 9: the same
 * with chunks of `seed` doubles visited in a scrambled order (n and seed powers of two). */
int pedn_device_math(int32_t device, int32_t op, int32_t n, const double* a, const double* b, uint64_t seed, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PEDN_H */
