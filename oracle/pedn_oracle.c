/* pedn_oracle.c -- TEST INFRASTRUCTURE: scalar CPU restatement of PedNStream's network_loading(t).
 *
 * This file is the parity oracle for the HIP engine.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  It is pinned against the real reference
 * (run with the injected RNG of oracle/rng_contract.py by oracle/ref_harness.py) through the golden fixtures under
 * tests/golden/ -- see tests/test_oracle_golden.py.
 *
 * One instance = one replica, state stored [field][link][T+1] (time fastest, like the reference's per-link numpy
 * arrays), plain sequential loops in the reference's own order:
 *
 *   step()            Network.network_loading            /root/reference/src/LTM/network.py:266-287
 *   dyn_tf()          PathFinder.calculate_node_turning_fractions / update_turning_fractions /
 *                     update_node_turn_probs / check_fractions   src/LTM/path_finder.py:717-737,591-689,561-589,691-715
 *   send_flow()       Link.cal_sending_flow + get_outflow  src/LTM/link.py:216-370,199-214
 *   recv_flow()       Link/Separator.cal_receiving_flow[_with_reverse]  src/LTM/link.py:372-416,480-512
 *   node_solve()      OneToOneNode.solve / RegularNode.solve('classic') + update_links  src/LTM/node.py:230-242,272-300,146-162
 *   link_update()     Link.update_link_density_flow + update_speeds + BiDirectionalFd.__call__
 *                     src/LTM/link.py:133-136,141-188,430-452; src/utils/functions.py:112-134
 *
 * Mixed precision follows numpy 2.x (NEP 50) scalar semantics of the reference: every place where the reference
 * holds an np.float32 is a C float here.  powf is glibc 2.35's algorithm restated (pw_powf) because numpy's float32
 * scalar power calls libm powf, whose result is not correctly rounded and therefore has to be reproduced, not
 * approximated; exp (softmax, path_finder.py:585) is an independent high-accuracy implementation (pw_exp) that the HIP
 * engine shares bit for bit.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pedn.h"

/* ------------------------------------------------------------------------------------------------ RNG contract */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

typedef struct { uint64_t seed; uint32_t replica, link, t, site; } rng_key;

static inline void rng_words(const rng_key* k, uint32_t call, uint32_t out[4]) {
  out[0] = k->t; out[1] = k->link; out[2] = k->site | (call << 8); out[3] = k->replica;
  philox4x32_10(out, (uint32_t)k->seed, (uint32_t)(k->seed >> 32));
}

/* Irwin-Hall(8) over the eight 16-bit halves of ONE Philox call; 0x1.3988e1409212ep-16 = sqrt(1.5) * 2^-16 */
static double rng_z(const rng_key* k) {
  uint32_t w[4];
  rng_words(k, 0, w);
  uint32_t s = 0;
  for (int i = 0; i < 4; ++i) s += (w[i] & 0xffffu) + (w[i] >> 16);
  return (double)((int32_t)s - 262140) * 0x1.3988e1409212ep-16;
}

static int64_t rng_binomial(int64_t n, double p, const rng_key* k, int mode) {
  if (n <= 0 || p <= 0.0) return 0;
  if (mode == PEDN_RNG_MEANFIELD) return (int64_t)floor((double)n * p);
  if (p >= 1.0) return n;
  if (n <= 16) {
    uint32_t thr = (uint32_t)floor(p * 65536.0);
    uint32_t w[4];
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
      if ((i & 7) == 0) rng_words(k, (uint32_t)(i >> 3), w);
      uint32_t h = (w[(i >> 1) & 3] >> (16 * (i & 1))) & 0xffffu;
      cnt += h < thr;
    }
    return cnt;
  }
  double mean = (double)n * p;
  double sd = sqrt(mean * (1.0 - p));
  double x = floor(mean + sd * rng_z(k) + 0.5);
  if (x < 0.0) x = 0.0;
  if (x > (double)n) x = (double)n;
  return (int64_t)x;
}

/* ------------------------------------------------------------------------------------------------ powf / exp */
/* glibc 2.35 powf (sysdeps/ieee754/flt-32/e_powf.c, from ARM Optimized Routines), restated for x >= 0 finite,
 * y > 0 finite -- the only domain the hot path uses (link.py:212 (1-F)**2, **3; link.py:317 rf**0.8).
 * Tables: __powf_log2_data (16 entries, degree-5 polynomial) and __exp2f_data (32 entries, degree-3). */
static const double PW_LOG2_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double PW_LOG2_POLY[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
                                       -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp0};
static const uint64_t PW_EXP2_TAB[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
static const double PW_EXP2_POLY[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

float pw_powf(float x, float y) {
  if (x == 0.0f) return 0.0f;
  uint32_t ix = f2u(x);
  if (ix < 0x00800000u) { /* subnormal: normalise */
    ix = f2u(x * 0x1p23f);
    ix -= 23u << 23;
  }
  /* log2_inline */
  uint32_t tmp = ix - 0x3f330000u;
  int i = (tmp >> (23 - 4)) % 16;
  uint32_t top = tmp & 0xff800000u;
  uint32_t iz = ix - top;
  int k = (int32_t)top >> 23;
  double invc = PW_LOG2_TAB[i][0], logc = PW_LOG2_TAB[i][1];
  double z = (double)u2f(iz);
  double r = z * invc - 1;
  double y0 = logc + (double)k;
  double r2 = r * r;
  double yy = PW_LOG2_POLY[0] * r + PW_LOG2_POLY[1];
  double p = PW_LOG2_POLY[2] * r + PW_LOG2_POLY[3];
  double r4 = r2 * r2;
  double q = PW_LOG2_POLY[4] * r + y0;
  q = p * r2 + q;
  yy = yy * r4 + q;
  double ylogx = (double)y * yy;
  if (ylogx <= -150.0) return 0.0f;
  /* exp2_inline */
  double kd = ylogx + 0x1.8p+47;
  uint64_t ki = d2u(kd);
  kd -= 0x1.8p+47;
  double rr = ylogx - kd;
  uint64_t t = PW_EXP2_TAB[ki % 32];
  t += ki << (52 - 5);
  double s = u2d(t);
  double zz = PW_EXP2_POLY[0] * rr + PW_EXP2_POLY[1];
  double rr2 = rr * rr;
  double e = PW_EXP2_POLY[2] * rr + 1;
  e = zz * rr2 + e;
  e = e * s;
  return (float)e;
}

/* exp(x) for the softmax (path_finder.py:585): numpy's double exp loop falls through to libm's exp (with the AVX512F
 * loop disabled, see oracle/ref_harness.py), i.e. glibc 2.35 `__exp_fma` on every x86 with FMA (sysdeps/ieee754/dbl-64/e_exp.c
 * built with -mfma -mavx2, selected by ifunc).  The algorithm (N = 128 table of 2^(k/128), degree-5 polynomial) is
 * restated here with the SAME fused multiply-adds the shipped binary uses (read off its disassembly):
 *   kd+Shift = fma(x, InvLn2N, Shift);  r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
 *   tmp = fma(r2*r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r));  result = fma(scale, tmp, scale)
 * fma() is exact by definition, so CPU and GPU agree bit for bit.  Domain of the main path: 2^-54 <= |x| < 512. */
static const uint64_t PW_EXP_TAB[128][2] = {
#include "exp_table.inc"
};

double pw_exp(double x) {
  uint32_t abstop = (uint32_t)(d2u(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x; /* |x| < 2^-54 */
    if (x != x) return x;
    if (abstop >= 0x409u) return (d2u(x) >> 63) ? 0.0 : INFINITY; /* |x| >= 1024 */
    /* 512 <= |x| < 1024: outside the softmax's range; libm's special-case scaling is not restated */
    return exp(x);
  }
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  double kd = fma(x, InvLn2N, Shift);
  uint64_t ki = d2u(kd);
  kd -= Shift;
  double r = fma(kd, NegLn2loN, fma(kd, NegLn2hiN, x));
  uint64_t idx = ki % 128;
  uint64_t top = ki << 45;
  double tail = u2d(PW_EXP_TAB[idx][0]);
  uint64_t sbits = PW_EXP_TAB[idx][1] + top;
  double r2 = r * r;
  double tmp = fma(r2 * r2, fma(r, C5, C4), fma(fma(r, C3, C2), r2, tail + r));
  double scale = u2d(sbits);
  return fma(scale, tmp, scale);
}

/* ------------------------------------------------------------------------------------------------ state */
struct pedn_oracle {
  pedn_model_desc m; /* shallow copy; arrays deep-copied below */
  int n_slots, Lall, T1;
  uint64_t seed;
  uint32_t replica;
  int mode;
  uint32_t flags;
  /* which paths of cal_sending_flow ran since the last reset (bench.py: the contract's bytes on EXECUTED paths): [0] calls,
   * [1] past the free-flow gate (link.py:267-269), [2] sending flow positive before the draw (:298), [3] get_outflow's four
   * inflow look-backs read (:323, :199-214), [4] activity draw (:351-358) */
  uint64_t tally[5];
  /* histories */
  double* f64[7]; /* [Lall][T1] for 0..3, [L][T1] for 4..6 */
  float* f32[6];  /* [L][T1]: tt, att, N, k, v, lf */
  float* rsum;    /* [L] */
  double *front, *back, *sepw, *sepnp; /* [L]; sepnp: separator width is a numpy float64 scalar (see PEDN_W_SEP_NUMPY) */
  double* tf;                  /* [n_turns] */
  double* demand;              /* [n_demand][T1] */
  double* od_w;                /* [n_od][T1] */
  double *upod_p, *ent_p;      /* scratch: P(od|up) [n_upod], P(down|up,od) [n_ent] */
  void* owned[64];
  int n_owned;
};
typedef struct pedn_oracle pedn_oracle;

#define F_IN 0
#define F_OUT 1
#define F_CI 2
#define F_CO 3
#define F_S 4
#define F_R 5
#define F_GATE 6
#define G_TT 0
#define G_ATT 1
#define G_N 2
#define G_K 3
#define G_V 4
#define G_LF 5

static void* own(pedn_oracle* o, const void* src, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (src && bytes) memcpy(p, src, bytes);
  o->owned[o->n_owned++] = p;
  return p;
}

static inline double* H(pedn_oracle* o, int f, int link) { return o->f64[f] + (size_t)link * o->T1; }
static inline float* G(pedn_oracle* o, int f, int link) { return o->f32[f] + (size_t)link * o->T1; }

/* Python negative-index wrap-around on a length T+1 array */
static inline int wrap(pedn_oracle* o, int i) {
  if (i < 0) i += o->T1;
  if (i < 0 || i >= o->T1) { o->flags |= PEDN_F_INDEX; return 0; }
  return i;
}

void pedn_oracle_reset(pedn_oracle* o) {
  const pedn_model_desc* m = &o->m;
  int L = m->n_links, T1 = o->T1;
  for (int f = 0; f < 4; ++f) memset(o->f64[f], 0, sizeof(double) * (size_t)o->Lall * T1);
  for (size_t i = 0; i < (size_t)L * T1; ++i) { o->f64[F_S][i] = -1.0; o->f64[F_R][i] = -1.0; }
  for (int f = 0; f < 6; ++f) memset(o->f32[f], 0, sizeof(float) * (size_t)L * T1);
  for (int l = 0; l < L; ++l) {
    float tt0 = m->link_tt0[l];
    G(o, G_TT, l)[0] = tt0;
    for (int t = 0; t < m->window && t < T1; ++t) G(o, G_ATT, l)[t] = tt0;
    o->rsum[l] = tt0;
    /* back_gate_width_data = width * ones (link.py:56), also for a Separator: its __init__ runs Link.__init__ first */
    for (int t = 0; t < T1; ++t) H(o, F_GATE, l)[t] = m->link_width[l];
  }
  o->flags = 0;
}

pedn_oracle* pedn_oracle_create(const pedn_model_desc* md, uint64_t seed, int32_t replica, int32_t mode) {
  pedn_oracle* o = (pedn_oracle*)calloc(1, sizeof(*o));
  o->m = *md;
  pedn_model_desc* m = &o->m;
  int N = m->n_nodes, L = m->n_links;
  o->n_slots = md->node_slot_ptr[N];
  o->Lall = L + m->n_vlinks;
  o->T1 = m->T + 1;
  o->seed = seed; o->replica = (uint32_t)replica; o->mode = mode;
#define CP(field, count, type) m->field = (const type*)own(o, md->field, sizeof(type) * (size_t)(count))
  CP(node_kind, N, int32_t); CP(node_slot_ptr, N + 1, int32_t); CP(node_turn_ptr, N + 1, int32_t);
  CP(node_demand_row, N, int32_t); CP(node_dyn, N, int32_t);
  CP(slot_in_link, o->n_slots, int32_t); CP(slot_out_link, o->n_slots, int32_t);
  CP(link_rev, L, int32_t); CP(link_sep, L, int32_t); CP(link_fd, L, int32_t); CP(link_tau_sw, L, int32_t);
  CP(link_fft, L, int32_t); CP(link_tt0, L, float);
  CP(link_length, L, double); CP(link_width, L, double); CP(link_vf, L, double); CP(link_kc, L, double);
  CP(link_kj, L, double); CP(link_gamma, L, double); CP(link_act, L, double); CP(link_bi, L, double);
  CP(link_noise, L, double);
  CP(node_up_ptr, N + 1, int32_t); CP(up_od_ptr, m->n_up + 1, int32_t); CP(upod_od, m->n_upod, int32_t);
  CP(node_grp_ptr, N + 1, int32_t); CP(grp_ent_ptr, m->n_grp + 1, int32_t); CP(grp_allphys, m->n_grp, int32_t);
  CP(ent_link, m->n_ent, int32_t); CP(ent_dist, m->n_ent, double);
  CP(turn_pair_ptr, m->n_turns + 1, int32_t); CP(pair_ent, m->n_pair, int32_t); CP(pair_upod, m->n_pair, int32_t);
#undef CP
  size_t T1 = (size_t)o->T1;
  for (int f = 0; f < 4; ++f) o->f64[f] = (double*)own(o, NULL, sizeof(double) * o->Lall * T1);
  for (int f = 4; f < 7; ++f) o->f64[f] = (double*)own(o, NULL, sizeof(double) * L * T1);
  for (int f = 0; f < 6; ++f) o->f32[f] = (float*)own(o, NULL, sizeof(float) * L * T1);
  o->rsum = (float*)own(o, NULL, sizeof(float) * L);
  o->front = (double*)own(o, md->front_gate0, sizeof(double) * L);
  o->back = (double*)own(o, md->back_gate0, sizeof(double) * L);
  o->sepw = (double*)own(o, md->sep_width0, sizeof(double) * L);
  o->sepnp = (double*)own(o, NULL, sizeof(double) * L);
  memset(o->sepnp, 0, sizeof(double) * L);
  o->tf = (double*)own(o, md->tf_init, sizeof(double) * m->n_turns);
  o->demand = (double*)own(o, md->demand, sizeof(double) * m->n_demand * T1);
  o->od_w = (double*)own(o, md->od_w, sizeof(double) * m->n_od * T1);
  o->upod_p = (double*)own(o, NULL, sizeof(double) * (m->n_upod + 1));
  o->ent_p = (double*)own(o, NULL, sizeof(double) * (m->n_ent + 1));
  m->front_gate0 = m->back_gate0 = m->sep_width0 = m->tf_init = m->demand = m->od_w = NULL;
  pedn_oracle_reset(o);
  return o;
}

void pedn_oracle_destroy(pedn_oracle* o) {
  if (!o) return;
  for (int i = 0; i < o->n_owned; ++i) free(o->owned[i]);
  free(o);
}

/* ------------------------------------------------------------------------------------------------ link model */
static inline double area_of(pedn_oracle* o, int l) {
  const pedn_model_desc* m = &o->m;
  return m->link_length[l] * (m->link_sep[l] ? o->sepw[l] : m->link_width[l]); /* link.py:128-131,454-456 */
}

/* Link.get_density (link.py:190-197) / Separator.get_density (:427-428) */
static inline float dens_of(pedn_oracle* o, int l, int t) {
  const pedn_model_desc* m = &o->m;
  if (m->link_sep[l]) return G(o, G_K, l)[t];
  float n = G(o, G_N, l)[t] + G(o, G_N, m->link_rev[l])[t];
  return n / (float)area_of(o, l);
}

static inline float clip01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

static double send_flow(pedn_oracle* o, int l, int tp) {
  const pedn_model_desc* m = &o->m;
  double* S = H(o, F_S, l);
  float dens = dens_of(o, l, tp);
  float att = G(o, G_ATT, l)[tp];
  int tau = (int)lrintf(att / (float)m->dt); /* link.py:260, round-half-even of an f32 */
  o->tally[0]++;
  if (tp < m->link_fft[l]) { S[tp] = 0.0; return 0.0; } /* link.py:267-269 */
  o->tally[1]++;
  if (tau <= 0) o->flags |= PEDN_F_SAME_STEP;
  int idx = tp + 1 - tau; if (idx < 0) idx = 0; /* link.py:274 */
  double kc = m->link_kc[l], kj = m->link_kj[l], vf = m->link_vf[l];
  float kk = G(o, G_K, l)[tp];
  float cf = clip01((kk - (float)kc) / (float)(kj - kc)); /* link.py:282 */
  float nped = G(o, G_N, l)[tp];
  double ff = H(o, F_CI, l)[idx] - H(o, F_CO, l)[tp];
  if (!(ff > 0.0)) ff = 0.0;
  double bnd = (double)(cf * nped) + (double)(1.0f - cf) * ff; /* link.py:284-288 */
  double smax = o->front[l] * kc * vf * m->dt;                 /* link.py:296 */
  double s = smax < bnd ? smax : bnd; /* link.py:297 */
  double orig = s;
  rng_key key = {o->seed, o->replica, (uint32_t)l, (uint32_t)tp, 0};
  if (s > 0.0) {
    o->tally[2]++;
    float rf = clip01(dens / (float)kj);                                 /* link.py:315 */
    float p = 0.7f + (float)(0.85 - 0.7) * pw_powf(rf, 0.8f);            /* link.py:317 */
    int diffusion_used = 0;
    if (dens <= (float)kc) {                                             /* link.py:323 */
      /* get_outflow, link.py:199-214 */
      o->tally[3]++;
      float F = 1.0f / (1.0f + (float)m->link_gamma[l] * att);
      float Gq = 1.0f - F;
      const double* in = H(o, F_IN, l);
      double d = (double)F * in[wrap(o, tp - tau)] + (double)(F * Gq) * in[wrap(o, tp - tau - 1)] +
                 (double)(F * pw_powf(Gq, 2.0f)) * in[wrap(o, tp - tau - 2)] +
                 (double)(F * pw_powf(Gq, 3.0f)) * in[wrap(o, tp - tau - 3)];
      d = ceil(d);
      if (!(d > 0.0)) d = 0.0;
      if (d > 0.0) { /* link.py:326-330 */
        double mix = 0.8 * d + (1 - 0.8) * s;
        s = floor(mix < s ? mix : s);
        diffusion_used = 1;
      }
    }
    if (!diffusion_used) { /* link.py:336-338, 342-344 */
      key.site = 0;
      s = (double)rng_binomial((int64_t)floor(s), (double)p, &key, o->mode);
    }
    if (s < 0.0) o->flags |= PEDN_F_NEG_SENDING;
  }
  if (m->link_act[l] > 0.0 && s > 1.0) { /* link.py:351-358 */
    o->tally[4]++;
    key.site = 1;
    s -= (double)rng_binomial((int64_t)floor(s), m->link_act[l], &key, o->mode);
  }
  if (!(s > 0.0)) s = 0.0;
  double sm = floor(0.8 * s + 0.2 * S[wrap(o, tp - 1)]); /* link.py:364 */
  s = orig < sm ? orig : sm;
  if (s < 0.0) o->flags |= PEDN_F_NEG_SENDING;
  S[tp] = s;
  return s;
}

static double recv_flow(pedn_oracle* o, int l, int tp, double s_rev) {
  const pedn_model_desc* m = &o->m;
  int tsw = m->link_tau_sw[l];
  double kjA = m->link_kj[l] * area_of(o, l);
  double b;
  if (m->link_sep[l]) { /* link.py:480-512 */
    if (tp + 1 - tsw < 0) b = kjA;
    else {
      if (tsw <= 0) o->flags |= PEDN_F_SAME_STEP;
      b = H(o, F_CO, l)[tp + 1 - tsw] + kjA - H(o, F_CI, l)[tp];
    }
  } else { /* link.py:372-405 */
    float nrev = G(o, G_N, m->link_rev[l])[tp];
    if (nrev < 0.0f) o->flags |= PEDN_F_NEG_BINOM;
    rng_key key = {o->seed, o->replica, (uint32_t)l, (uint32_t)tp, 2};
    double rp = (double)rng_binomial((int64_t)nrev, 0.9, &key, o->mode);
    if (tp + 1 - tsw < 0) b = kjA - rp;
    else {
      if (tsw <= 0) o->flags |= PEDN_F_SAME_STEP;
      b = H(o, F_CO, l)[tp + 1 - tsw] + kjA - rp - H(o, F_CI, l)[tp];
      if (!(b > 0.0)) b = 0.0;
    }
  }
  double rmax = o->back[l] * m->link_kc[l] * m->link_vf[l] * m->dt; /* link.py:393 */
  double r = rmax < b ? rmax : b;
  if (!(r > 0.0)) r = 0.0;
  double prev = H(o, F_R, l)[wrap(o, tp - 1)];
  if (prev >= 0.0) { /* link.py:400-401 */
    double sm = floor(r * 0.8 + prev * 0.2);
    r = sm < r ? sm : r;
  }
  if (m->link_sep[l]) return r > 0.0 ? r : 0.0; /* link.py:509-512 */
  r = r - s_rev;                                 /* link.py:415-416 */
  return r > 0.0 ? r : 0.0;
}

/* PathFinder.update_turning_fractions + check_fractions for one node (path_finder.py:591-715) */
static void dyn_tf(pedn_oracle* o, int n, int t) {
  const pedn_model_desc* m = &o->m;
  /* P(od|up), :599-615 */
  for (int u = m->node_up_ptr[n]; u < m->node_up_ptr[n + 1]; ++u) {
    int a = m->up_od_ptr[u], b = m->up_od_ptr[u + 1];
    double tot = 0.0;
    for (int q = a; q < b; ++q) { o->upod_p[q] = o->od_w[(size_t)m->upod_od[q] * o->T1 + t]; tot += o->upod_p[q]; }
    if (tot > 0.0) for (int q = a; q < b; ++q) o->upod_p[q] /= tot;
    else for (int q = a; q < b; ++q) o->upod_p[q] = (b - a) > 0 ? 1.0 / (double)(b - a) : 0.0;
  }
  /* P(down|up,od), update_node_turn_probs :561-589 */
  for (int g = m->node_grp_ptr[n]; g < m->node_grp_ptr[n + 1]; ++g) {
    int a = m->grp_ent_ptr[g], b = m->grp_ent_ptr[g + 1];
    double sumd = 0.0, sumc = 0.0;
    double cap[PEDN_MAX_DEGREE + 1], kd[PEDN_MAX_DEGREE + 1];
    float kf[PEDN_MAX_DEGREE + 1];
    for (int e = a; e < b; ++e) {
      int l = m->ent_link[e];
      if (l >= 0) {
        kf[e - a] = dens_of(o, l, wrap(o, t - 1));
        kd[e - a] = (double)kf[e - a];
        double c = H(o, F_R, l)[wrap(o, t - 2)];
        cap[e - a] = c >= 0.0 ? c : o->back[l] * m->link_vf[l] * m->link_kc[l] * m->dt; /* :575-576 */
      } else {
        kf[e - a] = 0.0f; kd[e - a] = 0.0; cap[e - a] = 100.0; /* :577-579 */
      }
      sumd = (e == a) ? m->ent_dist[e] : sumd + m->ent_dist[e];
      sumc = (e == a) ? cap[e - a] : sumc + cap[e - a];
    }
    double esum = 0.0;
    for (int e = a; e < b; ++e) {
      double nd;
      if (m->grp_allphys[g]) { /* float32 array branch of :581,583 */
        float x = kf[e - a] - 2.0f; if (!(x > 0.0f)) x = 0.0f;
        nd = (double)((float)m->pf_beta * (x / 8.0f));
      } else {
        double x = kd[e - a] - 2.0; if (!(x > 0.0)) x = 0.0;
        nd = m->pf_beta * (x / 8.0);
      }
      double u = (m->pf_alpha * m->ent_dist[e]) / (sumd + 1e-6) + nd - (m->pf_omega * cap[e - a]) / (sumc + 1e-6) + m->pf_eps;
      double ex = pw_exp(-m->pf_temp * u);
      o->ent_p[e] = ex;
      esum = (e == a) ? ex : esum + ex;
    }
    for (int e = a; e < b; ++e) o->ent_p[e] /= esum;
  }
  /* tf[turn] = sum P(down|up,od) P(od|up), :668-686 ; then check_fractions :691-715 */
  int t0 = m->node_turn_ptr[n], mdeg = m->node_slot_ptr[n + 1] - m->node_slot_ptr[n];
  for (int i = 0; i < mdeg; ++i) {
    double rowsum = 0.0;
    for (int j = 0; j < mdeg - 1; ++j) {
      int tn = t0 + i * (mdeg - 1) + j;
      double acc = 0.0;
      for (int q = m->turn_pair_ptr[tn]; q < m->turn_pair_ptr[tn + 1]; ++q)
        acc += o->ent_p[m->pair_ent[q]] * o->upod_p[m->pair_upod[q]];
      o->tf[tn] = acc;
      rowsum = (j == 0) ? acc : rowsum + acc;
    }
    if (fabs(rowsum - 1) > 1e-3) {
      for (int j = 0; j < mdeg - 1; ++j) {
        int tn = t0 + i * (mdeg - 1) + j;
        if (rowsum > 1e-6) o->tf[tn] = o->tf[tn] / rowsum;
        else o->tf[tn] = 1.0 / (double)(mdeg - 1);
      }
    }
  }
}

/* RegularNode.solve('optimal') (node.py:249-271): min c.x with c = (-1 per flow, w per penalty variable), x >= 0,
 *   rows 0..m-1      sum_j f_ij <= s_i          rows m..2m-1   sum_i f_ij <= r_j                      (get_matrix_A, :73-98)
 *   rows 2m..2m+E-1  phi_e * sum_{j} f_(src e)j - f_e + p_e^+ - p_e^- = 0                               (update_matrix_A_eq, :110-137)
 * by a dense primal simplex on the full tableau, Bland's rule, starting from the feasible basis {slacks, p^+}.  The HIP
 * engine runs the same operations in the same order (pedn_kernels.hpp: lp_solve).  Returns 0 on success; g[e] = floor(f_e). */
int pedn_oracle_lp(int m, const double* s, const double* r, const double* tf, double* x_out, double* g) {
  const int E = m * (m - 1), R = 2 * m + E, N = 3 * E + 2 * m, W = N + 1;
  const double tol = 1e-9, w = PEDN_LP_PENALTY;
  double* T = (double*)calloc((size_t)(R + 1) * W, sizeof(double));
  int* basis = (int*)malloc((size_t)R * sizeof(int));
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      if (i == j) continue;
      const int e = i * (m - 1) + (j < i ? j : j - 1);
      T[(size_t)i * W + e] = 1.0;
      T[(size_t)(m + j) * W + e] = 1.0;
    }
  for (int k = 0; k < 2 * m; ++k) {
    T[(size_t)k * W + 3 * E + k] = 1.0;
    T[(size_t)k * W + N] = k < m ? s[k] : r[k - m];
    basis[k] = 3 * E + k;
  }
  for (int e = 0; e < E; ++e) {
    const int i = e / (m - 1), k = 2 * m + e;
    for (int c = i * (m - 1); c < (i + 1) * (m - 1); ++c) T[(size_t)k * W + c] = tf[e];
    T[(size_t)k * W + e] = tf[e] - 1;
    T[(size_t)k * W + E + 2 * e] = 1.0;
    T[(size_t)k * W + E + 2 * e + 1] = -1.0;
    basis[k] = E + 2 * e;
  }
  /* objective row: reduced costs; the starting basis holds the p^+ (cost w) */
  for (int c = 0; c < N; ++c) T[(size_t)R * W + c] = c < E ? -1.0 : (c < 3 * E ? w : 0.0);
  for (int k = 2 * m; k < R; ++k)
    for (int c = 0; c <= N; ++c) T[(size_t)R * W + c] -= w * T[(size_t)k * W + c];
  int status = 1;
  for (int it = 0; it < 40 * (R + N); ++it) {
    int j = -1;
    for (int c = 0; c < N; ++c)
      if (T[(size_t)R * W + c] < -tol) { j = c; break; }
    if (j < 0) { status = 0; break; }
    int i = -1;
    double best = 0.0;
    for (int k = 0; k < R; ++k) {
      const double a = T[(size_t)k * W + j];
      if (a > tol) {
        const double ratio = T[(size_t)k * W + N] / a;
        if (i < 0 || ratio < best - 1e-12 || (fabs(ratio - best) <= 1e-12 && basis[k] < basis[i])) { best = ratio; i = k; }
      }
    }
    if (i < 0) break; /* unbounded: cannot happen, the flows are bounded by s */
    const double piv = T[(size_t)i * W + j];
    for (int c = 0; c <= N; ++c) T[(size_t)i * W + c] = T[(size_t)i * W + c] / piv;
    for (int k = 0; k <= R; ++k) {
      if (k == i) continue;
      const double f = T[(size_t)k * W + j];
      if (f == 0.0) continue;
      for (int c = 0; c <= N; ++c) T[(size_t)k * W + c] = T[(size_t)k * W + c] - f * T[(size_t)i * W + c];
    }
    basis[i] = j;
  }
  for (int e = 0; e < E; ++e) g[e] = 0.0;
  if (x_out) for (int c = 0; c < N; ++c) x_out[c] = 0.0;
  if (status == 0)
    for (int k = 0; k < R; ++k) {
      if (x_out) x_out[basis[k]] = T[(size_t)k * W + N];
      if (basis[k] < E) g[basis[k]] = floor(T[(size_t)k * W + N]);
    }
  free(T);
  free(basis);
  return status;
}

static void node_step(pedn_oracle* o, int n, int t) {
  const pedn_model_desc* m = &o->m;
  int s0 = m->node_slot_ptr[n], mdeg = m->node_slot_ptr[n + 1] - s0;
  int L = m->n_links, tp = t - 1;
  double s[PEDN_MAX_DEGREE], r[PEDN_MAX_DEGREE], qo[PEDN_MAX_DEGREE], qi[PEDN_MAX_DEGREE];
  if (mdeg > PEDN_MAX_DEGREE) { o->flags |= PEDN_F_INDEX; return; }
  if (m->node_dyn[n]) dyn_tf(o, n, t);
  for (int i = 0; i < mdeg; ++i) { /* node.py:172-178 */
    int l = m->slot_in_link[s0 + i];
    if (l >= L) s[i] = o->demand[(size_t)m->node_demand_row[n] * o->T1 + tp];
    else s[i] = send_flow(o, l, tp);
  }
  for (int j = 0; j < mdeg; ++j) { /* node.py:184-206 */
    int l = m->slot_out_link[s0 + j];
    if (l >= L) r[j] = 1e6;
    else {
      double srev = H(o, F_S, m->link_rev[l])[tp];
      if (srev < 0.0) o->flags |= PEDN_F_NEG_FLOW;
      r[j] = recv_flow(o, l, tp, srev);
      H(o, F_R, l)[tp] = r[j];
    }
  }
  for (int i = 0; i < mdeg; ++i) if (s[i] < 0.0 || r[i] < 0.0) o->flags |= PEDN_F_NEG_FLOW;
  if (m->node_kind[n] == 0) { /* OneToOneNode.solve, node.py:230-242 */
    qo[0] = qi[1] = s[0] < r[1] ? s[0] : r[1];
    qo[1] = qi[0] = s[1] < r[0] ? s[1] : r[0];
    if (qo[0] < 0.0 || qo[1] < 0.0) o->flags |= PEDN_F_NEG_FLOW;
  } else if (m->node_model == PEDN_NODE_OPTIMAL) { /* RegularNode.solve('optimal'), node.py:249-271 */
    double g[PEDN_MAX_DEGREE * (PEDN_MAX_DEGREE - 1)];
    if (pedn_oracle_lp(mdeg, s, r, o->tf + m->node_turn_ptr[n], NULL, g) != 0) o->flags |= PEDN_F_LP;
    for (int i = 0; i < mdeg; ++i) qo[i] = qi[i] = 0.0;
    for (int i = 0; i < mdeg; ++i)
      for (int j = 0; j < mdeg; ++j) {
        if (i == j) continue;
        const double f = g[i * (mdeg - 1) + (j < i ? j : j - 1)];
        qo[i] += f; qi[j] += f;
      }
    for (int i = 0; i < mdeg; ++i) { if (!(qo[i] > 0.0)) qo[i] = 0.0; if (!(qi[i] > 0.0)) qi[i] = 0.0; }
  } else { /* RegularNode.solve('classic'), node.py:272-300 */
    const double* tf = o->tf + m->node_turn_ptr[n];
    double ps[PEDN_MAX_DEGREE][PEDN_MAX_DEGREE], D[PEDN_MAX_DEGREE];
    for (int i = 0; i < mdeg; ++i)
      for (int j = 0; j < mdeg; ++j)
        ps[i][j] = (i == j) ? 0.0 * s[i] : tf[i * (mdeg - 1) + (j < i ? j : j - 1)] * s[i];
    for (int j = 0; j < mdeg; ++j) {
      double d = ps[0][j];
      for (int i = 1; i < mdeg; ++i) d += ps[i][j];
      D[j] = d != 0.0 ? d : 1e-5;
    }
    for (int i = 0; i < mdeg; ++i) qo[i] = qi[i] = 0.0;
    for (int i = 0; i < mdeg; ++i)
      for (int j = 0; j < mdeg; ++j) {
        if (i == j) continue;
        double a = ps[i][j];
        double b = r[j] * (ps[i][j] / D[j]);
        double g = floor(b < a ? b : a);
        qo[i] += g; qi[j] += g;
      }
    for (int i = 0; i < mdeg; ++i) { if (!(qo[i] > 0.0)) qo[i] = 0.0; if (!(qi[i] > 0.0)) qi[i] = 0.0; }
  }
  for (int i = 0; i < mdeg; ++i) { /* update_links, node.py:146-162 */
    int li = m->slot_in_link[s0 + i], lo = m->slot_out_link[s0 + i];
    H(o, F_OUT, li)[t] = qo[i];
    H(o, F_CO, li)[t] = H(o, F_CO, li)[t - 1] + qo[i];
    H(o, F_IN, lo)[t] = qi[i];
    H(o, F_CI, lo)[t] = H(o, F_CI, lo)[t - 1] + qi[i];
  }
}

static void link_density(pedn_oracle* o, int l, int t) { /* link.py:133-136 */
  float* N = G(o, G_N, l);
  double d = H(o, F_IN, l)[t] - H(o, F_OUT, l)[t];
  N[t] = (float)((double)N[t - 1] + d);
  if (o->m.link_sep[l] && o->sepnp[l] != 0.0) G(o, G_K, l)[t] = (float)((double)N[t] / area_of(o, l)); /* np.float32 / np.float64 */
  else G(o, G_K, l)[t] = N[t] / (float)area_of(o, l);
}

static void link_speed(pedn_oracle* o, int l, int t) { /* link.py:141-188 + functions.py:112-134 */
  const pedn_model_desc* m = &o->m;
  double vf = m->link_vf[l], kc = m->link_kc[l], kj = m->link_kj[l], len = m->link_length[l];
  float ks = G(o, G_K, l)[t];
  float ke = m->link_sep[l] ? ks : ks + (float)m->link_bi[l] * G(o, G_K, m->link_rev[l])[t];
  int is64;       /* 1: speed is a Python float (binary64) at this point, 0: np.float32 */
  double v64 = 0; float v32 = 0;
  int fd = m->link_fd[l];
  if (fd == 2 && ke <= (float)kc) { /* smulders free branch stays float32 */
    v32 = (float)vf * (1.0f - ke / (float)kj); is64 = 0;
  } else if (ke <= (float)kc) {
    v64 = vf; is64 = 1;
  } else {
    if (fd == 0) v32 = (float)((kc * vf) / (kj - kc)) * ((float)kj / ke - 1.0f);
    else if (fd == 1) v32 = ((float)(-vf) * (ke - (float)kj)) / (float)(kj - kc);
    else v32 = (float)(vf * kc) * (1.0f / ke - (float)(1 / kj));
    is64 = 0;
    if (!(v32 > 0.0f)) { v64 = 0.0; is64 = 1; } /* Python max(0, x) returns the int 0 */
  }
  if (m->link_noise[l] > 0.0) { /* functions.py:132-133 */
    double nz = 0.0;
    if (o->mode != PEDN_RNG_MEANFIELD) {
      rng_key key = {o->seed, o->replica, (uint32_t)l, (uint32_t)t, 3};
      nz = m->link_noise[l] * rng_z(&key);
    }
    if (is64) v64 = v64 + nz; else v32 = v32 + (float)nz;
  }
  if (is64) { if (!(v64 > 0.0)) { v64 = 0.0; } }
  else if (!(v32 > 0.0f)) { v64 = 0.0; is64 = 1; }
  float spd = is64 ? (float)v64 : v32;
  float tt;
  if (is64) tt = v64 > 0.0 ? (float)(len / v64) : (float)(len / 0.05);
  else tt = (float)len / v32; /* v32 > 0 here */
  G(o, G_V, l)[t] = spd;
  G(o, G_TT, l)[t] = tt;
  G(o, G_LF, l)[t] = G(o, G_K, l)[t] * spd;
  float rs = o->rsum[l] + tt;
  if (t >= m->window) {
    rs = rs - G(o, G_TT, l)[t - m->window];
    G(o, G_ATT, l)[t] = rs / (float)m->window;
  }
  o->rsum[l] = rs;
  H(o, F_GATE, l)[t] = m->link_sep[l] ? o->sepw[l] : o->back[l];
}

int pedn_oracle_step(pedn_oracle* o, int t) {
  const pedn_model_desc* m = &o->m;
  if (t < 1 || t > m->T) return PEDN_E_ARG;
  for (int n = 0; n < m->n_nodes; ++n) node_step(o, n, t);
  for (int l = 0; l < m->n_links; ++l) link_density(o, l, t);
  for (int l = 0; l < m->n_links; ++l) link_speed(o, l, t);
  return (int)o->flags;
}

int pedn_oracle_run(pedn_oracle* o, int t0, int t1) {
  for (int t = t0; t < t1; ++t) pedn_oracle_step(o, t);
  return (int)o->flags;
}

void pedn_oracle_reseed(pedn_oracle* o, uint64_t seed, int32_t replica) { o->seed = seed; o->replica = (uint32_t)replica; }

/* several independent replicas on host threads (cpu_baseline leg of bench.py) */
int pedn_oracle_run_many(pedn_oracle** os, int n, int t0, int t1) {
  int flags = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : flags)
  for (int i = 0; i < n; ++i) flags |= pedn_oracle_run(os[i], t0, t1);
  return flags;
}

/* ------------------------------------------------------------------------------------------------ accessors */
void pedn_oracle_set_demand(pedn_oracle* o, int node, const double* v, int n) {
  int row = o->m.node_demand_row[node];
  if (row < 0) return;
  double* d = o->demand + (size_t)row * o->T1;
  memset(d, 0, sizeof(double) * o->T1);
  memcpy(d, v, sizeof(double) * (size_t)(n < o->T1 ? n : o->T1));
}
void pedn_oracle_set_od_weights(pedn_oracle* o, int od, const double* v, int n) {
  memcpy(o->od_w + (size_t)od * o->T1, v, sizeof(double) * (size_t)(n < o->T1 ? n : o->T1));
}
void pedn_oracle_set_width(pedn_oracle* o, int which, int link, double v) {
  (which == PEDN_W_FRONT ? o->front : which == PEDN_W_BACK ? o->back : which == PEDN_W_SEP ? o->sepw : o->sepnp)[link] = v;
}
void pedn_oracle_set_tf(pedn_oracle* o, int node, const double* tf, int n) {
  memcpy(o->tf + o->m.node_turn_ptr[node], tf, sizeof(double) * (size_t)n);
}
const double* pedn_oracle_tf(pedn_oracle* o) { return o->tf; }
const void* pedn_oracle_field(pedn_oracle* o, int field) {
  return field < 7 ? (const void*)o->f64[field] : (const void*)o->f32[field - 7];
}
uint32_t pedn_oracle_flags(pedn_oracle* o) { return o->flags; }
void pedn_oracle_tally(pedn_oracle* o, uint64_t out[5]) { memcpy(out, o->tally, sizeof(o->tally)); }
float pedn_oracle_powf(float x, float y) { return pw_powf(x, y); }
double pedn_oracle_exp(double x) { return pw_exp(x); }
void pedn_oracle_philox(uint32_t ctr[4], uint32_t k0, uint32_t k1) { philox4x32_10(ctr, k0, k1); }
int64_t pedn_oracle_binomial(int64_t n, double p, uint64_t seed, uint32_t replica, uint32_t link, uint32_t t, uint32_t site) {
  rng_key k = {seed, replica, link, t, site};
  return rng_binomial(n, p, &k, PEDN_RNG_PHILOX);
}
double pedn_oracle_normal(double sigma, uint64_t seed, uint32_t replica, uint32_t link, uint32_t t) {
  rng_key k = {seed, replica, link, t, 3};
  return sigma * rng_z(&k);
}
