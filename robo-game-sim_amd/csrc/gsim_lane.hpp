// gsim_lane.hpp -- one LANE owns one arena: the streaming pass of the hybrid stepper.
//
// In almost every step of open play nothing touches anything: the reference fills its
// 51-column grid (populateStateGrid, src/GameSim.cpp:173-217), insertCollisions
// (cpp:360-511) finds no overlap in any column, the event list stays empty and the step
// result is simply grid column 50 (cpp:153-157).  For those steps the grid never needs to
// exist: a lane integrates all discs of its arena one sub-step at a time in registers and
// only watches for an overlap.
//
// The overlap watch is CONSERVATIVE: it fires on a superset of the reference's tests
// (cpp:374,397,420,443,471), using squared distances and slightly widened thresholds, so
// it costs a handful of multiply/compare per disc pair and no sqrt; the pair tests run only
// every few sub-steps, against a radius widened by the travel possible in between (see
// lane_step).  A lane whose watch fired commits nothing; its arena is re-run from the step's start state by the
// wave-per-arena pass (gsim_wave.hpp), which implements the reference's event loop
// exactly.  A lane whose watch stayed silent has, by construction, computed precisely what
// the reference computes for an event-free step: the same RK4 chain (same rk4_axis code),
// column 50, t_ += dt_.
#pragma once
#include "gsim_common.hpp"

namespace gsim {

// Widening of the watch thresholds.  The reference's wall test is fl(p + r) >= W and its pair
// test fl(sqrt(fl(dx^2+dy^2))) <= R; both are implied by the widened forms below with many
// orders of magnitude to spare (rounding moves these quantities by ~1e-16 relative).
constexpr double kWatchSlack = 1.0e-9;

// How the lane pass divides by the taus (chosen per launch by the host, see GSIM_DIV below).
enum DivMode : int { DIV_IEEE = 0, DIV_MARKSTEIN = 1, DIV_PLAYER_POW2 = 2, DIV_POW2_ONESTEP = 3 };

// One RK4 sub-step for M independent chains, written stage-major: every statement below is a
// loop over the chains, so consecutive instructions belong to different chains and a single
// wave per SIMD (65 536 arenas = 1024 waves = one per SIMD) issues at its ~5.2 cycles per FP64
// instruction instead of waiting 8 on each dependent one (tools/ubench/fp64rate.hip).  The compiler's
// default scheduler undoes the interleaving; the 1v1/2v2 objects are built with max-ilp (Makefile).
// Arithmetic per chain is exactly rk4_axis<DM>.

// The batch is chains [C0, C0 + M) of the arena's arrays; chains >= PKC belong to the puck.
template <int MODE, int C0, int M, int PKC>
__device__ __forceinline__ void rk4_chains(double *__restrict__ p_all, double *__restrict__ v_all,
                                           const double *__restrict__ u_all, double tau_pl, double inv_pl,
                                           double tau_pk, double inv_pk, double h) {
  const double half = h / 2.0, w = h / 6.0;
  // Power-of-two tau_player (DIV_PLAYER_POW2, DIV_POW2_ONESTEP): k = d * 2^j with d = u - v is exact, and
  // scaling by 2^j commutes with every rounding that follows (the magnitudes are inside the division
  // guard's window, nothing is subnormal), so the factor moves into the step constants -- themselves
  // exact multiples, host-checked -- and the players' four multiplications per sub-step disappear:
  // fl(k*(h/2)) = fl(d*(2^j*h/2)), ((k1 + 2*k2) + 2*k3) + k4 = 2^j*(((d1 + 2*d2) + 2*d3) + d4) bit for bit.
  constexpr bool FOLD = MODE == DIV_PLAYER_POW2 || MODE == DIV_POW2_ONESTEP;
  const double half_pl = FOLD ? half * inv_pl : half, h_pl = FOLD ? h * inv_pl : h, w_pl = FOLD ? w * inv_pl : w;
  double *p = p_all + C0, *v = v_all + C0;
  const double *u = u_all + C0;
  double a[M], q[M], r[M], accv[M], accp[M];
#define GSIM_EACH for (int c = 0; c < M; ++c)
#define GSIM_TAU(c) ((c) + C0 >= PKC ? tau_pk : tau_pl)
#define GSIM_INV(c) ((c) + C0 >= PKC ? inv_pk : inv_pl)
#define GSIM_COEF(c) ((c) + C0 >= PKC ? 0.2 : 1.0)
#define GSIM_IS_PK(c) ((c) + C0 >= PKC)
#define GSIM_HALF(c) (GSIM_IS_PK(c) ? half : half_pl)
#define GSIM_H(c) (GSIM_IS_PK(c) ? h : h_pl)
#define GSIM_W(c) (GSIM_IS_PK(c) ? w : w_pl)
// numerator of f's acceleration row: players u - v (cpp:66-67); puck 0 - 0.2*v (cpp:94-95, u is always 0),
// written as a negation: same value (only the sign of a zero can differ), and the negation is a free
// source modifier where the subtraction from 0.0 is an instruction.
#define GSIM_NUM(c, x) (GSIM_IS_PK(c) ? -(0.2 * (x)) : (u[c] - (x)))
// Division of every chain's a[] by its tau, by compile-time mode:
//   DIV_IEEE      IEEE '/' everywhere (taus outside div_tau<1>'s domain, or forced)
//   DIV_MARKSTEIN div_tau<1>'s five steps for every chain
//   DIV_PLAYER_POW2  tau_player is +-2^k (the default 0.5): a/tau = a*(1/tau) exactly -- scaling by a
//                 power of two is exact and rounds like '/' even into the denormals -- so the 2P
//                 player chains need one multiply; the puck chains use the five steps.
//   DIV_POW2_ONESTEP the same, and tau_puck is host-certified for div_tau<3>: three steps for the puck.
#define GSIM_DIV_MARK(COND)                                                                          \
  _Pragma("unroll") GSIM_EACH if (COND) q[c] = a[c] * GSIM_INV(c);                                    \
  _Pragma("unroll") GSIM_EACH if (COND) r[c] = __builtin_fma(-GSIM_TAU(c), q[c], a[c]);               \
  _Pragma("unroll") GSIM_EACH if (COND) q[c] = __builtin_fma(r[c], GSIM_INV(c), q[c]);                \
  _Pragma("unroll") GSIM_EACH if (COND) r[c] = __builtin_fma(-GSIM_TAU(c), q[c], a[c]);               \
  _Pragma("unroll") GSIM_EACH if (COND) q[c] = __builtin_fma(r[c], GSIM_INV(c), q[c]);
#define GSIM_DIV_ONE(COND)                                                                           \
  _Pragma("unroll") GSIM_EACH if (COND) q[c] = a[c] * GSIM_INV(c);                                    \
  _Pragma("unroll") GSIM_EACH if (COND) r[c] = __builtin_fma(-GSIM_TAU(c), q[c], a[c]);               \
  _Pragma("unroll") GSIM_EACH if (COND) q[c] = __builtin_fma(r[c], GSIM_INV(c), q[c]);
#define GSIM_DIV()                                                                                    \
  if (MODE == DIV_POW2_ONESTEP) {                                                                     \
    _Pragma("unroll") GSIM_EACH if (!GSIM_IS_PK(c)) q[c] = a[c];   /* d, the 2^j is in the constants */ \
    GSIM_DIV_ONE(GSIM_IS_PK(c))                                                                       \
  } else if (MODE == DIV_PLAYER_POW2) {                                                               \
    _Pragma("unroll") GSIM_EACH if (!GSIM_IS_PK(c)) q[c] = a[c];                                      \
    GSIM_DIV_MARK(GSIM_IS_PK(c))                                                                      \
  } else if (MODE == DIV_MARKSTEIN) {                                                                 \
    GSIM_DIV_MARK(true)                                                                               \
  } else {                                                                                            \
    _Pragma("unroll") GSIM_EACH q[c] = a[c] / GSIM_TAU(c);                                            \
  }
  // k1 = f(x)
#pragma unroll
  GSIM_EACH a[c] = GSIM_NUM(c, v[c]);
  GSIM_DIV();
#pragma unroll
  GSIM_EACH { accv[c] = q[c]; accp[c] = v[c]; }                       // k1v, k1p
  // k2 = f(x + k1*h/2)
#pragma unroll
  GSIM_EACH r[c] = v[c] + q[c] * GSIM_HALF(c);                                // x2v (= k2p)
#pragma unroll
  GSIM_EACH { accp[c] = __builtin_fma(r[c], 2.0, accp[c]); a[c] = GSIM_NUM(c, r[c]); }
  GSIM_DIV();
#pragma unroll
  GSIM_EACH accv[c] = __builtin_fma(q[c], 2.0, accv[c]);
  // k3 = f(x + k2*h/2)
#pragma unroll
  GSIM_EACH r[c] = v[c] + q[c] * GSIM_HALF(c);                                // x3v (= k3p)
#pragma unroll
  GSIM_EACH { accp[c] = __builtin_fma(r[c], 2.0, accp[c]); a[c] = GSIM_NUM(c, r[c]); }
  GSIM_DIV();
#pragma unroll
  GSIM_EACH accv[c] = __builtin_fma(q[c], 2.0, accv[c]);
  // k4 = f(x + k3*h)
#pragma unroll
  GSIM_EACH r[c] = v[c] + q[c] * GSIM_H(c);                                // x4v (= k4p)
#pragma unroll
  GSIM_EACH { accp[c] = accp[c] + r[c]; a[c] = GSIM_NUM(c, r[c]); }
  GSIM_DIV();
#pragma unroll
  GSIM_EACH { p[c] = p[c] + accp[c] * w; v[c] = v[c] + (accv[c] + q[c]) * GSIM_W(c); }   // cpp:87
#undef GSIM_DIV
#undef GSIM_DIV_MARK
#undef GSIM_DIV_ONE
#undef GSIM_W
#undef GSIM_H
#undef GSIM_HALF
#undef GSIM_IS_PK
#undef GSIM_NUM
#undef GSIM_COEF
#undef GSIM_INV
#undef GSIM_TAU
#undef GSIM_EACH
}

// An upper bound of sqrt(x) for the watch's speed bounds, x >= 0 and far below FLT_MAX: single-precision square root
// (1 ulp) of x rounded to float, the three roundings (< 2^-22 relative together) covered by the factor; the
// addend covers arguments that round to zero as a float.  Four instructions where the double-precision root is ~30.
__device__ __forceinline__ double sqrt_upper(double x) {
#if GSIM_FLOAT_BOUND
  return (double)__builtin_amdgcn_sqrtf((float)x) * (1.0 + 0x1p-20) + 1.0e-18;
#else
  return __builtin_sqrt(x) * (1.0 + 1.0e-12);
#endif
}

// All M chains of an arena, at most 10 at a time (more would spill registers at 4v4 and up).
template <int MODE, int M, int PKC, int C0 = 0>
__device__ __forceinline__ void rk4_all_chains(double *p, double *v, const double *u, double tau_pl, double inv_pl,
                                               double tau_pk, double inv_pk, double h) {
  constexpr int BATCH = GSIM_CHAIN_BATCH > 0 ? GSIM_CHAIN_BATCH : 10;
  constexpr int NB = (M + BATCH - 1) / BATCH, SZ = (M + NB - 1) / NB;
  constexpr int MB = (M - C0) < SZ ? (M - C0) : SZ;
  rk4_chains<MODE, C0, MB, PKC>(p, v, u, tau_pl, inv_pl, tau_pk, inv_pk, h);
  if constexpr (C0 + MB < M) rk4_all_chains<MODE, M, PKC, C0 + MB>(p, v, u, tau_pl, inv_pl, tau_pk, inv_pk, h);
}

// Chains are numbered c = 2*e + axis: the puck's two chains come last (c >= 2*P).
// Returns true when the arena must take the wave pass; discmask then holds the discs whose watch
// fired (all discs when the division guard tripped).
template <int N, int MODE, bool COH = false>
__device__ __forceinline__ bool lane_step(const StepArgs &p, const double *__restrict__ ctl, long long a,
                                          unsigned &discmask) {
  constexpr int P = 2 * N, E = P + 1, PK = P, M = 2 * E;
  const long long B = p.B;
  const double TAS = gld_at<COH>(p.state, 0 * B + a, p.n_state), TBS = gld_at<COH>(p.state, 1 * B + a, p.n_state);
  if (!(TAS < p.win && TBS < p.win)) return false;          // decided: run() is a no-op, cpp:121-123

  double pos[M], vel[M], u[M];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int row = state_row<N>(e);
    pos[2 * e + 0] = gld_at<COH>(p.state, (long long)(row + 0) * B + a, p.n_state);
    pos[2 * e + 1] = gld_at<COH>(p.state, (long long)(row + 1) * B + a, p.n_state);
    vel[2 * e + 0] = gld_at<COH>(p.state, (long long)(row + 2) * B + a, p.n_state);
    vel[2 * e + 1] = gld_at<COH>(p.state, (long long)(row + 3) * B + a, p.n_state);
    if (e != PK) {
      u[2 * e + 0] = ctl[(long long)(2 * e + 0) * B + a];
      u[2 * e + 1] = ctl[(long long)(2 * e + 1) * B + a];
    } else {
      u[2 * e + 0] = 0.0; u[2 * e + 1] = 0.0;               // cpp:188
    }
  }

  const double dt_col = p.dt_col;
  bool guard = false;
  bool wd[E];                                               // per-disc watch (lane masks live in SGPRs)
#pragma unroll
  for (int e = 0; e < E; ++e) wd[e] = false;
  if (MODE != DIV_IEEE) {
#pragma unroll
    for (int c = 0; c < M; ++c) guard |= !fast_division_safe(u[c], vel[c]);
  }
  // Sparse pair watch: the wall tests run at every column, the (far more numerous) pair tests only at
  // every `we`-th (we = p.watch_every, 5 or 1).  During an event-free step every velocity stays
  // between its start value and its command (RK4 of the lag with dt_col/tau <= 0.1, host-checked), so
  // no coordinate moves more than 1.25*vb*dt_col per column; a pair overlap at a skipped column
  // therefore still shows at the next checked one if the radius is widened by the travel of we-1
  // columns (two discs, two axes).  Walls stay dense and tight: discs slide along walls for many
  // steps without touching, and a widened wall test would flag every one of those steps.
  const int we = p.watch_every;
  // Speed bound of a PAIR: the two largest per-disc bounds s1 >= s2, s_e = hypot(|vx|max, |vy|max) (rounded up), so
  // the distance of any pair changes by at most travel = 1.25*(we-1)*dt_col*(s1 + s2) between two checks -- never more
  // than the 2*sqrt(2)*vb of the per-axis bound vb it replaces, and typically 0.6 of it: nearly all of C2's false
  // alarms are pairs inside the shell between the true and the widened radius.
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const double cx = __builtin_fmax(__builtin_fabs(vel[2 * e]), __builtin_fabs(u[2 * e]));
    const double cy = __builtin_fmax(__builtin_fabs(vel[2 * e + 1]), __builtin_fabs(u[2 * e + 1]));
    const double se = sqrt_upper(__builtin_fma(cx, cx, cy * cy));
    const double lo = __builtin_fmin(s1, se);
    s1 = __builtin_fmax(s1, se);
    s2 = __builtin_fmax(s2, lo);
  }
  const double grow = 1.25 * (double)(we - 1) * dt_col * (s1 + s2) * (1.0 + 1.0e-12);   // pair distance: two discs, Euclidean
  const double xthr_pl = kX / 2.0 - kRPlayer - kWatchSlack, ythr_pl = kY / 2.0 - kRPlayer - kWatchSlack;
  const double xthr_pk = kX / 2.0 - kRPuck - kWatchSlack, ythr_pk = kY / 2.0 - kRPuck - kWatchSlack;
  const double rpp = 2.0 * kRPlayer + grow, rpk = kRPlayer + kRPuck + grow;
  const double r2_pp = rpp * rpp * (1.0 + kWatchSlack), r2_pk = rpk * rpk * (1.0 + kWatchSlack);
  // At 3v3 every pair gets its own radius, widened by ITS two discs' bounds: the same argument, pair by pair.
  // 8.6 % fewer flagged arena-steps and +4 % there; at 2v2 it also flags 8.4 % less but the lane pass pays more for
  // the ten extra live values than the wave pass saves (asynchronous pool -3.5 %, pool +1.4 %), at 7v7 the same idea
  // put the kernel into scratch.
  // (The two variants are kept as separate code on purpose: the other team sizes' objects stay byte for byte what
  // they were -- at these register counts even dead code moved the 2v2 kernels by +-3 %.)
  constexpr bool PAIR_R = GSIM_PAIR_BOUNDS && N == 3;
  constexpr int NPAIR = PAIR_R ? E * (E - 1) / 2 : 1;
  double r2_pair[NPAIR];
  if constexpr (PAIR_R) {
    const double grow_unit = 1.25 * (double)(we - 1) * dt_col * (1.0 + 1.0e-12);
    double sd[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double cx = __builtin_fmax(__builtin_fabs(vel[2 * e]), __builtin_fabs(u[2 * e]));
      const double cy = __builtin_fmax(__builtin_fabs(vel[2 * e + 1]), __builtin_fabs(u[2 * e + 1]));
      sd[e] = sqrt_upper(__builtin_fma(cx, cx, cy * cy));
    }
    int q = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int o = e + 1; o < E; ++o) {
        const double r = ((e == PK || o == PK) ? kRPlayer + kRPuck : 2.0 * kRPlayer) + grow_unit * (sd[e] + sd[o]);
        r2_pair[q++] = r * r * (1.0 + kWatchSlack);
      }
    }
  }
  // Wall watch at the end points (p.wall_ends, same host check as the sparse pair watch).  A wall event
  // at column i needs the disc inside the wall zone at i while moving TOWARDS that wall at i-1
  // (cpp:374-392).  With v0 and u of one sign (or either zero) every stage velocity of every sub-step
  // keeps that sign, so the coordinate is monotone over the step: "in the zone and moving towards the
  // wall" at any column implies "in the zone" at column 50, which is the only column tested.  A chain
  // whose velocity may reverse (v0 and u of opposite signs) travels at most |v0|*dt in v0's direction
  // before it does (its speed only drops until then): it is flagged up front when that reaches the zone
  // of the wall it is heading for.  NaNs end up in column 50 and fail `<`.
  const bool wall_ends = p.wall_ends != 0;
  if (wall_ends) {
    const double reach = 1.001 * p.dt;                      // x |v0|: the speed only drops until the reversal
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double xthr = e == PK ? xthr_pk : xthr_pl, ythr = e == PK ? ythr_pk : ythr_pl;
      const double qx = pos[2 * e] + reach * vel[2 * e], qy = pos[2 * e + 1] + reach * vel[2 * e + 1];
      wd[e] |= (vel[2 * e] > 0.0 && u[2 * e] < 0.0 && !(qx < xthr)) || (vel[2 * e] < 0.0 && u[2 * e] > 0.0 && !(-qx < xthr)) ||
               (vel[2 * e + 1] > 0.0 && u[2 * e + 1] < 0.0 && !(qy < ythr)) ||
               (vel[2 * e + 1] < 0.0 && u[2 * e + 1] > 0.0 && !(-qy < ythr));
    }
  }
  int first_col = 0;                                        // first column of the first block with a flag
  for (int i = 0; i < kGrid; i += we) {
#pragma nounroll
    for (int j = 0; j < we; ++j) {
      rk4_all_chains<MODE, M, 2 * PK>(pos, vel, u, p.tau_player, p.inv_tau_player, p.tau_puck, p.inv_tau_puck, dt_col);
      if (!wall_ends) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double xthr = e == PK ? xthr_pk : xthr_pl, ythr = e == PK ? ythr_pk : ythr_pl;
          // NaN positions must not slip through as "no overlap": !(x < thr) is true for NaN
          wd[e] |= !(__builtin_fabs(pos[2 * e]) < xthr) || !(__builtin_fabs(pos[2 * e + 1]) < ythr);
        }
      }
    }
    if constexpr (PAIR_R) {
      int q = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int o = e + 1; o < E; ++o) {
          const double dx = pos[2 * e] - pos[2 * o], dy = pos[2 * e + 1] - pos[2 * o + 1];
          const bool near = !(__builtin_fma(dy, dy, dx * dx) > r2_pair[q++]);
          wd[e] |= near;
          wd[o] |= near;
        }
      }
    } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int o = e + 1; o < E; ++o) {
        const double r2 = (e == PK || o == PK) ? r2_pk : r2_pp;
        const double dx = pos[2 * e] - pos[2 * o], dy = pos[2 * e + 1] - pos[2 * o + 1];
        const bool near = !(__builtin_fma(dy, dy, dx * dx) > r2);   // the watch is conservative: fma is fine here
        wd[e] |= near;
        wd[o] |= near;
      }
    }
    }
    if (GSIM_LANE_HINT) {
      bool any = false;
#pragma unroll
      for (int e = 0; e < E; ++e) any |= wd[e];
      first_col = (first_col == 0 && any) ? i + 1 : first_col;
    }
  }
  if (wall_ends) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double xthr = e == PK ? xthr_pk : xthr_pl, ythr = e == PK ? ythr_pk : ythr_pl;
      wd[e] |= !(__builtin_fabs(pos[2 * e]) < xthr) || !(__builtin_fabs(pos[2 * e + 1]) < ythr);
    }
  }
  // discmask: bits 0..E-1 = discs whose watch fired, bits 16..21 = first column the wave pass has to
  // scan; the one-lane pass does not track it (measured: costs more here than the shorter scans save at
  // n <= 3), the two-lane pass does.  The division guard asks for everything.
  bool watch = guard;
  discmask = 0u;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    watch |= wd[e];
    discmask |= wd[e] ? (1u << e) : 0u;
  }
  discmask |= (unsigned)((GSIM_LANE_HINT && first_col >= 1) ? first_col : 1) << 16;
  if (guard) discmask = ((1u << E) - 1u) | (1u << 16);
  if (watch) return true;

#pragma unroll
  for (int e = 0; e < E; ++e) {                             // cpp:153-157
    const int row = state_row<N>(e);
    gst_at<COH>(p.state, (long long)(row + 0) * B + a, p.n_state, pos[2 * e + 0]);
    gst_at<COH>(p.state, (long long)(row + 1) * B + a, p.n_state, pos[2 * e + 1]);
    gst_at<COH>(p.state, (long long)(row + 2) * B + a, p.n_state, vel[2 * e + 0]);
    gst_at<COH>(p.state, (long long)(row + 3) * B + a, p.n_state, vel[2 * e + 1]);
  }
  gst_at<COH>(p.tcur, a, p.n_tcur, gld_at<COH>(p.tcur, a, p.n_tcur) + p.dt);                             // cpp:158
  return false;
}

// ---------------------------------------------------------------------------------------------
// Two lanes per arena: the even lane carries every disc's x chain, the odd lane the y chains.
//
// The chains of the two axes never interact inside an event-free step; only the overlap watch
// needs both (dx^2 + dy^2), which one DPP quad exchange per pair provides.  Same arithmetic per
// chain as lane_step, so the same bits.  What it buys: half the chains per lane, which is what lets the
// large teams fit at all (7v7: 15 chains per lane instead of 30) and, with the split grid, two arenas' wave
// passes side by side per wavefront.  It does NOT buy a second wavefront per SIMD: k_step_hybrid2 needs 274
// (2v2) to 256 + 242 (7v7) registers, one wavefront per SIMD like every other stepping kernel (65 536 arenas =
// 2048 wavefronts run as two successive waves over the chip's 1024 SIMDs).
__device__ __forceinline__ unsigned quad_xor1_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
}

template <int N, int MODE, bool COH = false>
__device__ __forceinline__ bool lane_step_split(const StepArgs &p, const double *__restrict__ ctl, long long a, int ax,
                                                unsigned &discmask) {
  constexpr int P = 2 * N, E = P + 1, PK = P;
  const long long B = p.B;
  const double TAS = gld_at<COH>(p.state, 0 * B + a, p.n_state), TBS = gld_at<COH>(p.state, 1 * B + a, p.n_state);
  if (!(TAS < p.win && TBS < p.win)) return false;          // decided (both lanes of the arena agree)

  double pos[E], vel[E], u[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int row = state_row<N>(e);
    pos[e] = gld_at<COH>(p.state, (long long)(row + ax) * B + a, p.n_state);
    vel[e] = gld_at<COH>(p.state, (long long)(row + 2 + ax) * B + a, p.n_state);
    u[e] = e != PK ? ctl[(long long)(2 * e + ax) * B + a] : 0.0;   // cpp:188
  }
  const double dt_col = p.dt_col;
  bool guard = false;
  bool wd[E];
#pragma unroll
  for (int e = 0; e < E; ++e) wd[e] = false;
  if (MODE != DIV_IEEE) {
#pragma unroll
    for (int e = 0; e < E; ++e) guard |= !fast_division_safe(u[e], vel[e]);
  }
  // sparse pair watch, see lane_step; vb must cover BOTH axes of the arena (pair distances mix them)
  const int we = p.watch_every;
  double s1 = 0.0, s2 = 0.0;                                // the two largest per-disc speed bounds, see lane_step
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const double c = __builtin_fmax(__builtin_fabs(vel[e]), __builtin_fabs(u[e]));
    const double cc = c * c;
    const double se = sqrt_upper(cc + quad_xor1(cc));       // both lanes of the arena: the same sum
    const double lo = __builtin_fmin(s1, se);
    s1 = __builtin_fmax(s1, se);
    s2 = __builtin_fmax(s2, lo);
  }
  const double grow = 1.25 * (double)(we - 1) * dt_col * (s1 + s2) * (1.0 + 1.0e-12);
  const double half_extent = ax == 0 ? kX / 2.0 : kY / 2.0;
  const double thr_pl = half_extent - kRPlayer - kWatchSlack, thr_pk = half_extent - kRPuck - kWatchSlack;
  const double rpp = 2.0 * kRPlayer + grow, rpk = kRPlayer + kRPuck + grow;
  const double r2_pp = rpp * rpp * (1.0 + kWatchSlack), r2_pk = rpk * rpk * (1.0 + kWatchSlack);
  const bool wall_ends = p.wall_ends != 0;                  // see lane_step
  bool wall_flag = false;                                   // a wall flag carries no column: scans start at 1
  if (wall_ends) {
    const double reach = 1.001 * p.dt;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double q = pos[e] + reach * vel[e], thr = e == PK ? thr_pk : thr_pl;
      const bool f = (vel[e] > 0.0 && u[e] < 0.0 && !(q < thr)) || (vel[e] < 0.0 && u[e] > 0.0 && !(-q < thr));
      wd[e] |= f;
      wall_flag |= f;
    }
  }
  int first_col = 0;
  for (int i = 0; i < kGrid; i += we) {
#pragma nounroll
    for (int j = 0; j < we; ++j) {
      rk4_all_chains<MODE, E, PK>(pos, vel, u, p.tau_player, p.inv_tau_player, p.tau_puck, p.inv_tau_puck, dt_col);
      if (!wall_ends) {
#pragma unroll
        for (int e = 0; e < E; ++e) wd[e] |= !(__builtin_fabs(pos[e]) < (e == PK ? thr_pk : thr_pl));   // NaN counts as overlap
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int o = e + 1; o < E; ++o) {
        const double r2 = (e == PK || o == PK) ? r2_pk : r2_pp;
        const double d = pos[e] - pos[o];
        const double sq = d * d;
        const bool near = !(sq + quad_xor1(sq) > r2);       // both lanes of the arena compute the same sum
        wd[e] |= near;
        wd[o] |= near;
      }
    }
    bool any = false;
#pragma unroll
    for (int e = 0; e < E; ++e) any |= wd[e];
    first_col = (first_col == 0 && any) ? i + 1 : first_col;
  }
  if (wall_ends) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool f = !(__builtin_fabs(pos[e]) < (e == PK ? thr_pk : thr_pl));
      wd[e] |= f;
      wall_flag |= f;
    }
  }
  unsigned bits = guard ? ((1u << E) - 1u) : 0u;
#pragma unroll
  for (int e = 0; e < E; ++e) bits |= wd[e] ? (1u << e) : 0u;
  bits |= quad_xor1_u32(bits);                              // a wall flag seen by either axis
  unsigned fc = (unsigned)(first_col < 1 ? kGrid : first_col);           // the earlier of the two lanes' first blocks
  const unsigned fo = quad_xor1_u32(fc);
  fc = fo < fc ? fo : fc;
  const bool from_one = guard || wall_flag;
  const unsigned other_from_one = quad_xor1_u32(from_one ? 1u : 0u);   // exchanged before any branch: both lanes active
  if (from_one || other_from_one != 0u) fc = 1u;
  discmask = bits | (fc << 16);
  if (bits != 0u) return true;

#pragma unroll
  for (int e = 0; e < E; ++e) {                             // cpp:153-157, this lane's axis
    const int row = state_row<N>(e);
    gst_at<COH>(p.state, (long long)(row + ax) * B + a, p.n_state, pos[e]);
    gst_at<COH>(p.state, (long long)(row + 2 + ax) * B + a, p.n_state, vel[e]);
  }
  if (ax == 0) gst_at<COH>(p.tcur, a, p.n_tcur, gld_at<COH>(p.tcur, a, p.n_tcur) + p.dt);                // cpp:158
  return false;
}

}  // namespace gsim
