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
// it costs a handful of multiply/compare per disc pair and no sqrt.  A lane whose watch
// fired commits nothing; its arena is re-run from the step's start state by the
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

// Magnitude window in which div_tau<true>'s residuals stay normal numbers (see its comment).
// u is held for the step; v only moves by factors >= 0.27 per sub-step (|R(z)| of RK4), so an
// entry check covers all 50 sub-steps: u == 0 or |u| >= 2^-700, and if u == 0 then v == 0 or
// |v| >= 2^-600; everything <= 2^500.  Anything else is handed to the exact wave pass.
__device__ __forceinline__ bool fast_division_safe(double u, double v) {
  const double lo_u = 0x1.0p-700, lo_v = 0x1.0p-600, hi = 0x1.0p500;
  const double au = __builtin_fabs(u), av = __builtin_fabs(v);
  const bool u_ok = (u == 0.0) || (au >= lo_u);
  const bool v_ok = (au >= lo_u) || (v == 0.0) || (av >= lo_v);
  return u_ok && v_ok && (au <= hi) && (av <= hi);
}

template <int N, bool FAST>
__device__ __forceinline__ bool lane_step(const StepArgs &p, const double *__restrict__ ctl, long long a) {
  constexpr int P = 2 * N, E = P + 1, PK = P;
  const long long B = p.B;
  const double TAS = p.state[0 * B + a], TBS = p.state[1 * B + a];
  if (!(TAS < p.win && TBS < p.win)) return false;          // decided: run() is a no-op, cpp:121-123

  Chain cx[E], cy[E];
  double ux[E], uy[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int row = state_row<N>(e);
    cx[e].p = p.state[(long long)(row + 0) * B + a];
    cy[e].p = p.state[(long long)(row + 1) * B + a];
    cx[e].v = p.state[(long long)(row + 2) * B + a];
    cy[e].v = p.state[(long long)(row + 3) * B + a];
    if (e != PK) {
      ux[e] = ctl[(long long)(2 * e + 0) * B + a];
      uy[e] = ctl[(long long)(2 * e + 1) * B + a];
    } else {
      ux[e] = 0.0; uy[e] = 0.0;                             // cpp:188
    }
  }

  const double dt_col = p.dt_col;
  bool watch = false;
  if (FAST) {
#pragma unroll
    for (int e = 0; e < E; ++e)
      watch |= !fast_division_safe(ux[e], cx[e].v) || !fast_division_safe(uy[e], cy[e].v);
  }
  for (int i = 1; i <= kGrid; ++i) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double coef = e == PK ? 0.2 : 1.0;
      const double tau = e == PK ? p.tau_puck : p.tau_player;
      const double inv = e == PK ? p.inv_tau_puck : p.inv_tau_player;
      cx[e] = rk4_axis<FAST>(cx[e], ux[e], coef, tau, inv, dt_col);
      cy[e] = rk4_axis<FAST>(cy[e], uy[e], coef, tau, inv, dt_col);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double rad = e == PK ? kRPuck : kRPlayer;
      const double xthr = kX / 2.0 - rad - kWatchSlack, ythr = kY / 2.0 - rad - kWatchSlack;
      // NaN positions must not slip through as "no overlap": !(x < thr) is true for NaN
      watch |= !(__builtin_fabs(cx[e].p) < xthr) || !(__builtin_fabs(cy[e].p) < ythr);
#pragma unroll
      for (int o = e + 1; o < E; ++o) {
        const double rsum = rad + (o == PK ? kRPuck : kRPlayer);
        const double r2 = rsum * rsum * (1.0 + kWatchSlack);
        const double dx = cx[e].p - cx[o].p, dy = cy[e].p - cy[o].p;
        watch |= !(__builtin_fma(dy, dy, dx * dx) > r2);     // the watch is conservative: fma is fine here
      }
    }
  }
  if (watch) return true;

#pragma unroll
  for (int e = 0; e < E; ++e) {                             // cpp:153-157
    const int row = state_row<N>(e);
    p.state[(long long)(row + 0) * B + a] = cx[e].p;
    p.state[(long long)(row + 1) * B + a] = cy[e].p;
    p.state[(long long)(row + 2) * B + a] = cx[e].v;
    p.state[(long long)(row + 3) * B + a] = cy[e].v;
  }
  p.tcur[a] = p.tcur[a] + p.dt;                             // cpp:158
  return false;
}

}  // namespace gsim
