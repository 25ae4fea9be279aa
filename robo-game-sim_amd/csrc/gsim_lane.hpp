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

template <int N>
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
  for (int i = 1; i <= kGrid; ++i) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double coef = e == PK ? 0.2 : 1.0;
      const double tau = e == PK ? p.tau_puck : p.tau_player;
      cx[e] = rk4_axis(cx[e], ux[e], coef, tau, dt_col);
      cy[e] = rk4_axis(cy[e], uy[e], coef, tau, dt_col);
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
        watch |= !(dx * dx + dy * dy > r2);
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
