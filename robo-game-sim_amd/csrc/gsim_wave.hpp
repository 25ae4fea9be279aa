// gsim_wave.hpp -- one lane-group (GW = 16/20/32/64 lanes of one wavefront) owns one arena.
//
// The whole step of the reference (GameSim::updateSim, src/GameSim.cpp:126-171) for one
// arena, with the 51-column state grid (state_grid_, include/GameSim.h:46) staged in LDS
// (GridView: all of it, or from 6v6 up and at 4v4 only the positions, velocities in L2):
//
//   fill      lanes (disc, axis) integrate their own RK4 chain down the 50 sub-steps
//             (populateStateGrid, cpp:173-217) and write the columns to LDS;
//   scan      lane <-> grid column: every lane runs the wall / disc-pair overlap tests of
//             one column (insertCollisions, cpp:360-511); a ballot over the group picks
//             the first column with an admissible event;
//   resolve   the earliest event is carried out by the group's first quad of lanes, which take
//             over the affected (disc, axis) chains and exchange partner values by DPP
//             quad_perm (carryOutFirstCollision, cpp:219-358, correctOverlap cpp:581-623);
//   list      the live events sit in registers, one slot per disc (an event's i_ is unique
//             in the list), replicated across the group; "stable sort descending, pop
//             the back" (cpp:227,510, Collision.h:24-30) becomes "minimum time, ties to
//             the most recently inserted".
//
// Control flow is uniform inside a group, so shuffles and ballots only ever pair lanes
// that are active together; different groups of one wave may diverge freely.
#pragma once
#include "gsim_common.hpp"
#include "gsim_rng.hpp"

namespace gsim {

// Exchanges inside an aligned quad of lanes by DPP quad_perm (no LDS round trip).
template <int CTRL>
__device__ __forceinline__ double quad_perm(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double quad_xor1(double x) { return quad_perm<0xB1>(x); }   // lanes [1,0,3,2]
__device__ __forceinline__ double quad_xor2(double x) { return quad_perm<0x4E>(x); }   // lanes [2,3,0,1]
__device__ __forceinline__ double quad_lane1(double x) { return quad_perm<0x55>(x); }  // lanes [1,1,1,1]

// Magnitude window in which div_tau<true>'s residuals stay normal numbers (see its comment).
// u is held for the step; v only moves by factors >= 0.27 per sub-step (|R(z)| of RK4), so one
// check covers the <= 52 sub-steps that follow it: u == 0 or |u| >= 2^-700, and if u == 0 then
// v == 0 or |v| >= 2^-600; everything <= 2^500.
__device__ __forceinline__ bool fast_division_safe(double u, double v) {
  const double lo_u = 0x1.0p-700, lo_v = 0x1.0p-600, hi = 0x1.0p500;
  const double au = __builtin_fabs(u), av = __builtin_fabs(v);
  const bool u_ok = (u == 0.0) || (au >= lo_u);
  const bool v_ok = (au >= lo_u) || (v == 0.0) || (av >= lo_v);
  return u_ok && v_ok && (au <= hi) && (av <= hi);
}

// rk4_axis with the division flavour chosen per chain at run time.  FAST: 0 = IEEE only, 1 = two
// correction steps, 3 = one (host-certified taus); `fast` = this chain's magnitudes are in the window.
template <int FAST>
__device__ __forceinline__ Chain rk4_dyn(Chain x, double u, double coef, double tau, double inv, double h, bool fast) {
  if (FAST != 0) {
    if (fast) return rk4_axis<FAST>(x, u, coef, tau, inv, h);
  }
  return rk4_axis<0>(x, u, coef, tau, inv, h);
}

// rk4_axis_noise with the division flavour chosen at run time like rk4_dyn (device-generator kernel).  The same window
// applies: for the puck (no noise, its velocity decays) fast_division_safe covers the sub-steps of one ten-column block; a
// player's velocity, re-drawn at every stage, cannot come closer to zero than ~sigma * 2^-130 without being zero, and the
// host only selects FAST != 0 for 2^-300 <= sigma <= 2^300 (launch_steps), so the check at the block's first column holds
// for all of it.
template <int FAST>
__device__ __forceinline__ Chain rk4_noise_dyn(Chain x, double u, double coef, double tau, double h, double sigma,
                                               const double *nz, double inv, bool fast) {
#ifdef GSIM_NOISE_NO_GUARD                                   // A/B only: what the run-time choice costs
  return rk4_axis_noise<FAST>(x, u, coef, tau, h, sigma, nz, inv);
#endif
  if (FAST != 0) {
    if (fast) return rk4_axis_noise<FAST>(x, u, coef, tau, h, sigma, nz, inv);
  }
  return rk4_axis_noise<0>(x, u, coef, tau, h, sigma, nz);
}

template <int GW>
__device__ __forceinline__ unsigned long long group_ballot(bool pred, int gbase) {
  unsigned long long m = __ballot(pred);
  if (GW < 64) m = (m >> gbase) & ((1ull << (GW & 63)) - 1ull);
  return m;
}

// Column stride of the LDS grid in doubles: odd, so the "lane <-> column" reads of scan()
// fall on distinct bank pairs.
template <int N>
struct WaveLayout {
  static constexpr int P = 2 * N, E = P + 1, PK = P;
  static constexpr int CS = 4 * E + 1;
  static constexpr int U_OFF = kCols * CS;          // controls: U[e][axis]
  static constexpr int DOUBLES = U_OFF + 2 * E + 1; // per arena (odd => groups shift banks)
};

#ifndef GSIM_DEBUG_BOUNDS
#define GSIM_DEBUG_BOUNDS 0
#endif
// Debug builds (make EXTRA=-DGSIM_DEBUG_BOUNDS=1): every grid access checks its indices; violations are counted
// in the diagnostic stamp area (rgs_get_stamps: [0] bad vel() indices, [1] the last one, [2]/[3] the same for
// pos()) and redirected to element 0.  GPU AddressSanitizer is not available on this pool.
#if GSIM_DEBUG_BOUNDS
#define GSIM_GRID_DBG(p) , (p).stamp_out
#else
#define GSIM_GRID_DBG(p)
#endif

// The 51-column grid of one arena.  SPLITV = false: positions and velocities side by side in LDS
// (WaveLayout).  SPLITV = true (two-lane hybrid from 6v6 up): LDS keeps the positions only -- every
// scan reads them, for every disc, at every column -- and the velocity rows, which are read only when a
// test fires (back-tracking, cpp:385-392, 482-494) and at the start of a carry-out (cpp:255, 309), live
// in a per-block global scratch that stays in L2.  Half the LDS per arena = twice the arenas re-run
// side by side under the same 40 KB-per-block budget.
template <int N, bool SPLITV>
struct GridView {
  static constexpr int E = 2 * N + 1;
  static constexpr int CS = SPLITV ? 2 * E + 1 : 4 * E + 1;          // odd: lane<->column reads hit distinct banks
  static constexpr int U_OFF = kCols * CS;
  static constexpr int LDS_DOUBLES = U_OFF + 2 * E + 1;
  static constexpr int GV_DOUBLES = SPLITV ? kCols * 2 * E : 0;
  double *lds;
  double *gv;
#if GSIM_DEBUG_BOUNDS
  unsigned long long *dbg;
#endif
  __device__ __forceinline__ double &pos(int g, int e, int ax) const {
#if GSIM_DEBUG_BOUNDS
    if (g < 0 || g > kGrid || e < 0 || e >= E || ax < 0 || ax > 1) {
      atomicAdd(&dbg[2], 1ull);
      dbg[3] = ((unsigned long long)(unsigned)g << 32) | ((unsigned long long)(unsigned)(e & 0xFFFF) << 16) | (unsigned)(ax & 0xFFFF);
      g = 0; e = 0; ax = 0;
    }
#endif
    return lds[g * CS + (SPLITV ? 2 : 4) * e + ax];
  }
  __device__ __forceinline__ double &vel(int g, int e, int ax) const {
#if GSIM_DEBUG_BOUNDS                                       // software bounds check of the grid indices (debug builds only)
    if (g < 0 || g > kGrid || e < 0 || e >= E || ax < 0 || ax > 1) {
      atomicAdd(&dbg[0], 1ull);
      dbg[1] = ((unsigned long long)(unsigned)g << 32) | ((unsigned long long)(unsigned)(e & 0xFFFF) << 16) | (unsigned)(ax & 0xFFFF);
      g = 0; e = 0; ax = 0;
    }
#endif
    // (Round 2 clamped these indices after a device fault whose cause was unknown.  The cause was found in round 3: a
    // hipcc defect that left kernarg OFFSETS stale -- not a grid index -- see DESIGN.md 3.4; tools/isa_lint.py now
    // rejects any build that shows it, and the clamp, which could only have hidden a wrong index, is gone.)
    if (SPLITV) return gv[g * (2 * E) + 2 * e + ax];
    return lds[g * CS + 4 * e + 2 + ax];
  }
  __device__ __forceinline__ double &ctl(int e, int ax) const { return lds[U_OFF + 2 * e + ax]; }
};

// wave_sync that also orders this wave's global-memory traffic (the split grid's velocity rows): a
// workgroup-scope fence waits for the outstanding stores; all waves of a workgroup share the CU's L1.
template <bool SPLITV>
__device__ __forceinline__ void grid_sync() {
  if (SPLITV) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    wave_sync();
  }
}

template <int N>
__device__ __forceinline__ double disc_radius(int e) { return e == 2 * N ? kRPuck : kRPlayer; }

// One column of insertCollisions (cpp:371-498) for disc `id` (may differ from lane to lane): the earliest
// candidate among UP, DOWN, RIGHT, LEFT and the other discs in entity order; strict '<' keeps the
// first found.  The pair test fl(sqrt(s)) <= R (cpp:471) is evaluated as s <= le, le = the largest
// double whose correctly rounded square root is <= R (host-computed, StepArgs::le_*): same truth
// value, no sqrt.
//
// Written for latency, since one lane group runs this inside a serial chain of up to 49 events: the
// overlap predicates of the column (4 wall zones, E-1 distances) are evaluated first in straight-line
// code over one batch of LDS loads; only a column with a predicate set -- rare outside sustained
// contact -- goes on to the back-tracking arithmetic.
// A pair's back-tracked time (cpp:482-494) is the same from either side (all differences change sign together, every product
// is unchanged), so in the first scans, where both discs of a near pair are evaluated at the same column, the second
// one takes it from here.
struct PairMemo {
  int a, b;                  // the pair last evaluated by this lane at the current column: disc a scanned, partner b
  bool valid;                // it had a candidate (dv.dp < 0 and d >= 0) ...
  double tc;                 // ... at this time
};

// KNOWN: the caller has the overlap predicates of this column already (column_flags): `near` = the discs within reach of
// `id`; the distance tests are not repeated, and `memo` carries the last pair's result over to the partner's turn.
template <int N, bool SPLITV, bool KNOWN = false>
__device__ __forceinline__ void column_candidate(const StepArgs &p, const GridView<N, SPLITV> &G, int g, int id, double t0,
                                                 double dt_col, double &best_t, int &best_j, int &best_kind,
                                                 unsigned known_near = 0u, PairMemo *memo = nullptr) {
  constexpr int E = 2 * N + 1, PK = 2 * N;
  const double rad = id == PK ? kRPuck : kRPlayer;
  const double kpx = G.pos(g, id, 0), kpy = G.pos(g, id, 1);
  unsigned near = known_near;
  if (!KNOWN) {
#pragma unroll
    for (int o = 0; o < E; ++o) {                            // cpp:466-471, every other disc
      const double ddx = kpx - G.pos(g, o, 0), ddy = kpy - G.pos(g, o, 1);
      const double le = (o == PK || id == PK) ? p.le_pk : p.le_pp;
      near |= (ddx * ddx + ddy * ddy <= le) ? (1u << o) : 0u;
    }
  }
  near &= ~(1u << id);
  const bool up = kpy + rad >= kY / 2.0, down = kpy - rad <= -kY / 2.0;      // cpp:374, 397
  const bool right = kpx + rad >= kX / 2.0, left = kpx - rad <= -kX / 2.0;   // cpp:420, 443
  best_t = 1.0e9; best_j = -1; best_kind = -1;               // Collision(), Collision.h:9
  if (!(near != 0u || up || down || right || left)) return;

  const int gm = g - 1;
  const double mpx = G.pos(gm, id, 0), mpy = G.pos(gm, id, 1), mvx = G.vel(gm, id, 0), mvy = G.vel(gm, id, 1);
  const double t_m = t0 + dt_col * (double)(g - 1);          // T row, cpp:201 (g-1 = 0 gives t_)
  if (up || down || right || left) {
    if (up && mvy > 0) {                                     // cpp:385-392
      const double tc = t_m + max0(div_ieee(kY / 2.0 - rad - mpy, mvy));
      if (tc < best_t) { best_t = tc; best_kind = UP; }
    }
    if (down && mvy < 0) {                                   // cpp:408-415
      const double tc = t_m + max0(div_ieee(-kY / 2.0 + rad - mpy, mvy));
      if (tc < best_t) { best_t = tc; best_kind = DOWN; }
    }
    if (right && mvx > 0) {                                  // cpp:431-438
      const double tc = t_m + max0(div_ieee(kX / 2.0 - rad - mpx, mvx));
      if (tc < best_t) { best_t = tc; best_kind = RIGHT; }
    }
    if (left && mvx < 0) {                                   // cpp:454-461
      const double tc = t_m + max0(div_ieee(-kX / 2.0 + rad - mpx, mvx));
      if (tc < best_t) { best_t = tc; best_kind = LEFT; }
    }
  }
  while (near != 0u) {                                       // cpp:482-497, ascending o = entity order
    const int o = __builtin_ctz(near);
    near &= near - 1u;
    bool have = false;
    double tc = 0.0;
    if (KNOWN && memo->a == o && memo->b == id) {            // evaluated from the partner's side a moment ago: same value
      have = memo->valid;
      tc = memo->tc;
    } else {
      const double sigma = rad + (o == PK ? kRPuck : kRPlayer);
      const double dpx = G.pos(gm, o, 0) - mpx, dpy = G.pos(gm, o, 1) - mpy;     // cpp:484
      const double dvx = G.vel(gm, o, 0) - mvx, dvy = G.vel(gm, o, 1) - mvy;     // cpp:485
      const double dpdv = dpx * dvx + dpy * dvy;
      const double dvdv = dvx * dvx + dvy * dvy;
      const double dpdp = dpx * dpx + dpy * dpy;
      const double d = dpdv * dpdv - dvdv * (dpdp - sigma * sigma);              // cpp:487
      const double dvdp = dvx * dpx + dvy * dpy;
      have = (dvdp < 0) & (d >= 0);                          // discs in resting contact fail this at every column
      if (have) tc = t_m + max0(div_ieee(-(dvdp + sqrt_ieee(d)), dvdv));         // cpp:490
      if (KNOWN) { memo->a = id; memo->b = o; memo->valid = have; memo->tc = tc; }
    }
    if (have) {
      const bool ok = tc < best_t;
      best_t = ok ? tc : best_t;
      best_j = ok ? o : best_j;
      best_kind = ok ? (int)PAIR : best_kind;
    }
  }
}

// The overlap predicates of grid column g for ALL discs at once (cpp:374,397,420,443,471): bit e of the
// result = disc e is inside a wall zone or within reach of another disc at this column, i.e. exactly
// the (disc, column) pairs for which column_candidate can produce anything.  `consider`: discs that
// can have a flag at all (the lane pass' conservative watch; all ones when unknown) -- a pair can only
// overlap if the watch fired for both of its discs.
// near_of (small teams, may be null): per disc, the discs within reach at this column -- what column_candidate<KNOWN> takes.
template <int N, bool SPLITV>
__device__ __forceinline__ unsigned column_flags(const StepArgs &p, const GridView<N, SPLITV> &G, int g, unsigned consider,
                                                 unsigned *near_of = nullptr) {
  constexpr int E = 2 * N + 1, PK = 2 * N;
  unsigned fl = 0u;
  if constexpr (N <= 4) {                                    // small teams: straight-line code over one batch of LDS loads
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (N >= 4 && !((consider >> e) & 1u)) continue;       // skip the loads of unflagged discs
      const double rad = disc_radius<N>(e);
      const double px = G.pos(g, e, 0), py = G.pos(g, e, 1);
      const bool wall = (py + rad >= kY / 2.0) | (py - rad <= -kY / 2.0) | (px + rad >= kX / 2.0) | (px - rad <= -kX / 2.0);
      fl |= wall ? (1u << e) : 0u;
#pragma unroll
      for (int o = e + 1; o < E; ++o) {
        const double ddx = px - G.pos(g, o, 0), ddy = py - G.pos(g, o, 1);
        const double le = (e == PK || o == PK) ? p.le_pk : p.le_pp;
        const bool nr = ddx * ddx + ddy * ddy <= le;
        fl |= nr ? ((1u << e) | (1u << o)) : 0u;
        if (near_of != nullptr) { near_of[e] |= nr ? (1u << o) : 0u; near_of[o] |= nr ? (1u << e) : 0u; }
      }
    }
  } else {                                                   // large teams (up to 105 pairs): loops over the flagged discs only,
    unsigned todo = consider;                                // rolled -- unrolled, the pair tests of 7v7 alone are ~900 instructions
    while (todo != 0u) {                                     // and pin more registers than the rest of the wave pass needs
      const int e = __builtin_ctz(todo);
      todo &= todo - 1u;
      const double rad = e == PK ? kRPuck : kRPlayer;
      const double px = G.pos(g, e, 0), py = G.pos(g, e, 1);
      const bool wall = (py + rad >= kY / 2.0) | (py - rad <= -kY / 2.0) | (px + rad >= kX / 2.0) | (px - rad <= -kX / 2.0);
      fl |= wall ? (1u << e) : 0u;
      unsigned others = todo;                                // flagged discs after e
      while (others != 0u) {
        const int o = __builtin_ctz(others);
        others &= others - 1u;
        const double ddx = px - G.pos(g, o, 0), ddy = py - G.pos(g, o, 1);
        const double le = (e == PK || o == PK) ? p.le_pk : p.le_pp;
        fl |= (ddx * ddx + ddy * ddy <= le) ? ((1u << e) | (1u << o)) : 0u;
      }
    }
  }
  return fl & consider;
}

// Live events, one slot per disc (slot index = the event's i_).  meta < 0: empty;
// else (seq << 8) | (kind << 5) | (j + 1).  Select-based (no branches): the list is replicated across
// the lanes of a group and sits on the serial path of every event.
template <int E>
struct EventSlots {
  double t[E];
  int meta[E];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int e = 0; e < E; ++e) { t[e] = 1.0e9; meta[e] = -1; }
  }
  __device__ __forceinline__ void put(int id, double tt, int j, int kind, int seq, bool really = true) {
    const int m = (seq << 8) | (kind << 5) | (j + 1);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool w = really & (e == id);
      t[e] = w ? tt : t[e];
      meta[e] = w ? m : meta[e];
    }
  }
  __device__ __forceinline__ bool any() const {
    int acc = -1;
#pragma unroll
    for (int e = 0; e < E; ++e) acc = acc > meta[e] ? acc : meta[e];
    return acc >= 0;
  }
  // Earliest event; among equal times the most recently inserted (largest seq).  Removes it.
  __device__ __forceinline__ void pop(int &ci, double &ct, int &cj, int &ckind) {
    ci = -1; ct = 0.0;
    int cm = -1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool better = (meta[e] >= 0) & ((cm < 0) | (t[e] < ct) | ((t[e] == ct) & (meta[e] > cm)));
      ci = better ? e : ci;
      ct = better ? t[e] : ct;
      cm = better ? meta[e] : cm;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) meta[e] = (e == ci) ? -1 : meta[e];
    cj = (cm & 31) - 1;
    ckind = (cm >> 5) & 7;
  }
  __device__ __forceinline__ void get(int id, double &tt, int &m) const {
    tt = 1.0e9; m = -1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      tt = (e == id) ? t[e] : tt;
      m = (e == id) ? meta[e] : m;
    }
  }
  // cpp:347-352: drop everything that touches ci or cj (cj = -1 for walls never matches).
  __device__ __forceinline__ void purge(int ci, int cj) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = (meta[e] & 31) - 1;
      const bool hit = (e == ci) | (e == cj) | ((j != -1) & ((j == ci) | (j == cj)));
      meta[e] = hit ? -1 : meta[e];
    }
  }
};

// This lane's part in carrying out one event (lanes 0..3 of the group): which (disc, axis) chain it
// integrates and the correction rule of the re-integrated columns.  Lanes 0,1 carry disc i (x, y);
// lanes 2,3 carry disc j of a pair event and a harmless duplicate of disc i for a wall event.
template <int N>
struct EventLane {
  int d, ax;
  bool first, is_pair, my_axis, positive_wall, store;
  double u, coef, tau, inv, r_own, rsum, lt, wall;
  double push101, push11;    // ((1.01*r_own)/(ri+rj)), ((1.1*r_own)/(ri+rj)): cpp:263-264, 588-589 (host-computed quotients)
  double mass_ratio;         // (2*m_other)/(mi+mj), cpp:273-274
};

template <int N, bool SPLITV>
__device__ __forceinline__ EventLane<N> event_lane(const StepArgs &p, const GridView<N, SPLITV> &G, int ci, int cj, int kind,
                                                   int lg) {
  constexpr int PK = 2 * N;
  EventLane<N> c;
  c.is_pair = kind == PAIR;
  c.first = (lg & 2) == 0;
  c.ax = lg & 1;
  c.d = (c.is_pair && !c.first) ? cj : ci;
  c.store = c.is_pair || c.first;
  c.r_own = disc_radius<N>(c.d);
  c.u = G.ctl(c.d, c.ax);
  c.coef = c.d == PK ? 0.2 : 1.0;
  c.tau = c.d == PK ? p.tau_puck : p.tau_player;
  c.inv = c.d == PK ? p.inv_tau_puck : p.inv_tau_player;
  const bool with_puck = ci == PK || cj == PK;
  c.rsum = disc_radius<N>(ci) + disc_radius<N>(c.is_pair ? cj : ci);
  c.lt = with_puck ? p.lt_pk : p.lt_pp;                      // s <= lt  <=>  rsum - sqrt(s) > 0
  c.push101 = !with_puck ? p.push101_pp : (c.d == PK ? p.push101_pkk : p.push101_plk);
  c.push11 = !with_puck ? p.push11_pp : (c.d == PK ? p.push11_pkk : p.push11_plk);
  c.mass_ratio = !with_puck ? p.ratio_pp : (c.d == PK ? p.ratio_vs_player : p.ratio_vs_puck);
  c.my_axis = (kind == UP || kind == DOWN) ? (c.ax == 1) : (c.ax == 0);
  c.positive_wall = (kind == UP || kind == RIGHT);
  c.wall = (kind == UP || kind == DOWN) ? kY / 2.0 : kX / 2.0;
  return c;
}

// correctOverlap of one re-integrated column (pair: cpp:581-592, wall: cpp:594-623), then the store
// into the grid.  Lanes 0..3 of the group, all active together (DPP exchanges).  The pair rule's square
// root sits behind a branch that is uniform over those four lanes (GSIM_FIX_BRANCH).
template <int N, bool SPLITV>
__device__ __forceinline__ void fix_store_column(const EventLane<N> &c, const GridView<N, SPLITV> &G, Chain &x, int g,
                                                 bool wall_fix) {
  if (c.is_pair) {
    const double po = quad_xor2(x.p);
    const double pij = c.first ? po - x.p : x.p - po;
    const double sq = pij * pij;
    const double s = sq + quad_xor1(sq);                             // the same value on the event's four lanes
#if GSIM_FIX_BRANCH
    if (s <= c.lt) {                                                 // overlap > 0 (cpp:584-585): rare after the impact column,
      const double overlap = c.rsum - sqrt_ieee(s);                  // and the square root is a third of this column's latency
      const double cc = (c.first ? -c.push11 : c.push11) * overlap;  // -(a*b) == (-a)*b exactly
      x.p = x.p + cc * pij;                                          // x - c*pij == x + (-(c*pij))
    }
#else
    const double overlap = c.rsum - sqrt_ieee(s);
    const double cc = (c.first ? -c.push11 : c.push11) * overlap;    // -(a*b) == (-a)*b exactly
    const double moved = x.p + cc * pij;                             // x - c*pij == x + (-(c*pij))
    x.p = (s <= c.lt) ? moved : x.p;
#endif
  } else if (c.my_axis && wall_fix) {
    const double overlap = c.positive_wall ? (x.p + c.r_own - c.wall) : (-x.p + c.r_own - c.wall);
    const double cc = 1.1 * overlap;
    const double moved = c.positive_wall ? x.p - cc : x.p + cc;
    x.p = (overlap > 0) ? moved : x.p;
  }
  if (c.store) {
    G.pos(g, c.d, c.ax) = x.p;
    G.vel(g, c.d, c.ax) = x.v;
  }
}

// Completes columns from..to of a pending event's discs (see arena_step_group): continues from grid
// column from-1 under that event's correction rule, exactly the reference's loop cpp:292-297 / 336-340.
template <int N, int FAST, bool SPLITV>
__device__ __forceinline__ void finish_columns(const StepArgs &p, const GridView<N, SPLITV> &G, int ci, int cj, int kind,
                                               int from, int to, double dt_col, int lg) {
  if (lg < 4) {
    const EventLane<N> c = event_lane<N, SPLITV>(p, G, ci, cj, kind, lg);
    Chain x;
    x.p = G.pos(from - 1, c.d, c.ax);
    x.v = G.vel(from - 1, c.d, c.ax);
    const bool fast = FAST != 0 && fast_division_safe(c.u, x.v);
    for (int g = from; g <= to; ++g) {
      x = rk4_dyn<FAST>(x, c.u, c.coef, c.tau, c.inv, dt_col, fast);
      fix_store_column<N, SPLITV>(c, G, x, g, true);
    }
  }
  grid_sync<SPLITV>();
}

// insertCollisions' inner loops (cpp:365-506) for the discs of a carried-out event over grid columns
// lo..hi: the first column whose best candidate satisfies t_col - t_ < dt_, per disc.  A pair's two
// scans run side by side (lanes [0, GW/2) <-> columns of disc i, lanes [GW/2, GW) <-> disc j): they
// only read the grid, so their order does not matter.  done_*: nothing (more) to look for; found_* and
// the event fields are set when a hit is found.  Group-uniform results.
template <int N, int GW, bool SPLITV>
__device__ __forceinline__ void scan_columns(const StepArgs &p, const GridView<N, SPLITV> &G, int ci, int cj, int lo, int hi,
                                             double t0, double dt, double dt_col, int lg, int gbase, bool &done_i,
                                             bool &done_j, bool &found_i, bool &found_j, double &ti, int &ji, int &ki,
                                             double &tj, int &jj, int &kj) {
  const bool two = cj >= 0;
  const int H = two ? GW / 2 : GW;
  const int which = (two && lg >= H) ? 1 : 0;
  const int l = which ? lg - H : lg;
  const int id = which ? cj : ci;
  for (int g0 = lo; g0 <= hi && !(done_i && done_j); g0 += H) {
    const int g = g0 + l;
    const bool want = which ? !done_j : !done_i;
    bool hit = false;
    double bt = 1.0e9;
    int bj = -1, bk = -1;
    if (g <= hi && want) {
      column_candidate<N, SPLITV>(p, G, g, id, t0, dt_col, bt, bj, bk);
      hit = (bt - t0 < dt);                                  // cpp:501
    }
    const unsigned long long m = group_ballot<GW>(hit, gbase);
    const unsigned long long mi = two ? (m & ((1ull << (GW / 2)) - 1ull)) : m;
    const unsigned long long mj = two ? (m >> (GW / 2)) : 0ull;
    if (!done_i && mi != 0ull) {
      const int src = gbase + __builtin_ctzll(mi);
      ti = __shfl(bt, src); ji = __shfl(bj, src); ki = __shfl(bk, src);
      done_i = true; found_i = true;
    }
    if (!done_j && mj != 0ull) {
      const int src = gbase + GW / 2 + __builtin_ctzll(mj);
      tj = __shfl(bt, src); jj = __shfl(bj, src); kj = __shfl(bk, src);
      done_j = true; found_j = true;
    }
  }
}

// Full reference step for arena `a`.  G: this group's LDS region (WaveLayout<N>::DOUBLES).
// lg: lane index inside the group; gbase: wave lane of the group's lane 0.
// discmask: bits 0..E-1 = discs whose overlap watch fired in the lane pass (all ones when unknown) --
// a disc outside the mask provably has no candidate in the first scan, so that scan is skipped;
// bits 16..21 = first grid column that can hold a candidate (1 when unknown), where the first scans
// start.
// Returns without touching anything if the game is already decided (cpp:121-123).  Return value (group-uniform):
// bit 0 = the step had a collision event, bit 1 = the first scans saw an overlap predicate of a considered disc at
// some column (a disc in a wall zone, a pair within reach): the lane pass' watch, a superset of those predicates,
// would have flagged this step.
//
// NOISE (w_stdev_ != 0): every RK4 of a player consumes 8 draws of the arena's tape in the reference's
// call order -- populateStateGrid: column by column, A1..Bn (cpp:199-215); carryOutFirstCollision:
// disc i then disc j to the impact instant, i then j for the remainder, then i, j per re-integrated
// column (cpp:255-256, 287-288, 294-295 / 309, 331, 338); the puck draws nothing.  `cur` is that
// cursor; reads past the tape are clamped and reported through consumed[] so the host can redo the
// step with a longer tape.
//
// NOISE: 0 none; 1 host tape as above; 2 the device generator (gsim_rng.hpp, one wavefront per arena): the draws are
// produced into an LDS window just ahead of their use -- ten grid columns at a time during the fill, one event's worth
// before each carry-out -- in the same order, and dropped from the window as they are consumed.
template <int N, int GW, int FAST, int NOISE = 0, bool SPLITV = false, bool COH = false>
__device__ unsigned arena_step_group(const StepArgs &p, const double *__restrict__ ctl, long long a,
                                 const GridView<N, SPLITV> G, int lg, int gbase, unsigned discmask, DevRng *rng = nullptr) {
  using L = WaveLayout<N>;
  constexpr int E = L::E, PK = L::PK;
  static_assert(2 * E <= GW, "group too narrow for the (disc, axis) chains");
  const long long B = p.B;
  double TAS = gld_at<COH>(p.state, 0 * B + a, p.n_state), TBS = gld_at<COH>(p.state, 1 * B + a, p.n_state);
  if (!(TAS < p.win && TBS < p.win)) return 0u;              // undecided(), cpp:57-60,121
  const double t0 = gld_at<COH>(p.tcur, a, p.n_tcur);
  const double dt = p.dt, dt_col = p.dt_col;
  static_assert(NOISE != 2 || (GW == 64 && !SPLITV), "device generator: one wavefront per arena");
  const double *tape = NOISE == 1 ? p.tape + p.tape_off[a] : (NOISE == 2 ? rng->buf : nullptr);
  const int tape_last = NOISE == 1 ? p.tape_n[a] - 8 : (NOISE == 2 ? 0x3FFFFFFF : 0);   // last legal start of an 8-draw record
  int cur = 0;                                              // draws consumed so far (group-uniform)
  const bool stamp = GSIM_STAMPS && p.stamps && gbase == 0;               // diagnostic: group 0 times its phases
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
  if (stamp) ts0 = __builtin_amdgcn_s_memtime();

  // ---- fill: populateStateGrid, cpp:173-217 ----
  if (NOISE == 2) {
    // the same chains, ten columns at a time: all 64 lanes generate the block's 8 * 10 * P draws, the chain lanes use them
    constexpr int FB = 10;
    const int e = lg >> 1, ax = lg & 1;
    const bool chain = lg < 2 * E;
    Chain x{0.0, 0.0};
    double u = 0.0, coef = 0.2, tau = p.tau_puck, inv = p.inv_tau_puck;   // cpp:188: puck input is (0, 0)
    if (chain) {
      const int row = state_row<N>(e);
      x.p = gld_at<COH>(p.state, (long long)(row + ax) * B + a, p.n_state);
      x.v = gld_at<COH>(p.state, (long long)(row + 2 + ax) * B + a, p.n_state);
      if (e != PK) { u = ctl[(long long)(2 * e + ax) * B + a]; coef = 1.0; tau = p.tau_player; inv = p.inv_tau_player; }
      G.ctl(e, ax) = u;
      G.pos(0, e, ax) = x.p;
      G.vel(0, e, ax) = x.v;
    }
    const double sg = (chain && e != PK) ? p.sigma : 0.0;
    for (int i0 = 1; i0 <= kGrid; i0 += FB) {
      rng->ensure(8 * FB * L::P, gbase + lg);
      if (chain) {
        // the division flavour is chosen once per block, outside the column loop (inside it costs 9 % of the kernel)
        if (FAST != 0 && fast_division_safe(u, x.v)) {
          for (int i = i0; i < i0 + FB; ++i) {
            x = rk4_axis_noise<FAST>(x, u, coef, tau, dt_col, sg, tape + 8 * ((i - i0) * L::P + (e != PK ? e : 0)) + ax, inv);
            G.pos(i, e, ax) = x.p;
            G.vel(i, e, ax) = x.v;
          }
        } else {
          for (int i = i0; i < i0 + FB; ++i) {
            x = rk4_axis_noise<0>(x, u, coef, tau, dt_col, sg, tape + 8 * ((i - i0) * L::P + (e != PK ? e : 0)) + ax);
            G.pos(i, e, ax) = x.p;
            G.vel(i, e, ax) = x.v;
          }
        }
      }
      rng->consume(8 * FB * L::P, gbase + lg);
    }
  } else
  if (lg < 2 * E) {
    const int e = lg >> 1, ax = lg & 1;
    const int row = state_row<N>(e);
    Chain x;
    x.p = gld_at<COH>(p.state, (long long)(row + ax) * B + a, p.n_state);
    x.v = gld_at<COH>(p.state, (long long)(row + 2 + ax) * B + a, p.n_state);
    double u = 0.0, coef = 0.2, tau = p.tau_puck, inv = p.inv_tau_puck;   // cpp:188: puck input is (0, 0)
    if (e != PK) { u = ctl[(long long)(2 * e + ax) * B + a]; coef = 1.0; tau = p.tau_player; inv = p.inv_tau_player; }
    G.ctl(e, ax) = u;
    G.pos(0, e, ax) = x.p;
    G.vel(0, e, ax) = x.v;
    const bool fast = FAST != 0 && fast_division_safe(u, x.v);
    const double sg = e != PK ? p.sigma : 0.0;
    for (int i = 1; i <= kGrid; ++i) {
      if (NOISE) {
        const int c0 = 8 * ((i - 1) * L::P + (e != PK ? e : 0));
        x = rk4_axis_noise(x, u, coef, tau, dt_col, sg, tape + (c0 < tape_last ? c0 : tape_last) + ax);
      } else
      x = rk4_dyn<FAST>(x, u, coef, tau, inv, dt_col, fast);
      G.pos(i, e, ax) = x.p;
      G.vel(i, e, ax) = x.v;
    }
  }
  if (NOISE == 1) cur = 8 * kGrid * L::P;
  grid_sync<SPLITV>();
  if (stamp) ts1 = __builtin_amdgcn_s_memtime();

  // ---- event loop, cpp:136-150 ----
  EventSlots<E> ev;
  ev.clear();
  int seq = 0, base = 1, counter = 0;
  bool scored = false;
  unsigned n_resolved = 0, n_skipped = 0, n_overflow = 0;
  bool saw_predicate = false;
  // First pass, checks = all discs (cpp:145 -> insertCollisions): lane <-> grid column.  Each chunk of
  // GW columns first gets the overlap predicates of every disc in one straight-line pass
  // (column_flags); only (disc, column) pairs with a predicate set can yield a candidate, so only those
  // are evaluated, disc by disc.  The scans of different discs do not influence each other (nothing
  // writes the grid), so walking the columns chunk-wise for all discs together finds, for every disc,
  // the same first admissible column as the reference's disc-by-disc order; the list order the
  // reference gets from that order (ascending disc id) is restored by numbering these events with
  // their disc id (seq = id; later insertions count on from E).
  {
    const unsigned consider = (discmask & 0xFFFFu) == 0xFFFFu ? ((1u << E) - 1u) : (discmask & 0xFFFFu);
    int scan_from = (int)((discmask >> 16) & 63u);           // first column that can hold a candidate (1 when unknown)
    scan_from = scan_from < 1 ? 1 : (scan_from > kGrid ? kGrid : scan_from);
    unsigned open = consider;                                // discs still looking for their first event
    for (int g0 = scan_from; g0 <= kGrid && open != 0u; g0 += GW) {
      const int g = g0 + lg;
      constexpr bool KN = GSIM_SCAN_KNOWN && N <= 3;       // the candidates reuse this pass' distance tests and each pair's time
      unsigned near_of[KN ? E : 1];
      if (KN) {
#pragma unroll
        for (int e = 0; e < E; ++e) near_of[e] = 0u;
      }
      PairMemo memo{-1, -1, false, 0.0};
      const unsigned fl = g <= kGrid ? (column_flags<N, SPLITV>(p, G, g, consider, KN ? near_of : nullptr) & open) : 0u;
      if (group_ballot<GW>(fl != 0u, gbase) == 0ull) continue;
      saw_predicate = true;
      unsigned todo = 0u;                                    // union of the lanes' flags (group-uniform)
#pragma unroll
      for (int e = 0; e < E; ++e) todo |= group_ballot<GW>(((fl >> e) & 1u) != 0u, gbase) != 0ull ? (1u << e) : 0u;
      while (todo != 0u) {
        const int id = __builtin_ctz(todo);
        todo &= todo - 1u;
        bool hit = false;
        double bt = 1.0e9;
        int bj = -1, bk = -1;
        if ((fl >> id) & 1u) {
          if constexpr (KN) {
            unsigned nr = 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) nr = (e == id) ? near_of[e] : nr;
            column_candidate<N, SPLITV, true>(p, G, g, id, t0, dt_col, bt, bj, bk, nr, &memo);
          } else {
            column_candidate<N, SPLITV>(p, G, g, id, t0, dt_col, bt, bj, bk);
          }
          hit = (bt - t0 < dt);                              // cpp:501
        }
        const unsigned long long m = group_ballot<GW>(hit, gbase);
        if (m != 0ull) {
          const int src = gbase + __builtin_ctzll(m);
          ev.put(id, __shfl(bt, src), __shfl(bj, src), __shfl(bk, src), id);
          open &= ~(1u << id);
        }
      }
    }
    seq = E;
  }
  counter = 1;
  const bool had_events = ev.any();
  if (stamp) ts2 = __builtin_amdgcn_s_memtime();

  // Lazily re-integrated columns.  The reference re-integrates the colliding discs over ALL remaining
  // columns base..50 (cpp:285-297, 330-340) and only then re-scans them from column base (cpp:147).  When
  // discs stay in contact the re-scan finds the next event in column base itself, and that event
  // re-integrates the same discs from there again: the 49 events of a step that runs into the iteration
  // cap (a puck spawned inside a player, C2: 4.5 % of the arenas in their first step) recompute ~2400
  // columns of which ~100 are ever read.  Here, once contact is being sustained (an event on the discs of
  // the event before it; an isolated event produces all its columns at once), the columns are produced in
  // growing chunks (1, 3, then half a group's width), each chunk scanned as soon as it exists -- lanes of the lower half of the
  // group take disc i's columns, the upper half disc j's; the two scans are independent of each other
  // since neither writes the grid -- and production stops once both discs have their event.  What is
  // left (columns vu+1..50 of the "pending" discs, to be continued from grid column vu under the
  // pending event's correction rule) is only computed if somebody can read it:
  //   * the next carried-out event involves exactly the pending discs: it restarts them from its own
  //     column base-1 (>= the pending event's base, event times never move back past a carried-out
  //     column), so only columns up to base-1 are completed and the rest is dropped unread;
  //   * any other carried-out event, and the end of the loop: the pending discs are completed to
  //     column 50 first, which leaves the grid exactly as the reference has it at that point.
  // Between a carried-out event and the next one nothing but the re-scans of i and j reads the grid
  // (skipped puck events, cpp:230-231, read nothing), so every value read is the value the reference
  // reads.  With noise the draws are bound to the RK4 call order and the whole remainder is produced
  // at once (one chunk).
  int pend_i = -1, pend_j = -1, pend_kind = -1, vu = kGrid;   // group-uniform
  int last_i = -1, last_j = -1, last_kind = -1;              // the event carried out last (fixed-point shortcut below)
  double last_t = 0.0, last_damage = 0.0;
  bool last_overflow = false;

  while (ev.any() && counter < kLoopCap) {
    int ci, cj, ckind;
    double ct;
    unsigned long long te0 = 0, te1 = 0, te2 = 0;
    if (stamp) te0 = __builtin_amdgcn_s_memtime();
    ev.pop(ci, ct, cj, ckind);                               // cpp:227
    counter++;
    if ((ci == PK || cj == PK) && scored) { n_skipped++; continue; }   // cpp:230-231
    const bool is_pair = ckind == PAIR;
    // the same discs as the event carried out just before: contact is being sustained, produce columns lazily
    const bool same_discs = last_i >= 0 && is_pair == (last_kind == PAIR) &&
                            ((ci == last_i && cj == last_j) || (ci == last_j && cj == last_i));
    bool repeats = !NOISE && same_discs && ct == last_t && ckind == last_kind;
    if (repeats && is_pair) {                                // the partner's own event, if it has one (slot cj)
      double tb; int mb;
      ev.get(cj, tb, mb);
      repeats = mb < 0 || tb > ct || ((mb & 31) - 1 == ci && ((mb >> 5) & 7) == PAIR);
    }
    if (repeats) {
      // Fixed point.  This event has the time, discs and kind of the one carried out just before it, so
      // it starts from the same column base-1 (that carry-out only wrote columns >= base) and rebuilds,
      // value for value, the columns that are already there; its re-scans then find what the last
      // re-scans found on that same grid, i.e. they re-insert this event (slot i) and the partner's
      // event b (slot j) with unchanged values, this one first.  The next pop therefore returns this
      // event again -- directly if b is later, and with i and j swapped if b is its equal-time mirror
      // image (b then wins the tie as the newer entry; cpp:259-274 are symmetric in i and j, so the
      // orientation changes no value) -- and so on: the reference repeats the identical iteration until
      // the cap ends the step (cpp:150).  That is what a puck spawned inside a player or a disc
      // pressed into a wall does.  (An equal-time b that is something else -- a wall event, another
      // partner -- would be popped next instead; then this is not a fixed point and the event is
      // carried out normally.)  Nothing but the counters and, for two players, the damage sums changes
      // during those iterations, so they are booked here without being recomputed.
      const int reps = kLoopCap - counter + 1;               // this iteration and all that follow, up to the cap
      n_resolved += (unsigned)reps;
      n_overflow += last_overflow ? (unsigned)reps : 0u;
      if (is_pair && ci != PK && cj != PK && lg == 0) {       // cpp:277-282, one rounding per iteration like the reference
        constexpr int Pn = L::P;
        double dij = gld_at<COH>(p.damage, (long long)(ci * Pn + cj) * B + a, p.n_damage), dji = gld_at<COH>(p.damage, (long long)(cj * Pn + ci) * B + a, p.n_damage);
        for (int r = 0; r < reps; ++r) { dij += last_damage; dji += last_damage; }
        gst_at<COH>(p.damage, (long long)(ci * Pn + cj) * B + a, p.n_damage, dij);
        gst_at<COH>(p.damage, (long long)(cj * Pn + ci) * B + a, p.n_damage, dji);
      }
      counter = kLoopCap;
      ev.put(ci, ct, cj, ckind, seq++);                      // the list is not empty when the cap ends the loop
      break;
    }
    if (!NOISE && pend_i >= 0) {
      const bool same = is_pair ? (pend_kind == PAIR && ((ci == pend_i && cj == pend_j) || (ci == pend_j && cj == pend_i)))
                                : (pend_kind != PAIR && ci == pend_i);
      if (!same) {
        finish_columns<N, FAST, SPLITV>(p, G, pend_i, pend_j, pend_kind, vu + 1, kGrid, dt_col, lg);
        pend_i = -1; vu = kGrid;
      }
    }
    base = (int)__builtin_floor(div_ieee(ct - t0, dt_col)) + 1;        // cpp:237
    last_i = ci; last_j = cj; last_kind = ckind; last_t = ct; last_overflow = base > kGrid; last_damage = 0.0;
    if (base > kGrid) { base = kGrid; n_overflow++; }        // reference: out-of-bounds column
    if (base < 1) base = 1;
    n_resolved++;
    if (!NOISE && pend_i >= 0 && base - 1 > vu) {            // the new event starts from a column not produced yet
      finish_columns<N, FAST, SPLITV>(p, G, pend_i, pend_j, pend_kind, vu + 1, base - 1, dt_col, lg);
      vu = base - 1;
    }
    const double dt_t = ct - (t0 + dt_col * (double)(base - 1));       // cpp:254,308

    // Lanes 0,1 carry disc i (x, y); lanes 2,3 carry disc j for a pair event and a harmless
    // duplicate of disc i for a wall event, so both kinds share one instruction stream.
    if (stamp) te1 = __builtin_amdgcn_s_memtime();
    const EventLane<N> el = event_lane<N, SPLITV>(p, G, ci, cj, ckind, lg);
    Chain x{0.0, 0.0};
    bool fast = false, goal = false;
    // noise bookkeeping: np players among the event's discs draw; this lane's disc is the slot-th of them
    const int np = NOISE ? ((ci != PK ? 1 : 0) + ((is_pair && cj != PK) ? 1 : 0)) : 0;
    const int slot = (is_pair && !el.first && ci != PK) ? 1 : 0;
    const double sg = (NOISE && el.d != PK) ? p.sigma : 0.0;
    int call = 0;                                            // this disc's RK4 calls so far in this event
    auto noisy = [&](Chain xx, double hh, bool usefast) {       // usefast: this chain's magnitudes are in div_tau's window (device generator only)
      const int c0 = cur + 8 * (call * np + slot);
      ++call;
      return rk4_noise_dyn<NOISE == 2 ? FAST : 0>(xx, el.u, el.coef, el.tau, hh, sg, tape + (c0 < tape_last ? c0 : tape_last) + el.ax, el.inv,
                                                  usefast);
    };
    if (NOISE == 2) rng->ensure(8 * np * 2, gbase + lg);     // the draws of the two RK4 calls around the impact (cur == 0: the window starts at them)
    if (lg < 4) {
      x.p = G.pos(base - 1, el.d, el.ax);
      x.v = G.vel(base - 1, el.d, el.ax);
      fast = FAST != 0 && fast_division_safe(el.u, x.v);
      if (NOISE) x = noisy(x, dt_t, fast); else
      x = rk4_dyn<FAST>(x, el.u, el.coef, el.tau, el.inv, dt_t, fast);   // cpp:255-256, 309

      if (is_pair) {                                         // cpp:242-282
        {                                                    // cpp:259-265 (pij not normalised)
          const double po = quad_xor2(x.p);
          const double pij = el.first ? po - x.p : x.p - po;
          const double sq = pij * pij;
          const double s = sq + quad_xor1(sq);
          const double overlap = el.rsum - sqrt_ieee(s);
          const double c = (el.first ? -el.push101 : el.push101) * overlap;   // ((1.01*r)/(ri+rj))*overlap, sign of the push
          const double moved = x.p + c * pij;                // x - c*pij == x + (-c)*pij exactly
          x.p = (s <= el.lt) ? moved : x.p;                  // only if overlap > 0
        }
        {                                                    // elastic impulse, cpp:268-274
          const double po = quad_xor2(x.p), vo = quad_xor2(x.v);
          const double dp = x.p - po, dv = x.v - vo;         // own - other on this axis
          const double pr = dv * dp, sq = dp * dp;
          const double dot1 = pr + quad_xor1(pr);
          const double dot2 = sq + quad_xor1(sq);
          const double s = div_ieee(el.mass_ratio * dot1, dot2);   // ((2*m_other)/(mi+mj))*dot1/dot2
          if (ci != PK && cj != PK) {                        // damage, cpp:277-282
            const double dvsq = dv * dv;
            const double dv_sq = 0.01 * (dvsq + quad_xor1(dvsq));
            last_damage = p.m_player * dv_sq;                // lane 0 of the group keeps it for the fixed-point shortcut
            if (lg == 0) {
              constexpr int Pn = L::P;
              gst_at<COH>(p.damage, (long long)(ci * Pn + cj) * B + a, p.n_damage, gld_at<COH>(p.damage, (long long)(ci * Pn + cj) * B + a, p.n_damage) + last_damage);
              gst_at<COH>(p.damage, (long long)(cj * Pn + ci) * B + a, p.n_damage, gld_at<COH>(p.damage, (long long)(cj * Pn + ci) * B + a, p.n_damage) + last_damage);
            }
          }
          x.v = x.v - s * dp;
        }
      } else {                                               // wall, cpp:300-327
        const bool flip = el.my_axis & (el.positive_wall ? (x.v > 0) : (x.v < 0));   // cpp:314-317
        x.v = flip ? -x.v : x.v;                             // v * -1.0
        const double py = quad_lane1(x.p);
        goal = el.d == PK && (ckind == LEFT || ckind == RIGHT) && py < kGoalH / 2.0 && py > -kGoalH / 2.0;  // cpp:321-322
      }
      fast = FAST != 0 && fast_division_safe(el.u, x.v);     // the velocity just changed
      if (goal) {
        x.p = 0.0; x.v = 0.0;                                // cpp:325
      } else if (NOISE) {
        x = noisy(x, dt_col - dt_t, fast);
      } else {
        x = rk4_dyn<FAST>(x, el.u, el.coef, el.tau, el.inv, dt_col - dt_t, fast);   // cpp:287-288, 331
      }
      fix_store_column<N, SPLITV>(el, G, x, base, !goal);    // cpp:289, 332 (no wall push after a goal, cpp:324-327)
      if (!NOISE && !same_discs) {                           // an isolated event: all remaining columns at once (one sync, full-width scans)
        for (int g = base + 1; g <= kGrid; ++g) {
          x = rk4_dyn<FAST>(x, el.u, el.coef, el.tau, el.inv, dt_col, fast);   // cpp:294-295, 338
          fix_store_column<N, SPLITV>(el, G, x, g, true);
        }
      }
    }
    goal = __shfl(goal ? 1 : 0, gbase) != 0;
    if (goal) {                                              // cpp:324-326
      if (!scored) { if (ckind == RIGHT) TAS += 1.0; else TBS += 1.0; }
      scored = true;
    }
    grid_sync<SPLITV>();

    if (stamp) te2 = __builtin_amdgcn_s_memtime();
    ev.purge(ci, cj);                                        // cpp:347-352
    // re-scan i_ then j_ (cpp:355-357 -> 147) over the columns as they are produced
    bool done_i = false, done_j = cj < 0, found_i = false, found_j = false;
    double ti = 0.0, tj = 0.0;
    int ji = -1, ki = -1, jj = -1, kj = -1;
    constexpr int H2 = GW / 2;
    int lo = base, hi = (NOISE || !same_discs) ? kGrid : base;
    if (NOISE == 2) {                                        // device generator: ten columns' draws at a time, like the fill
      rng->consume(8 * np * (goal ? 1 : 2), gbase + lg);     // (after a goal the remainder call did not happen, cpp:324-327)
      for (int g0 = base + 1; g0 <= kGrid; g0 += 10) {
        const int g1 = g0 + 9 < kGrid ? g0 + 9 : kGrid;
        rng->ensure(8 * np * (g1 - g0 + 1), gbase + lg);
        call = 0;
        if (lg < 4) {
          if (FAST != 0 && fast_division_safe(el.u, x.v)) {    // holds for the block's ten columns; chosen outside the loop
            for (int g = g0; g <= g1; ++g) {
              x = noisy(x, dt_col, true);                      // cpp:294-295, 338
              fix_store_column<N, SPLITV>(el, G, x, g, true);
            }
          } else {
            for (int g = g0; g <= g1; ++g) {
              x = noisy(x, dt_col, false);
              fix_store_column<N, SPLITV>(el, G, x, g, true);
            }
          }
        }
        rng->consume(8 * np * (g1 - g0 + 1), gbase + lg);
      }
      grid_sync<SPLITV>();
    } else if (NOISE && base < kGrid) {                      // host tape: the whole remainder in the reference's call order
      if (lg < 4) {
        for (int g = base + 1; g <= kGrid; ++g) {
          x = noisy(x, dt_col, false);                       // cpp:294-295, 338
          fix_store_column<N, SPLITV>(el, G, x, g, true);
        }
      }
      grid_sync<SPLITV>();
    }
    if (NOISE == 1) {                                        // calls per drawing disc: impact, remainder (not after a goal), columns
      const int np_ev = (ci != PK ? 1 : 0) + ((is_pair && cj != PK) ? 1 : 0);
      cur += 8 * np_ev * ((goal ? 1 : 2) + (kGrid - base));
    }
    for (int chunk = 0;; ++chunk) {
      scan_columns<N, GW, SPLITV>(p, G, ci, cj, lo, hi, t0, dt, dt_col, lg, gbase, done_i, done_j, found_i, found_j, ti, ji,
                                  ki, tj, jj, kj);
      if ((done_i && done_j) || hi >= kGrid) break;
      lo = hi + 1;
      const int width = chunk == 0 ? 3 : (cj < 0 ? GW : H2);
      hi = lo + width - 1 < kGrid ? lo + width - 1 : kGrid;
      if (lg < 4) {
        for (int g = lo; g <= hi; ++g) {
          x = rk4_dyn<FAST>(x, el.u, el.coef, el.tau, el.inv, dt_col, fast);   // cpp:294-295, 338
          fix_store_column<N, SPLITV>(el, G, x, g, true);
        }
      }
      grid_sync<SPLITV>();
    }
    vu = hi;
    if (vu < kGrid) { pend_i = ci; pend_j = cj; pend_kind = ckind; } else pend_i = -1;
    ev.put(ci, ti, ji, ki, seq, found_i);
    seq += found_i ? 1 : 0;
    ev.put(cj, tj, jj, kj, seq, found_j);
    seq += found_j ? 1 : 0;
    if (stamp && lg == 0) {
      atomicAdd(&p.stamp_out[STAMP_EV_HEAD], te1 - te0);
      atomicAdd(&p.stamp_out[STAMP_EV_CARRY], te2 - te1);
      atomicAdd(&p.stamp_out[STAMP_EV_SCAN], __builtin_amdgcn_s_memtime() - te2);
      atomicAdd(&p.stamp_out[STAMP_EV_COUNT], 1ull);
    }
  }
  if (!NOISE && pend_i >= 0)                                 // whatever is still pending is part of the final grid
    finish_columns<N, FAST, SPLITV>(p, G, pend_i, pend_j, pend_kind, vu + 1, kGrid, dt_col, lg);

  if (stamp) ts3 = __builtin_amdgcn_s_memtime();
  // ---- commit: column 50 -> state, t_ += dt_ (cpp:153-158) ----
  if (lg < 2 * E) {
    const int e = lg >> 1, ax = lg & 1;
    const int row = state_row<N>(e);
    gst_at<COH>(p.state, (long long)(row + ax) * B + a, p.n_state, G.pos(kGrid, e, ax));
    gst_at<COH>(p.state, (long long)(row + 2 + ax) * B + a, p.n_state, G.vel(kGrid, e, ax));
  }
  if (lg == 0) {
    if (NOISE == 1) p.consumed[a] = cur;
    gst_at<COH>(p.tcur, a, p.n_tcur, t0 + dt);
    if (scored) { gst_at<COH>(p.state, 0 * B + a, p.n_state, TAS); gst_at<COH>(p.state, 1 * B + a, p.n_state, TBS); }
    if (had_events) {
      atomicAdd(&p.counters[CNT_EVENT_STEPS], 1ull);
      if (n_resolved) atomicAdd(&p.counters[CNT_RESOLVED], (unsigned long long)n_resolved);
      if (ev.any()) atomicAdd(&p.counters[CNT_CAP_HITS], 1ull);
      if (n_overflow) atomicAdd(&p.counters[CNT_BASE_OVERFLOW], (unsigned long long)n_overflow);
      if (n_skipped) atomicAdd(&p.counters[CNT_SKIPPED], (unsigned long long)n_skipped);
    }
  }
  grid_sync<SPLITV>();   // the next step's fill must not overtake this step's column-50 reads
  if (stamp && lg == 0) {
    const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
    atomicAdd(&p.stamp_out[STAMP_FILL], ts1 - ts0);
    atomicAdd(&p.stamp_out[STAMP_SCAN0], ts2 - ts1);
    atomicAdd(&p.stamp_out[STAMP_EVENTS], ts3 - ts2);
    atomicMax(&p.stamp_out[STAMP_MAX_EVENTS], ts3 - ts2);
    atomicAdd(&p.stamp_out[STAMP_COMMIT], ts4 - ts3);
    atomicAdd(&p.stamp_out[STAMP_ARENAS], 1ull);
  }
  // One real vector instruction at the group's own EXEC before the function's regions close.  Without it the compiler
  // merges the EXEC restores of the nested tail regions (lane 0's stores, the aggregated atomics' single lane) into the
  // caller's join and then rematerialises loop-carried constants in the block in between, i.e. under the innermost,
  // single-lane EXEC (tools/isa_lint.py caught k_step_hybrid2<5,3> doing exactly that).
  if constexpr (N >= GSIM_TAIL_ANCHOR_FROM) asm volatile("v_nop");
  return (had_events ? 1u : 0u) | (saw_predicate ? 2u : 0u);
}

}  // namespace gsim
