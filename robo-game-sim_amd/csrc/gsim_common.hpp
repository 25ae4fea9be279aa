// gsim_common.hpp -- shared device-side definitions of the batched stepper.
//
// Arithmetic contract: IEEE binary64, one rounding per operation, in the
// association order of the reference (src/GameSim.cpp); the translation unit is
// compiled with -ffp-contract=off so hipcc forms no FMA from a*b+c.  Explicit
// __builtin_fma calls appear only where they are bit-identical to the reference's
// separate operations: div_tau<true> (a correctly rounded quotient) and k*2+acc
// (k*2 is exact), plus the conservative overlap watch of the lane pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic in-kernel phase stamps (s_memtime).  Compiled out by default: even as never-taken
// run-time branches they cost the one-lane hybrid kernel 2.7 % (measured).  `make STAMPS=1` builds
// the diagnostic library; rgs_get_stamps() then returns the phase totals.
#ifndef GSIM_WIDE_GROUPS
#define GSIM_WIDE_GROUPS 0
#endif
#ifndef GSIM_H2_WAVES
#define GSIM_H2_WAVES 1
#endif
#ifndef GSIM_H2_LDS_KB
#define GSIM_H2_LDS_KB 40
#endif
#ifndef GSIM_H1_WAVES
#define GSIM_H1_WAVES 1
#endif
#ifndef GSIM_CHAIN_BATCH
#define GSIM_CHAIN_BATCH 0
#endif
#ifndef GSIM_LANE_HINT
#define GSIM_LANE_HINT 0
#endif
#ifndef GSIM_H2_SPLIT_BIG
#define GSIM_H2_SPLIT_BIG 1        // two-lane hybrid, 6v6/7v7: velocity rows of the grid in global scratch
#endif
#ifndef GSIM_PAIR_BOUNDS
#define GSIM_PAIR_BOUNDS 1         // one-lane lane pass at 3v3: the pair watch's radius widened per pair by its own two discs' speed bounds
#endif
#ifndef GSIM_FLOAT_BOUND
#define GSIM_FLOAT_BOUND 1         // lane pass: the watch's per-disc speed bounds from a single-precision square root (rounded up)
#endif
#ifndef GSIM_FIX_BRANCH
#define GSIM_FIX_BRANCH 1          // correctOverlap of a re-integrated column: take the square root only when the discs overlap
#endif
#ifndef GSIM_H2_SPLIT_4
#define GSIM_H2_SPLIT_4 1          // the same at 4v4
#endif
#ifndef GSIM_SCAN_KNOWN
#define GSIM_SCAN_KNOWN 1          // wave pass, n <= 3: first-scan candidates reuse the predicate pass' distance tests and a pair's time from the partner's side
#endif
#ifndef GSIM_TAIL_ANCHOR_FROM
#define GSIM_TAIL_ANCHOR_FROM 1    // team size from which arena_step_group ends with a real vector instruction (see there)
#endif
#ifndef GSIM_H2_SPLIT_5
#define GSIM_H2_SPLIT_5 0          // diagnostic: the split grid at 5v5 too (the instantiation that once gave wrong results)
#endif
#ifndef GSIM_STAMPS
#define GSIM_STAMPS 0
#endif

namespace gsim {

constexpr int kGrid = 50;          // sub-steps per step        include/GameSim.h:6
constexpr int kLoopCap = 50;       // event-loop iteration cap  include/GameSim.h:7
constexpr int kCols = kGrid + 1;

// Arena geometry, src/GameSim.cpp:5-9.  Compile-time: the reference never changes them.
constexpr double kX = 10.0, kY = 5.0, kRPlayer = 0.2, kRPuck = 0.175, kGoalH = 1.0;

enum Kind : int { UP = 0, DOWN = 1, RIGHT = 2, LEFT = 3, PAIR = 4 };  // include/Collision.h:5

enum Counter : int {
  CNT_RESOLVED = 0,       // events carried out
  CNT_CAP_HITS = 1,       // steps ended by the 50-iteration cap
  CNT_BASE_OVERFLOW = 2,  // base index past the grid (clamped)
  CNT_SKIPPED = 3,        // puck events dropped after a goal, cpp:230-231
  CNT_EVENT_STEPS = 4,    // arena-steps whose first scan found an event
  CNT_FLAGGED = 5,        // hybrid: arena-steps the lane pass handed to the wave pass
  CNT_POOL_TIMEOUT = 6,   // pooled kernel: a bounded wait ran out (protocol error; results invalid)
  CNT_N = 8,
  // diagnostic stamps (StepArgs::stamps != 0): summed over waves, shader cycles
  STAMP_LANE = 0, STAMP_WAVE = 1, STAMP_ROUNDS = 2, STAMP_TOTAL = 3,
  STAMP_FILL = 4, STAMP_SCAN0 = 5, STAMP_EVENTS = 6, STAMP_COMMIT = 7, STAMP_ARENAS = 8,
  // maxima over waves (atomicMax): whole kernel, wave-pass cycles, rounds; and the longest single event loop
  STAMP_MAX_TOTAL = 9, STAMP_MAX_WAVE = 10, STAMP_MAX_ROUNDS = 11, STAMP_MAX_EVENTS = 12,
  // inside the event loop (group 0): pop .. carry-out start, carry-out, re-scan chunks; events counted
  STAMP_EV_HEAD = 13, STAMP_EV_CARRY = 14, STAMP_EV_SCAN = 15, STAMP_EV_COUNT = 16, STAMP_N = 17
};

// Everything one launch needs.  Arrays are SoA with the arena index fastest.
struct StepArgs {
  double dt, dt_col;                 // cpp:27-28
  double tau_player, tau_puck;       // cpp:43-44
  double inv_tau_player, inv_tau_puck;  // RN(1/tau), host-computed, for div_tau<true>
  double m_player, m_puck;           // cpp:45-46
  // Exact sqrt-free overlap tests.  For R = r_a + r_b (pp: two players, pk: player + puck):
  //   le = max{s : fl(sqrt(s)) <= R}   so  fl(sqrt(s)) <= R   <=>  s <= le   (cpp:471)
  //   lt = max{s : fl(sqrt(s)) <  R}   so  R - fl(sqrt(s)) > 0 <=>  s <= lt   (cpp:260-261, 584-585)
  double le_pp, le_pk, lt_pp, lt_pk;
  // Quotients the carry-out needs for every event, host-computed with IEEE '/' (same value as the
  // reference's run-time division).  Overlap push ((c*r_own)/(ri+rj)), c = 1.01 (cpp:263-264) and
  // c = 1.1 (OVERLAP_BUFFER, cpp:588-589): _pp own and other are players; _plk own = player, other = puck;
  // _pkk own = puck.  Impulse factor (2*m_other)/(mi+mj) (cpp:273-274): _pp; _vs_puck (own = player,
  // other = puck); _vs_player (own = puck).
  double push101_pp, push101_plk, push101_pkk, push11_pp, push11_plk, push11_pkk;
  double ratio_pp, ratio_vs_puck, ratio_vs_player;
  double win;                        // winning_score_ as a double (scores are doubles, SimState.h:25)
  long long B;                       // arenas
  double *state;                     // [S][B]
  double *tcur;                      // [B]    t_
  double *damage;                    // [P*P][B]
  const double *ctl;                 // [P][2][B] of step 0
  long long ctl_stride;              // doubles between consecutive steps' controls
  int n_steps;
  int watch_every;                   // lane pass: overlap watch every 5th column (1 = every column)
  int wall_ends;                     // lane pass: wall watch at column 50 only (monotone-coordinate argument, gsim_lane.hpp)
  // noise > 0 (wave kernel only): w_stdev_ (cpp:30) and the per-arena tape of N(0,1) draws the host
  // generated with the reference's own engine/distribution (cpp:3,41,66-67): arena a's draws are
  // tape[tape_off[a] + k], k < tape_n[a] (arenas whose step overflowed a short tape get a longer one elsewhere
  // in the buffer, the others keep theirs).
  double sigma;
  const double *tape;
  const long long *tape_off;         // [B]
  const int *tape_n;                 // [B]
  int *consumed;                     // [B] draws used by this step (> tape_n[a]: tape too short, redo)
  // trajectory snapshots inside a fused launch: after step k (1-based, counted within the launch) with
  // k % traj_every == 0 the state of every arena is copied to traj[(k/traj_every - 1)][S][B]
  double *traj;
  int traj_every;
  // the reference's log record of EVERY step of the launch (src/GameSim.cpp:161-170, include/utils.h:64-76), written by
  // the same lanes at the same point as the snapshots: rec[k][R][B], R = 3 + 2 + 2P + P*P + 1 (see snapshot_state);
  // tprev[B] = t_ of every arena when the launch began (for the "executed" row of the launch's first step).
  double *rec;
  const double *tprev;
  // device noise generator (gsim_rng.hpp): per arena the engine state, and the normals generated ahead of use that
  // the last launch left over ([B][kRngLeft], the first rng_left_n[a] valid)
  unsigned *rng_x;
  int *rng_left_n;
  double *rng_left;
  double *vgrid;                     // two-lane hybrid, n >= 4: velocity rows of the grids (GridView<N, true>)
  int stamps;                        // diagnostic: accumulate s_memtime per phase into stamp_out
  unsigned long long *stamp_out;     // [STAMP_N]
  unsigned long long *counters;      // [CNT_N]
  long long n_state, n_tcur, n_damage;   // element counts of state / tcur / damage (only the COH accessors use them)
  // Pooled kernels: which arena each lane slot of each workgroup steps (slot = 256 * workgroup + thread; null = the
  // arena with the slot's own index).  The host deals arenas that spent most of the last launches in the wave pass
  // evenly over the workgroups (rgs_hip.hip, rebalance): an arena's data never moves, only who computes it.
  unsigned long long *wave_log;          // diagnostic (STAMPS build, stamps on): 8 values per wavefront of the asynchronous pool
  const int *perm;
  unsigned *heat;                        // [B] arena-steps handed to the wave pass since the last deal (null: not counted)
};

// Global-memory accessors of the per-arena arrays (state, tcur, damage).  Plain loads and stores: an arena is
// only ever touched by wavefronts of ONE workgroup (its owner, or in the pooled kernel a sibling on the same CU
// behind workgroup-scope fences).  COH = true is the sc1 (write-through, cross-XCD coherent) form a chip-wide
// hand-over would need; it is kept for experiments and not instantiated by the shipped kernels.
typedef unsigned int gsim_v2u __attribute__((ext_vector_type(2)));
constexpr int kAuxSc1 = 16;                                  // cache-policy bits of the raw buffer builtins: sc1

template <bool COH>
__device__ __forceinline__ double gld_at(const double *base, long long idx, long long count) {
  if (COH) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(count * 8), 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(idx * 8), 0, kAuxSc1));
  }
  return base[idx];
}
template <bool COH>
__device__ __forceinline__ void gst_at(double *base, long long idx, long long count, double v) {
  if (COH) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(count * 8), 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gsim_v2u, v), r, (int)(idx * 8), 0, kAuxSc1);
  } else {
    base[idx] = v;
  }
}

// State-vector row of disc e's first component (include/SimState.h:14-23; players after the puck).
template <int N>
__device__ __forceinline__ int state_row(int e) {
  return e == 2 * N ? 2 : 6 + 4 * e;
}

// a / b, correctly rounded (hipcc lowers fp64 '/' to the div_scale/rcp/fma/div_fmas/div_fixup
// sequence, which is IEEE-exact; checked on the device by rgs_selftest_math).
__device__ __forceinline__ double div_ieee(double a, double b) { return a / b; }
__device__ __forceinline__ double sqrt_ieee(double a) { return __builtin_sqrt(a); }

// max(0, x) as the reference's fmax(0.0, x) (cpp:388,...): NaN -> 0.
__device__ __forceinline__ double max0(double x) { return x > 0.0 ? x : 0.0; }

// One axis of one disc: position and velocity along x or y.  The reference's
// 4-vector RK4 (cpp:70-116) never couples the axes, so a (p, v) pair carries a
// complete, independent chain.
struct Chain {
  double p, v;
};

// a / tau, correctly rounded, for the per-batch constants tau_player / tau_puck.
//
// FAST = false: the compiler's IEEE sequence (11 instructions, three of them quarter rate).
// FAST = true : reciprocal multiply plus two Markstein corrections, 5 full-rate instructions:
//     q0 = RN(a*y)            y = RN(1/tau) from the host
//     r0 = RN(a - tau*q0)     q1 = RN(q0 + r0*y)      -> |q1 - a/tau| <= (1/2 + 2^-51) ulp
//     r1 =    a - tau*q1      q2 = RN(q1 + r1*y)      (r1 exact: q1 is faithful)
//   q2's pre-rounding value is a/tau + (a/tau - q1)*eps with |eps| <= B*2^-106 (B = tau's 53-bit
//   significand), i.e. closer to a/tau than any rounding boundary can be (a quotient of two
//   53-bit numbers stays >= 1/(B*2^53) away from every midpoint) as long as B is not within a
//   few units of 2^53 - 1; so q2 = RN(a/tau).  Preconditions, enforced by the callers:
//   tau normal with 2^-100 <= |tau| <= 2^100 and a significand not ending in all ones (host,
//   fast_division_ok); a == 0 or 2^-760 <= |a| <= 2^520 so r1 never underflows (lane_step's
//   entry guard).  DESIGN.md "Division by tau" has the full argument; rgs_selftest_divtau and
//   tests/test_gpu_parity.py check it against IEEE '/' on adversarial near-midpoint inputs.
// DM: 0 = IEEE '/', 1 = the five steps above, 2 = tau is +-2^k so a*(1/tau) is the exact quotient,
// 3 = ONE correction step (q1 above), used only for taus the host has certified: q1 = RN(a/tau + d)
// with |d| < 2^-51 ulp, so q1 can differ from RN(a/tau) only if a/tau lies within 2^-51 ulp of a
// rounding midpoint, i.e. for significands A of a with |A*2^53 - M*B| <= 7 (A >= B) or
// |A*2^54 - M*B| <= 7 (A < B), M odd; for a given tau those finitely many A are enumerated by modular
// inverse and each is checked on the host (one_step_certified, rgs_hip.hip).  A power-of-two tau is
// trivially fine (r0 = 0).  (bool template arguments still work: false = 0, true = 1.)
template <int DM>
__device__ __forceinline__ double div_tau(double a, double tau, double inv) {
  if (DM == 2) return a * inv;
  if (DM == 3) {
    const double q0 = a * inv;
    const double r0 = __builtin_fma(-tau, q0, a);
    return __builtin_fma(r0, inv, q0);
  }
  if (DM == 1) {
    const double q0 = a * inv;
    const double r0 = __builtin_fma(-tau, q0, a);
    const double q1 = __builtin_fma(r0, inv, q0);
    const double r1 = __builtin_fma(-tau, q1, a);
    return __builtin_fma(r1, inv, q1);
  }
  return a / tau;
}

#ifndef GSIM_RK4_FMA_NUM
#define GSIM_RK4_FMA_NUM 1
#endif
#if GSIM_RK4_FMA_NUM
#define GSIM_RK4_NUM(nc, v, u) __builtin_fma((nc), (v), (u))
#else
#define GSIM_RK4_NUM(nc, v, u) ((u) + (nc) * (v))
#endif

// RK4_player / RK4_puck (src/GameSim.cpp:70-88, 98-116) restricted to one axis.
//   f(x) = (v, (u - coef*v)/tau): coef = 1 for players (cpp:66-67 with sigma = 0;
//   1.0*v is exact), 0.2 for the puck (cpp:94-95, whose u is always 0).
// The position rows of x2..x4 never feed f, so only their velocity rows exist here.
// cpp:87 combines x + (k1 + k2*2 + k3*2 + k4)*(dt/6); k*2 is exact, so fma(k, 2, acc) rounds
// exactly once like the reference's multiply-then-add and is bit-identical to it.
// The numerator u - coef*v is ONE fma here: for a player coef = 1, the product is exact and fma(-1, v, u) = RN(u - v);
// for the puck u = 0 (always, cpp:188) and fma(-0.2, v, 0) = -RN(0.2*v) = RN(0 - RN(0.2*v)), zeros included
// (v = +-0 gives +0 both ways).  Callers keep to (coef == 1) or (u == 0).  One dependent operation less per stage
// on the wave pass' serial chains.
template <int DM>
__device__ __forceinline__ Chain rk4_axis(Chain x, double u, double coef, double tau, double inv, double h) {
  const double half = h / 2.0;
  const double nc = -coef;
  const double k1p = x.v;
  const double k1v = div_tau<DM>(GSIM_RK4_NUM(nc, x.v, u), tau, inv);
  const double x2v = x.v + k1v * half;
  const double k2p = x2v;
  const double k2v = div_tau<DM>(GSIM_RK4_NUM(nc, x2v, u), tau, inv);
  const double x3v = x.v + k2v * half;
  const double k3p = x3v;
  const double k3v = div_tau<DM>(GSIM_RK4_NUM(nc, x3v, u), tau, inv);
  const double x4v = x.v + k3v * h;
  const double k4p = x4v;
  const double k4v = div_tau<DM>(GSIM_RK4_NUM(nc, x4v, u), tau, inv);
  const double w = h / 6.0;
  Chain r;
  r.p = x.p + (__builtin_fma(k3p, 2.0, __builtin_fma(k2p, 2.0, k1p)) + k4p) * w;   // cpp:87
  r.v = x.v + (__builtin_fma(k3v, 2.0, __builtin_fma(k2v, 2.0, k1v)) + k4v) * w;
  return r;
}

// RK4_player with the noise term of f_player (cpp:62-88): xdot_v = (u - v)/tau + w_stdev * N(0,1), two
// draws per f_player call in the order vx, vy, i.e. this axis' draw of stage s is nz[2*s].  The puck
// passes sigma = 0 (f_puck draws nothing; adding +0.0 changes no value).  IEEE division.
// DM != 0 (device-generator kernel only): the division by tau through div_tau<DM>, which equals the IEEE quotient whenever
// its magnitude window holds; the callers check the window per chain (rk4_noise_dyn / fast_division_safe, gsim_wave.hpp).
template <int DM = 0>
__device__ __forceinline__ Chain rk4_axis_noise(Chain x, double u, double coef, double tau, double h, double sigma,
                                                const double *nz, double inv = 0.0) {
  const double half = h / 2.0;
  const double k1p = x.v;
  const double k1v = div_tau<DM>(u - coef * x.v, tau, inv) + sigma * nz[0];
  const double x2v = x.v + k1v * half;
  const double k2p = x2v;
  const double k2v = div_tau<DM>(u - coef * x2v, tau, inv) + sigma * nz[2];
  const double x3v = x.v + k2v * half;
  const double k3p = x3v;
  const double k3v = div_tau<DM>(u - coef * x3v, tau, inv) + sigma * nz[4];
  const double x4v = x.v + k3v * h;
  const double k4p = x4v;
  const double k4v = div_tau<DM>(u - coef * x4v, tau, inv) + sigma * nz[6];
  const double w = h / 6.0;
  Chain r;
  r.p = x.p + (__builtin_fma(k3p, 2.0, __builtin_fma(k2p, 2.0, k1p)) + k4p) * w;
  r.v = x.v + (__builtin_fma(k3v, 2.0, __builtin_fma(k2v, 2.0, k1v)) + k4v) * w;
  return r;
}

// Snapshot of arena a's state rows into the trajectory buffer (see StepArgs::traj).  Called by the
// `nl` lanes that own the arena after its step k0 (0-based) is complete and visible.
//
// With StepArgs::rec the same lanes also write the reference's log record of the step (cpp:161-170): rows t, TAS, TBS,
// puck x y, every player's x y, then the P "damage.row(k)" blocks exactly as the reference's logVectors writes them -- P
// contiguous doubles starting at flat COLUMN-MAJOR element k (utils.h:75 takes row(k).data() of a column-major Matrix and
// writes rows*cols doubles from there), i.e. block k, entry q = damage(row (k+q) % P, column (k+q) / P) -- and a last row
// "executed": 1.0 if this step advanced the arena's clock (a decided game logs nothing, cpp:121-123), judged against the
// previous record of the launch (the launch's first step: against StepArgs::tprev).
__device__ __forceinline__ bool recording(const StepArgs &p) { return p.traj != nullptr; }   // (records: REC kernels only, which check p.rec themselves)

// REC: only the kernels the host launches for recorded runs carry the record code (k_step_hybrid<.., REC = true>, the
// noise kernels): inlined into every kernel it cost the pooled 2v2 kernels 100 registers and 5 % (measured).
template <int N, bool COH = false, bool REC = false>
__device__ __forceinline__ void snapshot_state(const StepArgs &p, long long a, int k0, int lane, int nl) {
  constexpr int P = 2 * N, S = 2 + 4 * (P + 1), R = 3 + 2 + 2 * P + P * P + 1;
  const long long B = p.B;
  if (REC && p.rec != nullptr) {
    double *dst = p.rec + (long long)k0 * R * B;
    for (int r = lane; r < R; r += nl) {
      double v;
      if (r == 0) v = gld_at<COH>(p.tcur, a, p.n_tcur);
      else if (r < 5) v = gld_at<COH>(p.state, (long long)(r - 1) * B + a, p.n_state);               // TAS, TBS, puck x, y
      else if (r < 5 + 2 * P) { const int e = (r - 5) >> 1, ax = (r - 5) & 1; v = gld_at<COH>(p.state, (long long)(6 + 4 * e + ax) * B + a, p.n_state); }
      else if (r < R - 1) { const int flat = (r - 5 - 2 * P) / P + (r - 5 - 2 * P) % P; v = gld_at<COH>(p.damage, (long long)((flat % P) * P + flat / P) * B + a, p.n_damage); }
      else {
        const double before = k0 == 0 ? p.tprev[a] : (p.rec + (long long)(k0 - 1) * R * B)[a];
        v = gld_at<COH>(p.tcur, a, p.n_tcur) != before ? 1.0 : 0.0;
      }
      dst[(long long)r * B + a] = v;
    }
  }
  if (p.traj == nullptr || (k0 + 1) % p.traj_every != 0) return;
  double *dst = p.traj + (long long)((k0 + 1) / p.traj_every - 1) * S * B;
  for (int r = lane; r < S; r += nl) dst[(long long)r * B + a] = gld_at<COH>(p.state, (long long)r * B + a, p.n_state);
}

// Orders LDS traffic between lanes of ONE wavefront: LDS operations of a wave
// execute in order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace gsim
