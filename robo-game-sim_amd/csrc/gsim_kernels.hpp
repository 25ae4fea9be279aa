// gsim_kernels.hpp -- the __global__ kernels (templates over the team size n) and the per-n launch
// table.  rgs_kernels_n.hip instantiates everything for ONE n (compiled once per n, in parallel);
// rgs_hip.hip (host side + C ABI) only sees the GsimOps table.
#pragma once
#include "gsim_common.hpp"
#include "gsim_wave.hpp"
#include "gsim_lane.hpp"

namespace gsim {

constexpr int kWavesPerBlock = 4;

// ---------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------

// One lane-group per arena; n_steps consecutive steps per launch, state round-trips through
// the SoA arrays (L2-resident) between steps.
template <int N, int GW, int FAST, bool NOISE = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_step_wave(StepArgs p) {
  extern __shared__ double lds[];
  using L = WaveLayout<N>;
  constexpr int GPW = 64 / GW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / GW, lg = lane % GW;
  const long long a = ((long long)blockIdx.x * kWavesPerBlock + wave) * GPW + grp;
  if (a >= p.B) return;
  const GridView<N, false> G{lds + (wave * GPW + grp) * L::DOUBLES, nullptr GSIM_GRID_DBG(p)};
  for (int k = 0; k < p.n_steps; ++k) {
    arena_step_group<N, GW, FAST, NOISE, false>(p, p.ctl + (long long)k * p.ctl_stride, a, G, lg, lane - lg, 0xFFFFu | (1u << 16));
    if (p.traj != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the commit's stores, then the copy's loads
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      snapshot_state<N>(p, a, k, lg, GW);
    }
  }
}

// Hybrid: every lane streams one arena through the step (gsim_lane.hpp); arenas whose overlap
// watch fired are then re-run, one at a time, by the whole wave with the grid in LDS
// (gsim_wave.hpp).  One wave per block: waves never wait for each other, so an arena stuck in
// a 49-event step only delays the 63 arenas that share its wave.
// Lane-group width of the hybrid's wave pass: the narrowest power of two that holds the 2E
// (disc, axis) chains, so 4 (n <= 3) or 2 (n >= 4) flagged arenas are re-run side by side.
template <int N, bool SPLITV = false, int BUDGET_KB = 40>
struct HybridGroups {
  // narrowest group (a multiple of 4 lanes, so the carry-out's DPP quad stays aligned) holding the 2E chains
  static constexpr int GW = (2 * (2 * N + 1) <= 16 && !GSIM_WIDE_GROUPS) ? 16
                            : ((2 * (2 * N + 1) <= 20 && SPLITV) ? 20 : 32);
  // Arenas re-run side by side = LDS grids per block.  The kernels' register footprint allows 4 waves
  // per CU, so a block may use at most 160 KB / 4 = 40 KB of LDS or residency drops below one wave
  // per SIMD (measured at 7v7: two 24.9 KB grids -> 3 blocks per CU, 1.48e8; one grid -> 1.79e8).
  static constexpr int GRID_BYTES = 8 * GridView<N, SPLITV>::LDS_DOUBLES;
  static constexpr int FIT = (BUDGET_KB * 1024) / GRID_BYTES;
  static constexpr int GPW = (64 / GW) < FIT ? (64 / GW) : (FIT < 1 ? 1 : FIT);
};

template <int N, int MODE>
__global__ __launch_bounds__(64, GSIM_H1_WAVES) void k_step_hybrid(StepArgs p) {
  constexpr int FAST = MODE == DIV_IEEE ? 0 : (MODE == DIV_POW2_ONESTEP ? 3 : 1);   // wave pass division
  extern __shared__ double lds[];
  constexpr int GW = HybridGroups<N>::GW, GPW = HybridGroups<N>::GPW;
  const int lane = threadIdx.x;
  const int grp = lane / GW, lg = lane % GW;
  const long long a0 = (long long)blockIdx.x * 64;
  const long long a = a0 + lane;
  const bool mine = a < p.B;
  const GridView<N, false> G{lds + grp * WaveLayout<N>::DOUBLES, nullptr GSIM_GRID_DBG(p)};
  unsigned long long t_lane = 0, t_wave = 0, n_rounds = 0, t_begin = 0;
  if (GSIM_STAMPS && p.stamps) t_begin = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < p.n_steps; ++k) {
    const double *ctl = p.ctl + (long long)k * p.ctl_stride;
    unsigned discmask = 0u;
    unsigned long long t0 = 0;
    if (GSIM_STAMPS && p.stamps) t0 = __builtin_amdgcn_s_memtime();
    const bool slow = mine && lane_step<N, MODE>(p, ctl, a, discmask);
    unsigned long long m = __ballot(slow);
    if (m != 0ull && lane == 0) atomicAdd(&p.counters[CNT_FLAGGED], (unsigned long long)__builtin_popcountll(m));
    __syncthreads();                                 // lane-pass stores visible to the wave pass
    unsigned long long t1 = 0;
    if (GSIM_STAMPS && p.stamps) { t1 = __builtin_amdgcn_s_memtime(); t_lane += t1 - t0; }
    while (m != 0ull) {                              // GPW flagged arenas per round, one per lane group
      int bit = -1;
      unsigned long long mm = m;
#pragma unroll
      for (int i = 0; i < GPW; ++i) {
        const int b = mm != 0ull ? __builtin_ctzll(mm) : -1;
        if (i == grp) bit = b;
        mm &= mm - 1ull;                             // 0 & anything stays 0
      }
      m = mm;
      n_rounds++;
      const unsigned dm = (unsigned)__shfl((int)discmask, bit < 0 ? 0 : bit);
      if (bit >= 0) arena_step_group<N, GW, FAST, false, false>(p, ctl, a0 + bit, G, lg, lane - lg, dm);
    }
    __syncthreads();
    if (GSIM_STAMPS && p.stamps) t_wave += __builtin_amdgcn_s_memtime() - t1;
    if (mine) snapshot_state<N>(p, a, k, 0, 1);
  }
  if (GSIM_STAMPS && p.stamps && lane == 0) {
    atomicAdd(&p.stamp_out[STAMP_LANE], t_lane);
    atomicAdd(&p.stamp_out[STAMP_WAVE], t_wave);
    atomicAdd(&p.stamp_out[STAMP_ROUNDS], n_rounds);
    const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
    atomicAdd(&p.stamp_out[STAMP_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_WAVE], t_wave);
    atomicMax(&p.stamp_out[STAMP_MAX_ROUNDS], n_rounds);
  }
}

// ---------------------------------------------------------------------------------------
// Pooled hybrid: the lane pass of k_step_hybrid, but the flagged arena-steps of ALL wavefronts go through
// one chip-wide queue and every wavefront re-runs whatever it can claim from it.
//
// Why.  With one wavefront per SIMD (65 536 arenas = 1024 wavefronts) a launch of k_step_hybrid ends with
// its slowest wavefront, and what makes a wavefront slow is the number of wave-pass rounds its own 64
// arenas need (C2, first 25 steps: 17 rounds on average, 40 at most; 1000 steps: 790 / 949) -- rounds that
// mostly run with one or two of their GPW lane groups filled, because a wavefront can only pair up arenas
// it owns.  Here a flagged arena-step is a task (arena, step, disc mask) that any wavefront may run: the
// rounds are full (GPW arenas side by side, they cost the same as a round with one), a wavefront with
// many flagged arenas is helped by those with none, and wavefronts that have finished their steps keep
// serving the queue until the launch is over.  The arithmetic of a task is arena_step_group, unchanged.
//
// Protocol (all of it agent scope; every access to state / tcur / damage is an sc1 access, gld / gst):
//   home wavefront, step k:  lane pass; n = flagged lanes.  If n: drain its stores, q_pending[w] = n,
//     reserve n tickets (atomic add on the tail), write the payloads, drain, publish q_seq[slot] = t + 1.
//     Then claim and run rounds until q_pending[w] is 0 again (somebody -- possibly itself -- has run
//     each of its tasks); a wavefront with n = 0 lends at most one round per step.
//   any wavefront, one round: claim up to GPW consecutive tickets by CAS on the head; lane group g waits
//     for q_seq of ticket h + g, reads the payload, runs arena_step_group for that arena at that step,
//     drains its stores and decrements q_pending of the arena's home wavefront.
//   An arena is owned by exactly one wavefront at any time: its home between steps, the claimer of its
//     ticket while the task runs; the home does not touch it again before q_pending[w] reads 0.
//   Progress: a wavefront never waits while the queue holds an unclaimed ticket (it claims it), a claimed
//     ticket is being run by a live wavefront, and nobody depends on a finished wavefront staying, so
//     every wait is bounded by one round.
// Tickets and q_ctrl[2] are monotonic over the launches of a handle (no per-launch reset); the ring has
// at least B slots, and at most one task per arena is outstanding.
// ---------------------------------------------------------------------------------------
// Every spin is bounded (a protocol error must not hang the GPU): ~2^22 polls of >= 64 cycles, i.e. > 0.1 s,
// four orders of magnitude above any legitimate wait; a timeout is counted and reported by the host as an error.
constexpr int kPoolSpinLimit = 1 << 22;

#ifndef GSIM_POOL_FENCES
#define GSIM_POOL_FENCES 1
#endif

__device__ __forceinline__ void pool_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pool_release() {             // stores of this wavefront visible chip-wide before what follows
  pool_drain();
  if (GSIM_POOL_FENCES) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    pool_drain();
  }
}
__device__ __forceinline__ void pool_acquire() {             // after a matched poll, before loading what it guards
  if (GSIM_POOL_FENCES) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    pool_drain();
  }
  __syncthreads();
}
__device__ __forceinline__ unsigned long long pool_ld(const unsigned long long *q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned pool_ld(const unsigned *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pool_ld(const int *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// payload: arena (26 bits) | step within the launch (16) | discs whose watch fired (16) | first column (6)
__device__ __forceinline__ unsigned long long pool_pack(long long a, int k, unsigned discmask) {
  return (unsigned long long)a | ((unsigned long long)(unsigned)k << 26) | ((unsigned long long)(discmask & 0xFFFFu) << 42) |
         ((unsigned long long)((discmask >> 16) & 63u) << 58);
}

// One round: claims up to GPW tickets and runs them.  Returns false when the queue had nothing to claim.
template <int N, int FAST>
__device__ __forceinline__ bool pool_round(const StepArgs &p, const GridView<N, false> &G, int lane, int grp, int lg) {
  constexpr int GW = HybridGroups<N>::GW, GPW = HybridGroups<N>::GPW;
  unsigned long long h = 0ull;
  int take = 0;
  if (lane == 0) {
    for (int tries = 0; tries < 8 && take == 0; ++tries) {
      h = pool_ld(&p.q_ctrl[1]);
      const long long avail = (long long)(pool_ld(&p.q_ctrl[0]) - h);
      if (avail <= 0) break;
      const int want = avail < GPW ? (int)avail : GPW;
      unsigned long long expect = h;
      if (__hip_atomic_compare_exchange_strong(&p.q_ctrl[1], &expect, h + (unsigned long long)want, __ATOMIC_RELAXED,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        take = want;
    }
  }
  h = __shfl(h, 0);
  take = __shfl(take, 0);
  if (take == 0) return false;
  unsigned long long payload = 0ull;
  if (grp < take) {
    const unsigned long long t = h + (unsigned long long)grp;
    const unsigned long long slot = t & p.q_mask;
    if (lg == 0) {
      int spins = 0;                                         // the producer is between its ticket and its publish: a few hundred cycles
      while (pool_ld(&p.q_seq[slot]) != (unsigned)(t + 1ull)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kPoolSpinLimit) { atomicAdd(&p.counters[CNT_POOL_TIMEOUT], 1ull); break; }
      }
      payload = pool_ld(&p.q_task[slot]);
    }
    payload = __shfl(payload, lane - lg);
  }
  const long long arena = (long long)(payload & ((1ull << 26) - 1ull));
  pool_acquire();                                            // the arenas' state, written by other wavefronts
  if (grp < take) {
    const int kk = (int)((payload >> 26) & 0xFFFFull);
    const unsigned dm = (unsigned)((payload >> 42) & 0xFFFFull) | ((unsigned)((payload >> 58) & 63ull) << 16);
    arena_step_group<N, GW, FAST, false, false, true>(p, p.ctl + (long long)kk * p.ctl_stride, arena, G, lg, lane - lg, dm);
  }
  pool_release();                                            // results written through before the home wavefront is told
  if (grp < take && lg == 0) __hip_atomic_fetch_add(&p.q_pending[arena >> 6], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

template <int N, int MODE>
__global__ __launch_bounds__(64, GSIM_H1_WAVES) void k_step_pool(StepArgs p) {
  constexpr int FAST = MODE == DIV_IEEE ? 0 : (MODE == DIV_POW2_ONESTEP ? 3 : 1);   // wave pass division
  extern __shared__ double lds[];
  constexpr int GW = HybridGroups<N>::GW;
  const int lane = threadIdx.x;
  const int grp = lane / GW, lg = lane % GW;
  const long long a0 = (long long)blockIdx.x * 64;
  const long long a = a0 + lane;
  const bool mine = a < p.B;
  const GridView<N, false> G{lds + grp * WaveLayout<N>::DOUBLES, nullptr GSIM_GRID_DBG(p)};
  int *my_pending = p.q_pending + blockIdx.x;
  const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
  unsigned long long t_lane = 0, t_wave = 0, n_rounds = 0, t_begin = 0;
  if (GSIM_STAMPS && p.stamps) t_begin = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < p.n_steps; ++k) {
    const double *ctl = p.ctl + (long long)k * p.ctl_stride;
    unsigned discmask = 0u;
    unsigned long long t0 = 0, t1 = 0;
    if (GSIM_STAMPS && p.stamps) t0 = __builtin_amdgcn_s_memtime();
    const bool slow = mine && lane_step<N, MODE, true>(p, ctl, a, discmask);
    const unsigned long long m = __ballot(slow);
    const int n = __builtin_popcountll(m);
    if (n != 0) {
      pool_release();                                        // these arenas' state as of step k is out before they are offered
      unsigned long long tk = 0ull;
      if (lane == 0) {
        atomicAdd(&p.counters[CNT_FLAGGED], (unsigned long long)n);
        __hip_atomic_store(my_pending, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __hip_atomic_fetch_add(&p.q_ctrl[0], (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tk = __shfl(tk, 0) + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (slow) __hip_atomic_store(&p.q_task[tk & p.q_mask], pool_pack(a, k, discmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pool_drain();                                          // payloads (and q_pending) before the tickets become readable
      if (slow) __hip_atomic_store(&p.q_seq[tk & p.q_mask], (unsigned)(tk + 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (GSIM_STAMPS && p.stamps) { t1 = __builtin_amdgcn_s_memtime(); t_lane += t1 - t0; }
    int waits = 0;
    for (int lent = 0;;) {                                   // serve the queue
      int pend = 0;
      if (n != 0) {
        if (lane == 0) pend = pool_ld(my_pending);
        pend = __shfl(pend, 0);
        if (pend == 0) break;                                // every task of this wavefront's arenas has been run
      } else if (lent != 0) {
        break;                                               // nothing of its own: one round lent per step
      }
      if (pool_round<N, FAST>(p, G, lane, grp, lg)) { lent = 1; n_rounds++; }
      else if (n == 0) break;
      else {                                                 // its remaining tasks are running elsewhere: at most one round's time
        __builtin_amdgcn_s_sleep(2);
        if (++waits > kPoolSpinLimit) { if (lane == 0) atomicAdd(&p.counters[CNT_POOL_TIMEOUT], 1ull); break; }
      }
    }
    if (n != 0) pool_acquire();                              // results of its arenas, written by whoever ran them
    if (GSIM_STAMPS && p.stamps) t_wave += __builtin_amdgcn_s_memtime() - t1;
    if (mine) snapshot_state<N, true>(p, a, k, 0, 1);
  }
  // This wavefront's arenas are done.  If the whole grid is resident at once (q_stay, set by the host when
  // there are no more wavefronts than SIMDs) it keeps serving the queue until every wavefront has finished
  // its steps -- that is what evens out the end of the launch.  Nobody depends on it staying (a wavefront
  // whose tasks nobody claims runs them itself), so the wait is bounded in time as well: should the grid
  // not be resident after all (another process on the GPU), it leaves after a quarter of its own run time.
  if (lane == 0) __hip_atomic_fetch_add(&p.q_ctrl[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.q_stay) {
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned long long patience = (t_start - t_launch) / 4 + 2000000ull;
    for (;;) {
      if (pool_round<N, FAST>(p, G, lane, grp, lg)) { n_rounds++; continue; }
      unsigned long long done = 0ull;
      if (lane == 0) done = pool_ld(&p.q_ctrl[2]);
      done = __shfl(done, 0);
      if (done >= p.q_done_target || __builtin_amdgcn_s_memtime() - t_start > patience) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (GSIM_STAMPS && p.stamps && lane == 0) {
    const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
    atomicAdd(&p.stamp_out[STAMP_LANE], t_lane);
    atomicAdd(&p.stamp_out[STAMP_WAVE], t_wave);
    atomicAdd(&p.stamp_out[STAMP_ROUNDS], n_rounds);
    atomicAdd(&p.stamp_out[STAMP_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_WAVE], t_wave);
    atomicMax(&p.stamp_out[STAMP_MAX_ROUNDS], n_rounds);
  }
}

// Hybrid with two lanes per arena (gsim_lane.hpp, lane_step_split): 32 arenas per wave.
// Which team sizes use the split grid in the two-lane hybrid: +14..18 % at 6v6/7v7 where it doubles
// the grids per block; at 4v4 it makes three 20-lane groups fit; 5v5 (two grids fit either way) loses
// 5 % with it.
template <int N>
struct Hybrid2Split {
  static constexpr bool value = N >= 6 || N == 4;
  static constexpr int LDS_KB = GSIM_H2_LDS_KB;
};

template <int N, int MODE>
__global__ __launch_bounds__(64, GSIM_H2_WAVES) void k_step_hybrid2(StepArgs p) {
  constexpr int FAST = MODE == DIV_IEEE ? 0 : (MODE == DIV_POW2_ONESTEP ? 3 : 1);   // wave pass division
  constexpr bool SPLITV = Hybrid2Split<N>::value;        // velocity rows of the grid in global scratch
  extern __shared__ double lds[];
  using HG = HybridGroups<N, SPLITV, Hybrid2Split<N>::LDS_KB>;
  constexpr int GW = HG::GW, GPW = HG::GPW;
  const int lane = threadIdx.x;
  const int grp = lane / GW, lg = lane % GW;
  const long long a0 = (long long)blockIdx.x * 32;
  const long long a = a0 + (lane >> 1);
  const int ax = lane & 1;
  const bool mine = a < p.B;
  const GridView<N, SPLITV> G{lds + grp * GridView<N, SPLITV>::LDS_DOUBLES,
                              SPLITV ? p.vgrid + ((long long)blockIdx.x * GPW + grp) * GridView<N, SPLITV>::GV_DOUBLES : nullptr GSIM_GRID_DBG(p)};
  unsigned long long t_lane = 0, t_wave = 0, n_rounds = 0, t_begin = 0;
  if (GSIM_STAMPS && p.stamps) t_begin = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < p.n_steps; ++k) {
    const double *ctl = p.ctl + (long long)k * p.ctl_stride;
    unsigned discmask = 0u;
    unsigned long long t0 = 0, t1 = 0;
    if (GSIM_STAMPS && p.stamps) t0 = __builtin_amdgcn_s_memtime();
    const bool slow = mine && lane_step_split<N, MODE>(p, ctl, a, ax, discmask);
    unsigned long long m = __ballot(slow) & 0x5555555555555555ull;      // one bit per arena (its x lane)
    if (m != 0ull && lane == 0) atomicAdd(&p.counters[CNT_FLAGGED], (unsigned long long)__builtin_popcountll(m));
    __syncthreads();
    if (GSIM_STAMPS && p.stamps) { t1 = __builtin_amdgcn_s_memtime(); t_lane += t1 - t0; }
    while (m != 0ull) {
      int bit = -1;
      unsigned long long mm = m;
#pragma unroll
      for (int i = 0; i < GPW; ++i) {
        const int b = mm != 0ull ? __builtin_ctzll(mm) : -1;
        if (i == grp) bit = b;
        mm &= mm - 1ull;
      }
      m = mm;
      n_rounds++;
      const unsigned dm = (unsigned)__shfl((int)discmask, bit < 0 ? 0 : bit);
      if (bit >= 0) arena_step_group<N, GW, FAST, false, SPLITV>(p, ctl, a0 + (bit >> 1), G, lg, lane - lg, dm);
    }
    __syncthreads();
    if (GSIM_STAMPS && p.stamps) t_wave += __builtin_amdgcn_s_memtime() - t1;
    if (mine) snapshot_state<N>(p, a, k, ax, 2);
  }
  if (GSIM_STAMPS && p.stamps && lane == 0) {
    atomicAdd(&p.stamp_out[STAMP_LANE], t_lane);
    atomicAdd(&p.stamp_out[STAMP_WAVE], t_wave);
    atomicAdd(&p.stamp_out[STAMP_ROUNDS], n_rounds);
    const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
    atomicAdd(&p.stamp_out[STAMP_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_WAVE], t_wave);
    atomicMax(&p.stamp_out[STAMP_MAX_ROUNDS], n_rounds);
  }
}

// GameSim::reset on the device (src/GameSim.cpp:27-39); players spawn in rows (n = 2: +-Y/4).
template <int N>
__global__ void k_reset(double *state, double *tcur, double *damage, const double *x0, int x0_batched,
                        long long B) {
  constexpr int P = 2 * N, E = P + 1;
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= B) return;
  state[0 * B + a] = 0.0;
  state[1 * B + a] = 0.0;
  for (int c = 0; c < 4; ++c) state[(long long)(2 + c) * B + a] = x0_batched ? x0[(long long)c * B + a] : x0[c];
  for (int e = 0; e < P; ++e) {
    const int k = e < N ? e : e - N;
    const double y = kY / 2.0 - (2 * k + 1) * kY / (2.0 * N);
    const double x = e < N ? -kX / 4.0 : kX / 4.0;
    const int row = 6 + 4 * e;
    state[(long long)(row + 0) * B + a] = x;
    state[(long long)(row + 1) * B + a] = y;
    state[(long long)(row + 2) * B + a] = 0.0;
    state[(long long)(row + 3) * B + a] = 0.0;
  }
  for (int q = 0; q < P * P; ++q) damage[(long long)q * B + a] = 0.0;
  tcur[a] = 0.0;
  (void)E;
}

// One log record per arena (src/GameSim.cpp:161-170, include/utils.h:64-76), SoA: rec[R][B] with
// R = 3 + 2 + 2P + P*P + 1: t, TAS, TBS, puck xy, players' xy, the P "damage.row(k)" blocks exactly as
// the reference writes them (P contiguous doubles starting at flat column-major element k), and a
// last row "executed" (1.0 if this step advanced the arena: a decided game logs nothing, cpp:121).
template <int N>
__global__ void k_record(const double *state, const double *tcur, const double *damage, double *tprev,
                         double *rec, long long B) {
  constexpr int P = 2 * N;
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= B) return;
  int r = 0;
  const double t = tcur[a];
  rec[(long long)(r++) * B + a] = t;
  rec[(long long)(r++) * B + a] = state[0 * B + a];
  rec[(long long)(r++) * B + a] = state[1 * B + a];
  rec[(long long)(r++) * B + a] = state[2 * B + a];
  rec[(long long)(r++) * B + a] = state[3 * B + a];
  for (int e = 0; e < P; ++e) {
    rec[(long long)(r++) * B + a] = state[(long long)(6 + 4 * e) * B + a];
    rec[(long long)(r++) * B + a] = state[(long long)(7 + 4 * e) * B + a];
  }
  for (int k = 0; k < P; ++k)
    for (int q = 0; q < P; ++q) {
      const int flat = k + q;                         // column-major flat index: row = flat % P, col = flat / P
      rec[(long long)(r++) * B + a] = damage[(long long)((flat % P) * P + flat / P) * B + a];
    }
  rec[(long long)(r++) * B + a] = (t != tprev[a]) ? 1.0 : 0.0;
  tprev[a] = t;
}


// What the host needs from one n, resolved at run time through gsim_ops(n).
enum LaunchKind : int { LAUNCH_WAVE = 0, LAUNCH_WAVE_NOISE = 1, LAUNCH_HYBRID = 2, LAUNCH_HYBRID2 = 3, LAUNCH_POOL = 4 };

struct GsimOps {
  // kind: LaunchKind; mode: DivMode (hybrids) or the wave pass division level 0/1/3 (wave kernels)
  hipError_t (*step)(const StepArgs &args, hipStream_t stream, int kind, int mode);
  hipError_t (*reset)(double *state, double *tcur, double *damage, const double *x0, int x0_batched, long long B,
                      hipStream_t stream);
  hipError_t (*record)(const double *state, const double *tcur, const double *damage, double *tprev, double *rec,
                       long long B, hipStream_t stream);
  size_t (*vgrid_doubles)(long long B);              // global scratch the two-lane hybrid needs (0: none)
};

}  // namespace gsim

extern "C" const gsim::GsimOps *gsim_ops_n1();
extern "C" const gsim::GsimOps *gsim_ops_n2();
extern "C" const gsim::GsimOps *gsim_ops_n3();
extern "C" const gsim::GsimOps *gsim_ops_n4();
extern "C" const gsim::GsimOps *gsim_ops_n5();
extern "C" const gsim::GsimOps *gsim_ops_n6();
extern "C" const gsim::GsimOps *gsim_ops_n7();
