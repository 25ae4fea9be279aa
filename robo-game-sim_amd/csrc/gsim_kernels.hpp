// gsim_kernels.hpp -- the __global__ kernels (templates over the team size n) and the per-n launch
// table.  rgs_kernels_n.hip instantiates everything for ONE n (compiled once per n, in parallel);
// rgs_hip.hip (host side + C ABI) only sees the GsimOps table.
#pragma once
#include "gsim_common.hpp"
#include "gsim_wave.hpp"
#include "gsim_lane.hpp"

namespace gsim {

constexpr int kWavesPerBlock = 4;

// Register budget of the wave-per-arena kernel, as waves per SIMD (amdgpu_waves_per_eu): see DESIGN.md 3.1.
#ifndef GSIM_WAVE_WAVES
#define GSIM_WAVE_WAVES 0
#endif
#if GSIM_WAVE_WAVES > 0
#define GSIM_WAVE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(GSIM_WAVE_WAVES, GSIM_WAVE_WAVES)))
#else
#define GSIM_WAVE_OCCUPANCY
#endif

// ---------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------

// One lane-group per arena; n_steps consecutive steps per launch, state round-trips through
// the SoA arrays (L2-resident) between steps.
template <int N, int GW, int FAST, int NOISE = 0>
__global__ __launch_bounds__(64 * kWavesPerBlock) GSIM_WAVE_OCCUPANCY void k_step_wave(StepArgs p) {
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
    if (recording(p) || (NOISE != 0 && p.rec != nullptr)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the commit's stores, then the copy's loads
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      snapshot_state<N, false, NOISE != 0>(p, a, k, lg, GW);
    }
  }
}

// One wavefront per arena with the reference's noise generator ON THE DEVICE (gsim_rng.hpp): n_steps fused steps per
// launch, any number of arenas; bit-identical to the reference's draws where the host's log() could be restated
// (GSIM_HOST_LOG, gsim_hostlog.hpp), statistical parity otherwise (k_step_wave<.., NOISE = 1> is fed with host-drawn tapes).
#ifndef GSIM_RNG_WAVES
#define GSIM_RNG_WAVES GSIM_WAVE_WAVES                       // wavefronts per SIMD of the device-noise kernel (0: the compiler's choice); measured at 2v2 / 3v3: 2 = 3 (+-1 %), 4 is 3 % slower
#endif
#if GSIM_RNG_WAVES > 0
#define GSIM_RNG_OCCUPANCY __attribute__((amdgpu_waves_per_eu(GSIM_RNG_WAVES, GSIM_RNG_WAVES)))
#else
#define GSIM_RNG_OCCUPANCY
#endif
template <int N, int FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) GSIM_RNG_OCCUPANCY void k_step_wave_rng(StepArgs p) {
  extern __shared__ double lds[];
  using L = WaveLayout<N>;
  constexpr int WIN = rng_window<N>();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long a = (long long)blockIdx.x * kWavesPerBlock + wave;
  double *const logtab = lds + kWavesPerBlock * (L::DOUBLES + WIN);
  DevRng::load_log_table(logtab, threadIdx.x, 64 * kWavesPerBlock);
  __syncthreads();
  if (a >= p.B) return;
  double *mine = lds + wave * (L::DOUBLES + WIN);
  const GridView<N, false> G{mine, nullptr GSIM_GRID_DBG(p)};
  DevRng rng;
  rng.buf = mine + L::DOUBLES;
  rng.cap = WIN;
  rng.x = p.rng_x[a];
  rng.have = p.rng_left_n[a];
  rng.lane_jump = DevRng::lcg_pow_lane(lane);
  rng.logtab = logtab;
  for (int i = lane; i < rng.have; i += 64) rng.buf[i] = p.rng_left[a * kRngLeft + i];
  wave_sync();
  for (int k = 0; k < p.n_steps; ++k) {
    arena_step_group<N, 64, FAST, 2, false>(p, p.ctl + (long long)k * p.ctl_stride, a, G, lane, 0, 0xFFFFu | (1u << 16), &rng);
    if (recording(p) || p.rec != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      snapshot_state<N, false, true>(p, a, k, lane, 64);
    }
  }
  wave_sync();
  if (lane == 0) { p.rng_x[a] = rng.x; p.rng_left_n[a] = rng.have < kRngLeft ? rng.have : kRngLeft; }
  for (int i = lane; i < rng.have && i < kRngLeft; i += 64) p.rng_left[a * kRngLeft + i] = rng.buf[i];
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

template <int N, int MODE, bool REC = false>
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
    if (mine) snapshot_state<N, false, REC>(p, a, k, 0, 1);
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
// Pooled hybrid: the lane pass of k_step_hybrid, four wavefronts (one per SIMD of a CU) per workgroup, and
// the flagged arena-steps of the workgroup's 256 arenas go through ONE queue in LDS from which each of the
// four wavefronts claims full rounds.
//
// Why.  In k_step_hybrid a wavefront can only pair up arenas it owns, so most wave-pass rounds run with one or
// two of their GPW lane groups filled (C2: 1.9 arenas per round; a round costs the same with one arena as with
// four), and a wavefront with many flagged arenas makes its 63 other arenas wait while its three neighbours on
// the CU idle or run half-empty rounds of their own.  Pooled over 256 arenas the rounds fill up (C2, 25 steps:
// 17 414 rounds -> about half), the four wavefronts share them, and a wavefront that has finished its steps
// keeps serving its siblings.  The arithmetic of a task is arena_step_group, unchanged.
//
// A chip-wide queue in global memory (all 1024 wavefronts) was built first and measured: device-scope atomics
// on one hot address cost ~1 us each across the eight XCDs and serialise, 4 ms per step.  LDS atomics cost
// ~100 cycles, and everything a workgroup shares stays inside one CU: the hand-over of an arena's state
// between two of its wavefronts needs workgroup-scope fences only (same L1, same L2).
//
// Protocol (workgroup scope):
//   home wavefront, step k:  lane pass; n = flagged lanes.  If n: release fence (its stores of step k-1 are
//     done), pend[w] = n, reserve n tickets (atomic add on tail), write the payloads, then seq[slot] = t + 1.
//     Then claim and run rounds until pend[w] is 0 again; a wavefront with n = 0 lends at most one round.
//   any wavefront, one round: claim up to GPW consecutive tickets by CAS on head; lane group g waits for
//     seq of ticket h + g, reads the payload, acquire fence, runs arena_step_group for that arena at that
//     step, release fence, decrements pend of the arena's home wavefront.
//   An arena is owned by exactly one wavefront at any time: its home between steps, the claimer while its
//     task runs; the home does not touch it again before pend[w] reads 0 (+ acquire fence).
//   Progress: the four wavefronts of a workgroup are resident together by construction; a wavefront never
//     waits while the queue holds an unclaimed ticket, and a claimed ticket is being run.  Spins are
//     bounded anyway (kPoolSpinLimit) and a timeout is reported as an error by the host.
// ---------------------------------------------------------------------------------------
constexpr int kPoolWaves = 4;                                // wavefronts per workgroup = SIMDs per CU
constexpr int kPoolSlots = 256;                              // ring: one slot per arena of the workgroup
constexpr int kPoolSpinLimit = 1 << 24;                      // polls of >= 64 cycles: > 0.4 s, far above any legitimate wait

struct PoolQueue {                                           // lives in LDS behind the grids
  unsigned tail, head, done, pad;
  int pend[kPoolWaves];
  unsigned seq[kPoolSlots];
  unsigned long long task[kPoolSlots];
  unsigned fin[kPoolSlots];                                  // asynchronous form: steps completed by the wave pass, per arena
  int front[kPoolWaves];                                     // asynchronous form: step count of each wavefront's most advanced arena
  int lastk[kPoolSlots];                                     // asynchronous form: the arena's last wave-pass step ...
  unsigned char streak[kPoolSlots];                          // ... and how many consecutive steps up to it had a collision
  int arena[kPoolSlots];                                     // the arena each lane slot of this workgroup steps (StepArgs::perm)
};

template <typename T>
__device__ __forceinline__ T lds_ld(const T *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// payload: arena index inside the workgroup (8 bits) | step within the launch (26) | discs (16) | first column (6)
__device__ __forceinline__ unsigned long long pool_pack(int local, int k, unsigned discmask) {
  return (unsigned long long)(unsigned)local | ((unsigned long long)(unsigned)k << 8) | ((unsigned long long)(discmask & 0xFFFFu) << 34) |
         ((unsigned long long)((discmask >> 16) & 63u) << 50);
}

// One round: claims up to GPW tickets and runs them.  Returns the number of tickets run (0: the queue had nothing to claim).
#ifndef GSIM_POOL_LAGMAX
#define GSIM_POOL_LAGMAX 32
#endif
#ifndef GSIM_POOL_STREAK
#define GSIM_POOL_STREAK 2
#endif
template <int N, int FAST, bool ASYNC>
__device__ __forceinline__ int pool_round(const StepArgs &p, PoolQueue *Q, const GridView<N, false> &G, long long wg_a0, int lane,
                                           int grp, int lg) {
  constexpr int GW = HybridGroups<N>::GW, GPW = HybridGroups<N>::GPW;
  unsigned h = 0u;
  int take = 0;
  if (lane == 0) {
    for (int tries = 0; tries < 8 && take == 0; ++tries) {
      h = lds_ld(&Q->head);
      const int avail = (int)(lds_ld(&Q->tail) - h);
      if (avail <= 0) break;
      const int want = avail < GPW ? avail : GPW;
      unsigned expect = h;
      if (__hip_atomic_compare_exchange_strong(&Q->head, &expect, h + (unsigned)want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP))
        take = want;
    }
  }
  h = (unsigned)__shfl((int)h, 0);
  take = __shfl(take, 0);
  if (take == 0) return 0;
  unsigned long long payload = 0ull;
  if (grp < take) {
    const unsigned t = h + (unsigned)grp;
    const unsigned slot = t & (kPoolSlots - 1);
    if (lg == 0) {
      int spins = 0;                                         // the producer is between its ticket and its publish
      while (lds_ld(&Q->seq[slot]) != t + 1u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kPoolSpinLimit) { atomicAdd(&p.counters[CNT_POOL_TIMEOUT], 1ull); break; }
      }
      payload = lds_ld(&Q->task[slot]);
    }
    payload = __shfl(payload, lane - lg);
  }
  const int local = (int)(payload & 255ull);
  wg_acquire();                                              // the arenas' state, written by the other wavefronts of this CU
  unsigned ev = 0u;                                          // arena_step_group's result: bit 0 events, bit 1 overlap predicates seen
  if (grp < take) {
    const int kk = (int)((payload >> 8) & 0x3FFFFFFull);
    const unsigned dm = (unsigned)((payload >> 34) & 0xFFFFull) | ((unsigned)((payload >> 50) & 63ull) << 16);
    ev = arena_step_group<N, GW, FAST, false, false>(p, p.ctl + (long long)kk * p.ctl_stride, (long long)Q->arena[local], G, lg, lane - lg, dm);
  }
  wg_release();                                              // results stored before the home wavefront is told
  if (grp < take && lg == 0) {
    const int kk = (int)((payload >> 8) & 0x3FFFFFFull);
    if (!ASYNC) {
      __hip_atomic_fetch_add(&Q->pend[local >> 6], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      // Lasting contact (a puck pushed along, discs pressed together, and just as well a puck RESTING against a player
      // or a disc sliding along a wall, which resolve nothing but sit inside an overlap predicate at every column):
      // GSIM_POOL_STREAK steps in a row in which the first scans saw a predicate.  The lane pass' watch is a superset
      // of the predicates, so such an arena is flagged by every lane pass -- a wasted lane slot, a queue round trip and
      // a step of lag each time; instead its next step is posted right here and it stays in the wave pass, at the
      // wave pass' own pace, until a step goes by without a predicate (or it gets ahead of its wavefront).  (Round 2
      // followed up eventful steps only; the driver window's hot arenas -- pucks resting after a spawn jam -- have none.)
      // Bit 31 of the completion word tells its lane to keep waiting.  Not while trajectories are recorded:
      // the snapshot of step kk is taken by the arena's lane and must see the state of step kk.
      const unsigned st = ev == 0u ? 0u : (Q->lastk[local] == kk - 1 ? (unsigned)Q->streak[local] + 1u : 1u);
      Q->streak[local] = (unsigned char)(st > 200u ? 200u : st);
      Q->lastk[local] = kk;
      const bool more = GSIM_POOL_STREAK > 0 && st >= (unsigned)GSIM_POOL_STREAK && !recording(p) && kk + 1 < p.n_steps &&
                        kk + 1 <= lds_ld(&Q->front[local >> 6]) + 1;
      if (more) {
        const unsigned tk = __hip_atomic_fetch_add(&Q->tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&Q->task[tk & (kPoolSlots - 1)], pool_pack(local, kk + 1, 0xFFFFu | (1u << 16)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&Q->seq[tk & (kPoolSlots - 1)], tk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        atomicAdd(&p.counters[CNT_FLAGGED], 1ull);
        if (p.heat != nullptr) atomicAdd(&p.heat[Q->arena[local]], 1u);
      }
      __hip_atomic_store(&Q->fin[local], (unsigned)(kk + 1) | (more ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  return take;                                               // arena-steps run by this round (> 0)
}

template <int N, int MODE>
__global__ __launch_bounds__(64 * kPoolWaves, 1) void k_step_pool(StepArgs p) {
  constexpr int FAST = MODE == DIV_IEEE ? 0 : (MODE == DIV_POW2_ONESTEP ? 3 : 1);   // wave pass division
  extern __shared__ double lds[];
  constexpr int GW = HybridGroups<N>::GW, GPW = HybridGroups<N>::GPW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / GW, lg = lane % GW;
  const long long wg_a0 = (long long)blockIdx.x * (64 * kPoolWaves);
  const bool mine = wg_a0 + threadIdx.x < p.B;
  const long long a = (mine && p.perm != nullptr) ? (long long)p.perm[wg_a0 + threadIdx.x] : wg_a0 + threadIdx.x;
  const GridView<N, false> G{lds + (wave * GPW + grp) * WaveLayout<N>::DOUBLES, nullptr GSIM_GRID_DBG(p)};
  PoolQueue *Q = (PoolQueue *)(lds + kPoolWaves * GPW * WaveLayout<N>::DOUBLES);
  if (threadIdx.x < kPoolSlots) { Q->seq[threadIdx.x] = 0u; Q->arena[threadIdx.x] = (int)a; }
  if (threadIdx.x < kPoolWaves) Q->pend[threadIdx.x] = 0;
  if (threadIdx.x == 0) { Q->tail = 0u; Q->head = 0u; Q->done = 0u; }
  __syncthreads();
  unsigned long long t_lane = 0, t_wave = 0, n_rounds = 0, t_begin = 0;
  if (GSIM_STAMPS && p.stamps) t_begin = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < p.n_steps; ++k) {
    const double *ctl = p.ctl + (long long)k * p.ctl_stride;
    unsigned discmask = 0u;
    unsigned long long t0 = 0, t1 = 0;
    if (GSIM_STAMPS && p.stamps) t0 = __builtin_amdgcn_s_memtime();
    const bool slow = mine && lane_step<N, MODE>(p, ctl, a, discmask);
    const unsigned long long m = __ballot(slow);
    const int n = __builtin_popcountll(m);
    if (n != 0) {
      wg_release();                                          // these arenas' state as of step k is stored before they are offered
      unsigned tk = 0u;
      if (slow && p.heat != nullptr) atomicAdd(&p.heat[a], 1u);
      if (lane == 0) {
        atomicAdd(&p.counters[CNT_FLAGGED], (unsigned long long)n);
        __hip_atomic_store(&Q->pend[wave], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tk = __hip_atomic_fetch_add(&Q->tail, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      tk = (unsigned)__shfl((int)tk, 0) + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (slow) {
        __hip_atomic_store(&Q->task[tk & (kPoolSlots - 1)], pool_pack((int)threadIdx.x, k, discmask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&Q->seq[tk & (kPoolSlots - 1)], tk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (GSIM_STAMPS && p.stamps) { t1 = __builtin_amdgcn_s_memtime(); t_lane += t1 - t0; }
    int waits = 0;
    for (int lent = 0;;) {                                   // serve the queue
      int pend = 0;
      if (n != 0) {
        if (lane == 0) pend = lds_ld(&Q->pend[wave]);
        pend = __shfl(pend, 0);
        if (pend == 0) break;                                // every task of this wavefront's arenas has been run
      } else if (lent != 0) {
        break;                                               // nothing of its own: one round lent per step
      }
      if (pool_round<N, FAST, false>(p, Q, G, wg_a0, lane, grp, lg)) { lent = 1; n_rounds++; waits = 0; }
      else if (n == 0) break;
      else {                                                 // its remaining tasks are running on a sibling: at most one round's time
        __builtin_amdgcn_s_sleep(4);
        if (++waits > kPoolSpinLimit) { if (lane == 0) atomicAdd(&p.counters[CNT_POOL_TIMEOUT], 1ull); break; }
      }
    }
    if (n != 0) wg_acquire();                                // results of its arenas, stored by whoever ran them
    if (GSIM_STAMPS && p.stamps) t_wave += __builtin_amdgcn_s_memtime() - t1;
    if (mine) snapshot_state<N>(p, a, k, 0, 1);
  }
  // this wavefront's arenas are done; keep serving its siblings until all four are (they are resident: bounded)
  if (lane == 0) __hip_atomic_fetch_add(&Q->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // (A timeout here only ends this wavefront's help -- its own arenas are complete -- so it is not an error; a sibling
  // that is really stuck reports its own timeout.  The budget restarts with every served round and finished sibling.)
  unsigned seen_done = 0u;
  for (int waits = 0;;) {
    if (pool_round<N, FAST, false>(p, Q, G, wg_a0, lane, grp, lg)) { n_rounds++; waits = 0; continue; }
    unsigned done = 0u;
    if (lane == 0) done = lds_ld(&Q->done);
    done = (unsigned)__shfl((int)done, 0);
    if (done >= (unsigned)kPoolWaves) break;
    if (done != seen_done) { seen_done = done; waits = 0; }
    __builtin_amdgcn_s_sleep(8);
    if (++waits > kPoolSpinLimit) break;
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

// Asynchronous pooled hybrid.  Same queue, same tasks; what changes is who waits.  In k_step_pool a wavefront
// runs its flagged arenas' rounds right after its lane pass and the 60-odd unflagged arenas of the wavefront
// wait for them.  Here every LANE keeps its own step count: a lane whose arena was flagged posts the task and
// sits out lane passes until the task's completion word shows up; the other lanes of the wavefront go on with
// their next steps, each with its own step's controls.  After a lane pass a wavefront serves one round if the
// queue holds a full one or a task it has seen waiting GSIM_POOL_AGE times (default 1: anything that is there;
// waiting for full rounds -- a round costs the same ~40 k cycles with one arena as with GPW -- was measured: a
// third fewer rounds, 12 % slower, the waiting time is lag), and everything there is when none of its lanes
// can step.  A wavefront ends when each of its arenas has done its n_steps; the lag a lane collects (one or two
// lane passes per flagged step) is what the asynchrony costs, and three rules bound it: the lag bound
// (GSIM_POOL_LAGMAX, below), the follow-up of arenas in lasting contact (GSIM_POOL_STREAK, pool_round) and the
// last few stragglers going through the wave pass (GSIM_POOL_FEW).  Arithmetic per arena-step: lane_step or
// arena_step_group, both unchanged -- the order in which arena-steps of DIFFERENT arenas are run has no
// influence on any value.  DESIGN.md 3.3b has the measurements behind each rule.
#ifndef GSIM_POOL_AGE
#define GSIM_POOL_AGE 1
#endif
#ifndef GSIM_POOL_BUSYMAX
#define GSIM_POOL_BUSYMAX 99                                 // hold the wavefront while this many of its arenas are out (0: GPW); 99: off (measured: C2 -8 %)
#endif
#ifndef GSIM_POOL_FEW
#define GSIM_POOL_FEW 4
#endif
template <int N, int MODE>
__global__ __launch_bounds__(64 * kPoolWaves, 1) void k_step_apool(StepArgs p) {
  constexpr int FAST = MODE == DIV_IEEE ? 0 : (MODE == DIV_POW2_ONESTEP ? 3 : 1);   // wave pass division
  extern __shared__ double lds[];
  constexpr int GW = HybridGroups<N>::GW, GPW = HybridGroups<N>::GPW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / GW, lg = lane % GW;
  const long long wg_a0 = (long long)blockIdx.x * (64 * kPoolWaves);
  const bool mine = wg_a0 + threadIdx.x < p.B;
  const long long a = (mine && p.perm != nullptr) ? (long long)p.perm[wg_a0 + threadIdx.x] : wg_a0 + threadIdx.x;
  const GridView<N, false> G{lds + (wave * GPW + grp) * WaveLayout<N>::DOUBLES, nullptr GSIM_GRID_DBG(p)};
  PoolQueue *Q = (PoolQueue *)(lds + kPoolWaves * GPW * WaveLayout<N>::DOUBLES);
  if (threadIdx.x < kPoolSlots) { Q->seq[threadIdx.x] = 0u; Q->fin[threadIdx.x] = 0u; Q->streak[threadIdx.x] = 0; Q->lastk[threadIdx.x] = -2; Q->arena[threadIdx.x] = (int)a; }
  if (threadIdx.x < kPoolWaves) { Q->pend[threadIdx.x] = 0; Q->front[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { Q->tail = 0u; Q->head = 0u; Q->done = 0u; }
  __syncthreads();
  unsigned long long t_lane = 0, t_wave = 0, n_rounds = 0, t_begin = 0;
  if (GSIM_STAMPS && p.stamps) t_begin = __builtin_amdgcn_s_memtime();
  const int K = p.n_steps;
  int kl = 0;                                                // steps this lane's arena has completed
  bool busy = false;                                         // its step kl is with the wave pass
  unsigned seen_head = 0xFFFFFFFFu;
  int seen_times = 0, waits = 0;
  unsigned long long n_iter = 0, t_main = 0, n_posted = 0, n_served_arenas = 0;
  for (;;) {
    // completions of posted tasks
    unsigned fw = 0u;
    if (busy) fw = lds_ld(&Q->fin[threadIdx.x]);
    const int fk = (int)(fw & 0x7FFFFFFFu);
    const bool moved = busy && fk > kl;
    if (__ballot(moved) != 0ull) wg_acquire();               // those arenas' state, stored by whoever ran them
    if (moved) {
      if ((fw >> 31) == 0u) { busy = false; snapshot_state<N>(p, a, fk - 1, 0, 1); }   // (a step that is followed up: no trajectories then)
      kl = fk;
    }
    const bool active = mine && !busy && kl < K;
    const unsigned long long am = __ballot(active);
    if (am == 0ull && __ballot(busy) == 0ull) break;         // every arena of this wavefront has done its steps
    // Lag bound: an arena GSIM_POOL_LAGMAX steps behind the most advanced arena of its wavefront is waited for
    // (no lane pass until it is back), so a wavefront runs at most n_steps + GSIM_POOL_LAGMAX lane passes however
    // often single arenas of it are flagged; arenas in lasting contact make their wavefront run in step with them.
    int f = mine ? kl : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int g = __shfl_xor(f, o); f = g > f ? g : f; }
    if (lane == 0) __hip_atomic_store(&Q->front[wave], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long bm = __ballot(busy);
    const bool hold = __ballot(busy && f - kl >= GSIM_POOL_LAGMAX) != 0ull ||
                      __builtin_popcountll(bm) >= (GSIM_POOL_BUSYMAX > 0 ? GSIM_POOL_BUSYMAX : GPW);
    unsigned long long t0 = 0, t1 = 0;
    if (GSIM_STAMPS && p.stamps) t0 = __builtin_amdgcn_s_memtime();
    unsigned discmask = 0xFFFFu | (1u << 16);                // a step posted without a lane pass: every disc, from column 1
    bool slow = false;
    // a lane pass costs the same for one arena as for 64: the last few stragglers go through the wave pass
    const bool few = !recording(p) && __builtin_popcountll(am) <= GSIM_POOL_FEW;
    if (am != 0ull && !few && !hold) {
      n_iter++;
      unsigned dm = 0u;
      slow = active && lane_step<N, MODE>(p, p.ctl + (long long)kl * p.ctl_stride, a, dm);
      if (slow) discmask = dm;
      if (active && !slow) { snapshot_state<N>(p, a, kl, 0, 1); kl++; }
    }
    const bool post = slow || (few && !hold && active);
    const unsigned long long m = __ballot(post);
    if (m != 0ull) {
      const int n = __builtin_popcountll(m);
      if (GSIM_STAMPS) n_posted += (unsigned long long)n;
      wg_release();                                          // these arenas' state as of their step is stored before they are offered
      unsigned tk = 0u;
      if (lane == 0) {
        atomicAdd(&p.counters[CNT_FLAGGED], (unsigned long long)n);
        tk = __hip_atomic_fetch_add(&Q->tail, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      tk = (unsigned)__shfl((int)tk, 0) + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (slow && p.heat != nullptr) atomicAdd(&p.heat[a], 1u);
      if (post) {
        __hip_atomic_store(&Q->task[tk & (kPoolSlots - 1)], pool_pack((int)threadIdx.x, kl, discmask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&Q->seq[tk & (kPoolSlots - 1)], tk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        busy = true;
      }
    }
    if (GSIM_STAMPS && p.stamps) { t1 = __builtin_amdgcn_s_memtime(); t_lane += t1 - t0; }
    // serve: a full round, or a task that has been seen waiting before, or anything at all when this
    // wavefront cannot take a lane pass now
    const bool can_step = !hold && __builtin_popcountll(__ballot(mine && !busy && kl < K)) > (!recording(p) ? GSIM_POOL_FEW : 0);
    bool served = false;
    for (;;) {
      unsigned h = 0u;
      int avail = 0;
      if (lane == 0) { h = lds_ld(&Q->head); avail = (int)(lds_ld(&Q->tail) - h); }
      h = (unsigned)__shfl((int)h, 0);
      avail = __shfl(avail, 0);
      if (avail <= 0) { seen_times = 0; seen_head = 0xFFFFFFFFu; break; }
      if (h == seen_head) seen_times++; else { seen_head = h; seen_times = 1; }
      if (avail < GPW && can_step && seen_times < GSIM_POOL_AGE) break;
      if (const int got = pool_round<N, FAST, true>(p, Q, G, wg_a0, lane, grp, lg)) { n_rounds++; served = true; if (GSIM_STAMPS) n_served_arenas += (unsigned long long)got; }
      seen_times = 0; seen_head = 0xFFFFFFFFu;
      if (can_step) break;                                   // one round per lane pass while there are lane passes to do
    }
    if (!can_step && !served && m == 0ull) {                 // what this wavefront waits for is running elsewhere
      if (__ballot(busy && (int)(lds_ld(&Q->fin[threadIdx.x]) & 0x7FFFFFFFu) > kl) == 0ull) {
        __builtin_amdgcn_s_sleep(4);
        if (++waits > kPoolSpinLimit) { if (lane == 0) atomicAdd(&p.counters[CNT_POOL_TIMEOUT], 1ull); break; }
      } else {
        waits = 0;                                           // a completion came in: the bound measures ONE stall
      }
    } else {
      waits = 0;                                             // a lane pass, a posted task or a served round is progress
    }
    if (GSIM_STAMPS && p.stamps) t_wave += __builtin_amdgcn_s_memtime() - t1;
  }
  if (GSIM_STAMPS && p.stamps) t_main = __builtin_amdgcn_s_memtime() - t_begin;
  // this wavefront's arenas are done; keep serving its siblings until all four are (they are resident: bounded)
  if (lane == 0) __hip_atomic_fetch_add(&Q->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // (as in k_step_pool: a timeout here only ends this wavefront's help and is not an error)
  unsigned seen_done = 0u;
  for (int spins = 0;;) {
    if (const int got = pool_round<N, FAST, true>(p, Q, G, wg_a0, lane, grp, lg)) { n_rounds++; spins = 0; if (GSIM_STAMPS) n_served_arenas += (unsigned long long)got; continue; }
    unsigned done = 0u;
    if (lane == 0) done = lds_ld(&Q->done);
    done = (unsigned)__shfl((int)done, 0);
    if (done >= (unsigned)kPoolWaves) break;
    if (done != seen_done) { seen_done = done; spins = 0; }
    __builtin_amdgcn_s_sleep(8);
    if (++spins > kPoolSpinLimit) break;
  }
  if (GSIM_STAMPS && p.stamps && lane == 0) {
    const unsigned long long t_all = __builtin_amdgcn_s_memtime() - t_begin;
    atomicAdd(&p.stamp_out[STAMP_LANE], t_lane);
    atomicAdd(&p.stamp_out[STAMP_WAVE], t_wave);
    atomicAdd(&p.stamp_out[STAMP_ROUNDS], n_rounds);
    atomicAdd(&p.stamp_out[STAMP_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_TOTAL], t_all);
    atomicMax(&p.stamp_out[STAMP_MAX_WAVE], t_wave);
    atomicMax(&p.stamp_out[STAMP_MAX_ROUNDS], n_iter);   // here: lane passes of the slowest wavefront
    if (p.wave_log != nullptr) {                         // per-wavefront record for tools/wave_log.py
      unsigned long long *w = p.wave_log + 8ull * ((unsigned long long)blockIdx.x * kPoolWaves + wave);
      w[0] = t_all; w[1] = t_lane; w[2] = t_wave; w[3] = n_iter; w[4] = n_rounds; w[5] = t_main; w[6] = n_posted; w[7] = n_served_arenas;
    }
  }
}

// Hybrid with two lanes per arena (gsim_lane.hpp, lane_step_split): 32 arenas per wave.
// Which team sizes use the split grid in the two-lane hybrid: +14..18 % at 6v6/7v7 where it doubles
// the grids per block; at 4v4 it makes three 20-lane groups fit; 5v5 (two grids fit either way) loses
// 5 % with it.
template <int N>
struct Hybrid2Split {
  static constexpr bool value = (GSIM_H2_SPLIT_BIG && N >= 6) || (GSIM_H2_SPLIT_4 && N == 4) || (GSIM_H2_SPLIT_5 && N == 5);
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

// What the host needs from one n, resolved at run time through gsim_ops(n).
enum LaunchKind : int { LAUNCH_WAVE = 0, LAUNCH_WAVE_NOISE = 1, LAUNCH_HYBRID = 2, LAUNCH_HYBRID2 = 3, LAUNCH_POOL = 4, LAUNCH_APOOL = 5,
                        LAUNCH_WAVE_RNG = 6, LAUNCH_HYBRID_REC = 7 };

struct GsimOps {
  // kind: LaunchKind; mode: DivMode (hybrids) or the wave pass division level 0/1/3 (wave kernels)
  hipError_t (*step)(const StepArgs &args, hipStream_t stream, int kind, int mode);
  hipError_t (*reset)(double *state, double *tcur, double *damage, const double *x0, int x0_batched, long long B,
                      hipStream_t stream);
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
