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
  const GridView<N, false> G{lds + (wave * GPW + grp) * L::DOUBLES, nullptr};
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
  const GridView<N, false> G{lds + grp * WaveLayout<N>::DOUBLES, nullptr};
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
                              SPLITV ? p.vgrid + ((long long)blockIdx.x * GPW + grp) * GridView<N, SPLITV>::GV_DOUBLES : nullptr};
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
enum LaunchKind : int { LAUNCH_WAVE = 0, LAUNCH_WAVE_NOISE = 1, LAUNCH_HYBRID = 2, LAUNCH_HYBRID2 = 3 };

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
