// gsim_rng.hpp -- the reference's noise generator on the device (second option).
//
// The reference draws w_stdev_ * N(0,1) twice per f_player call (src/GameSim.cpp:66-67) from
// std::default_random_engine (include/GameSim.h:72) through std::normal_distribution<double> (h:73).  With the
// build's compiler these are libstdc++ 11's (NOT part of /root/reference; restated from their published algorithm,
// /usr/include/c++/11/bits/random.h:1604 and random.tcc:1802-1835, 3348-3380):
//
//   engine        minstd_rand0: x <- 16807 * x mod (2^31 - 1), output x; seed s -> s mod m, 0 -> 1
//   canonical     generate_canonical<double, 53>: two engine outputs d1, d2;
//                 c = ((d1 - 1) + (d2 - 1) * r) / fl(r * r),  r = 2147483646;  c >= 1 -> nextafter(1, 0)
//   normal pair   Marsaglia's polar method: x = 2 c1 - 1, y = 2 c2 - 1, r2 = x x + y y, again while r2 > 1 or r2 == 0;
//                 mult = sqrt(-2 log(r2) / r2); the call returns y * mult, the next call the cached x * mult.
//
// One ATTEMPT of the polar loop consumes exactly four engine outputs whether it is accepted or not, so attempt i of a
// stream reads outputs 4i+1 .. 4i+4: their positions are known in advance and the engine can jump (x_{n+k} = a^k x_n mod
// m).  The 64 lanes of a wavefront therefore evaluate 64 consecutive attempts at once, a ballot compacts the accepted
// ones in order, and the stream of normals comes out in exactly the reference's order.  Every integer step and every
// +, *, / and sqrt is the reference's (IEEE, same association).  log() is the HOST's: gsim_hostlog.hpp restates glibc's
// table-driven log with the constants of the installed libm.so.6 and is accepted at build time only if it reproduces the
// host's log() bit for bit (GSIM_HOST_LOG = 1); the normals are then the reference's bit for bit and so is every noisy
// trajectory, at any ensemble size (rgs_noise_is_exact() = 1).  Where that check does not pass (another C library), the
// device library's log is used (<= 1 ulp from glibc's): normals within 2 ulp, noisy trajectories match STATISTICALLY only.
//
// All RK4 calls of a player consume 8 draws, an even number: between calls the distribution never holds a cached second
// value, so the generator's whole state between steps is the engine's x (plus, here, the normals a wavefront generated
// ahead but has not consumed yet, which are kept in order and consumed first).
#pragma once
#include "gsim_common.hpp"
#include "gsim_hostlog.hpp"

namespace gsim {

constexpr unsigned kLcgA = 16807u, kLcgM = 2147483647u;      // minstd_rand0
constexpr int kRngSlack = 256;                               // room for one iteration's overshoot (<= 128) plus what a step leaves unconsumed
constexpr int kLogTabDoubles = 256;                          // the host log's {1/c, log c} table, kept in LDS by the kernels that draw
constexpr int kRngLeft = 256;                                // normals an arena may carry from one launch to the next (global memory)
template <int N>
constexpr int rng_window() { return 80 * 2 * N + kRngSlack; }   // doubles of LDS per arena: ten columns' draws of all players (an event's blocks need less)

__host__ __device__ constexpr unsigned lcg_mulmod(unsigned x, unsigned c) {     // x * c mod (2^31 - 1); x, c < m
  unsigned long long v = (unsigned long long)x * c;                              // < 2^62
  v = (v & kLcgM) + (v >> 31);                                                   // 2^31 = 1 (mod m)
  v = (v & kLcgM) + (v >> 31);
  return (unsigned)(v >= kLcgM ? v - kLcgM : v);
}
__host__ __device__ constexpr unsigned lcg_pow(unsigned e) {                     // a^e mod m
  unsigned r = 1u, b = kLcgA;
  while (e) { if (e & 1u) r = lcg_mulmod(r, b); b = lcg_mulmod(b, b); e >>= 1; }
  return r;
}
__host__ __device__ inline unsigned lcg_pow_rt(unsigned e) { return lcg_pow(e); }   // the same at run time

__host__ __device__ inline unsigned lcg_seed(long long s) {                      // linear_congruential_engine::seed, random.tcc:118-127
  const unsigned long long u = (unsigned long long)s;                            // result_type is 64-bit unsigned on this ABI
  const unsigned x = (unsigned)(u % kLcgM);
  return x == 0u ? 1u : x;
}

// generate_canonical<double, 53> from two consecutive engine outputs (random.tcc:3348-3380).
__device__ __forceinline__ double lcg_canonical(unsigned d1, unsigned d2) {
  const double r = 2147483646.0;                                                 // max() - min() + 1
  const double tmp = 4611686009837453316.0;                                      // fl(r * r): the literal rounds to the same double
  double sum = (double)(d1 - 1u);
  sum = sum + (double)(d2 - 1u) * r;
  // sum / tmp, correctly rounded: reciprocal + two Markstein corrections (div_tau<1>, gsim_common.hpp; exact for any
  // divisor whose significand does not end in a run of ones and a dividend that is 0 or in [1, 2^63) -- both hold
  // here), 5 full-rate instructions instead of the IEEE sequence's 11; the bit-equality test of the uniforms covers it
  double ret = div_tau<1>(sum, tmp, 1.0 / tmp);
  if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;                                    // nextafter(1, 0)
  return ret;
}

// The generator of ONE arena, owned by one wavefront (all 64 lanes call every member together).
// buf: LDS, capacity cap doubles; have: normals in buf (in stream order, not yet consumed); x: engine state before the
// next attempt that has not been evaluated.  Wave-uniform.
struct DevRng {
  double *buf;
  int cap, have;
  unsigned x;
  unsigned lane_jump;                                                            // a^(4 lane) mod m, set once per kernel
  const double *logtab;                                                          // LDS copy of the host log's table (load_log_table)

  // Appends normals until at least `need` are buffered (need + 128 <= cap; the callers keep need <= cap - kRngSlack).
  __device__ __forceinline__ void ensure(int need, int lane) {
    constexpr unsigned kJump256 = lcg_pow(256u);
    while (have < need) {
      const unsigned s = lcg_mulmod(x, lane_jump);                               // state before this lane's attempt: x * a^(4 lane)
      const unsigned d1 = lcg_mulmod(s, kLcgA), d2 = lcg_mulmod(d1, kLcgA), d3 = lcg_mulmod(d2, kLcgA), d4 = lcg_mulmod(d3, kLcgA);
      const double vx = 2.0 * lcg_canonical(d1, d2) - 1.0;                       // random.tcc:1820-1821
      const double vy = 2.0 * lcg_canonical(d3, d4) - 1.0;
      const double r2 = vx * vx + vy * vy;
      const bool ok = !(r2 > 1.0 || r2 == 0.0);
      const unsigned long long m = __ballot(ok);
      if (ok) {
#if GSIM_HOST_LOG
        const double lg = host_log(r2, logtab);                                  // the host's log(), bit for bit (gsim_hostlog.hpp)
#else
        const double lg = ::log(r2);                                             // the device library's: <= 1 ulp from the host's
#endif
        const double mult = sqrt_ieee(div_ieee(-2.0 * lg, r2));                  // random.tcc:1826
        const int at = have + 2 * __builtin_popcountll(m & ((1ull << lane) - 1ull));
        buf[at] = vy * mult;                                                     // returned first (:1829)
        buf[at + 1] = vx * mult;                                                 // the cached value, returned by the next call (:1827)
      }
      have += 2 * __builtin_popcountll(m);
      x = lcg_mulmod(x, kJump256);
    }
    wave_sync();
  }

  // Drops the first n normals (consumed by the step), keeping the rest in order.
  __device__ __forceinline__ void consume(int n, int lane) {
    const int rest = have - n;                                                   // <= the generator's overshoot + an event's unused tail: < 4 * 64
    wave_sync();
    double keep[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) keep[q] = (lane + 64 * q < rest) ? buf[n + lane + 64 * q] : 0.0;
    wave_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) if (lane + 64 * q < rest) buf[lane + 64 * q] = keep[q];
    have = rest;
    wave_sync();
  }

  // Copies the host log's table into LDS (256 doubles at dst); every thread of the block calls it, then synchronises.
  __device__ static __forceinline__ void load_log_table(double *dst, int tid, int nthreads) {
#if GSIM_HOST_LOG
    for (int i = tid; i < kLogTabDoubles; i += nthreads) dst[i] = kHostLogTabDev[i];
#endif
  }

  __device__ static __forceinline__ unsigned lcg_pow_lane(int lane) {            // a^(4 lane) mod m
    unsigned r = 1u, b = lcg_pow(4u);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) { if ((lane >> bit) & 1) r = lcg_mulmod(r, b); b = lcg_mulmod(b, b); }
    return r;
  }
};

}  // namespace gsim
