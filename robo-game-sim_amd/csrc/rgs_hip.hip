// rgs_hip.hip -- kernels and C ABI of the MI355X batched stepper (see include/rgs_hip.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sys/resource.h>
#include <unistd.h>

#include "../../include/rgs_hip.h"
#include "gsim_kernels.hpp"
#include "auto_rates.h"

using namespace gsim;

namespace {

thread_local std::string g_err;

// Frees device allocations of the self-test entry points on every exit path.
struct DeviceScratch {
  std::vector<void *> ptrs;
  ~DeviceScratch() { for (void *q : ptrs) (void)hipFree(q); }
  hipError_t alloc(double **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) ptrs.push_back(*out);
    return e;
  }
};

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(RGS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)

__global__ void k_undecided(const double *state, long long B, double win, uint8_t *out) {
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= B) return;
  out[a] = (state[a] < win && state[B + a] < win) ? 1 : 0;
}

// ---- dealing hot arenas evenly over the pooled kernels' workgroups (see rebalance() below) ----
constexpr int kDealWG = 256;     // arenas per workgroup of the pooled kernels (64 * kPoolWaves)
constexpr int kDealMax = 64;     // arenas one workgroup gives away or takes in per deal

// One block per FULL workgroup w of the pooled kernels: resets its slots to the identity deal, counts the hot arenas
// (heat >= thr) among its own 256, lists the first kDealMax hot and the first kDealMax cold slots in slot order
// (deterministic), and clears the heat for the next window.
__global__ void k_deal_scan(const unsigned *heat_in, unsigned *heat_clear, int *perm, unsigned thr, int *hot_count, int *hot_list,
                            int *cold_list) {
  __shared__ int wave_hot[4], wave_cold[4];
  const int w = blockIdx.x, t = threadIdx.x;
  const long long slot = (long long)w * kDealWG + t;
  const bool hot = heat_in[slot] >= thr;
  perm[slot] = (int)slot;
  heat_clear[slot] = 0u;
  const unsigned long long mh = __ballot(hot), mc = __ballot(!hot);
  const int wave = t >> 6, lane = t & 63;
  if (lane == 0) { wave_hot[wave] = __builtin_popcountll(mh); wave_cold[wave] = __builtin_popcountll(mc); }
  __syncthreads();
  int bh = 0, bc = 0, nh = 0;
  for (int q = 0; q < 4; ++q) { if (q < wave) { bh += wave_hot[q]; bc += wave_cold[q]; } nh += wave_hot[q]; }
  const unsigned long long below = (1ull << lane) - 1ull;
  const int ih = bh + __builtin_popcountll(mh & below), ic = bc + __builtin_popcountll(mc & below);
  if (hot && ih < kDealMax) hot_list[w * kDealMax + ih] = (int)slot;
  if (!hot && ic < kDealMax) cold_list[w * kDealMax + ic] = (int)slot;
  if (t == 0) hot_count[w] = nh;
}

// One block of 1024 threads levels the hot arenas: with total = q * n_wg + r every workgroup should end with q or q + 1.
// Gives, in this order: first everything above q + 1 ("must"), then the one between q + 1 and q ("may"); places
// likewise: first the ones up to q, then the one from q to q + 1.  The k-th give meets the k-th place and
// max(sum of must-gives, sum of must-places) swaps are made, which is enough for both musts and never takes a workgroup
// outside [q, q + 1] (a first version matched surplus above ceil(avg) against places below it in workgroup order: the
// first takers filled up, the last ones stayed at 3 of 11).  perm[hot slot] <-> perm[cold slot]; at most kDealMax per
// workgroup and deal.  Thread t owns workgroups [t*per, (t+1)*per); sums and prefix sums go through LDS (one thread
// walking hot_count[] in global memory cost 0.3 ms).
__global__ void k_deal_match(const int *hot_count, const int *hot_list, const int *cold_list, int n_wg, int per, int *perm,
                             unsigned *heat_tail, int *n_swapped) {
  extern __shared__ int sh[];                           // 4 exclusive prefix arrays of n_wg + 1: must-give, may-give, must-take, may-take
  int *pre[4] = {sh, sh + (n_wg + 1), sh + 2 * (n_wg + 1), sh + 3 * (n_wg + 1)};
  __shared__ int part[4][1024];
  __shared__ int q_sh;
  const int t = threadIdx.x, w0 = t * per, w1 = (w0 + per < n_wg) ? w0 + per : n_wg;
  for (int i = t; i < kDealWG; i += blockDim.x) { perm[n_wg * kDealWG + i] = n_wg * kDealWG + i; heat_tail[i] = 0u; }   // ragged last workgroup
  int sum = 0;
  for (int w = w0; w < w1; ++w) sum += hot_count[w];
  part[0][t] = sum;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) { if (t < off) part[0][t] += part[0][t + off]; __syncthreads(); }
  if (t == 0) q_sh = part[0][0] / n_wg;
  __syncthreads();
  const int q = q_sh;
  auto amounts = [&](int h, int *v) {                   // this workgroup's must-give, may-give, must-take, may-take
    const int cap_give = h < kDealMax ? h : kDealMax, cold = kDealWG - h, cap_take = cold < kDealMax ? cold : kDealMax;
    int mg = h > q + 1 ? h - (q + 1) : 0, yg = h >= q + 1 ? 1 : 0, mt = h < q ? q - h : 0, yt = h <= q ? 1 : 0;
    mg = mg < cap_give ? mg : cap_give; yg = (mg + yg <= cap_give) ? yg : 0;
    mt = mt < cap_take ? mt : cap_take; yt = (mt + yt <= cap_take) ? yt : 0;
    v[0] = mg; v[1] = yg; v[2] = mt; v[3] = yt;
  };
  int mine[4] = {0, 0, 0, 0};
  for (int w = w0; w < w1; ++w) { int v[4]; amounts(hot_count[w], v); for (int c = 0; c < 4; ++c) mine[c] += v[c]; }
  __syncthreads();
  for (int c = 0; c < 4; ++c) part[c][t] = mine[c];
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {            // inclusive Hillis-Steele scans
    int v[4];
    for (int c = 0; c < 4; ++c) v[c] = t >= off ? part[c][t - off] : 0;
    __syncthreads();
    for (int c = 0; c < 4; ++c) part[c][t] += v[c];
    __syncthreads();
  }
  int run[4];
  for (int c = 0; c < 4; ++c) run[c] = part[c][t] - mine[c];
  for (int w = w0; w < w1; ++w) {
    int v[4]; amounts(hot_count[w], v);
    for (int c = 0; c < 4; ++c) { pre[c][w] = run[c]; run[c] += v[c]; }
  }
  if (t == 1023) for (int c = 0; c < 4; ++c) pre[c][n_wg] = part[c][1023];
  __syncthreads();
  const int must_give = pre[0][n_wg], must_take = pre[2][n_wg];
  int m = must_give > must_take ? must_give : must_take;          // <= must + may on either side (sum of h = q*n_wg + r) ...
  const int all_give = must_give + pre[1][n_wg], all_take = must_take + pre[3][n_wg];
  m = m < all_give ? m : all_give;                                // ... unless the per-deal cap of a workgroup cut its amounts
  m = m < all_take ? m : all_take;
  if (t == 0) *n_swapped = m;
  auto owner = [&](const int *p, int r, int &w, int &j) {         // largest w with p[w] <= r; j = r - p[w]
    int lo = 0, hi = n_wg;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p[mid] <= r) lo = mid; else hi = mid; }
    w = lo; j = r - p[lo];
  };
  for (int r = t; r < m; r += blockDim.x) {
    int wg, jg, wt, jt;
    if (r < must_give) owner(pre[0], r, wg, jg);
    else { owner(pre[1], r - must_give, wg, jg); int v[4]; amounts(hot_count[wg], v); jg += v[0]; }       // after its must-gives
    if (r < must_take) owner(pre[2], r, wt, jt);
    else { owner(pre[3], r - must_take, wt, jt); int v[4]; amounts(hot_count[wt], v); jt += v[2]; }       // after its must-takes
    const int hot_slot = hot_list[wg * kDealMax + jg], cold_slot = cold_list[wt * kDealMax + jt];
    perm[hot_slot] = cold_slot;                         // the identity deal was just restored: slot s held arena s
    perm[cold_slot] = hot_slot;
  }
}

// Log writer, device half: rec (K, R, B) -- the record of every step for every arena, row R-1 = 1.0 where the step executed --
// to runs (n_files, K, R-1): arena ids[q]'s executed records one after another (a decided game logs nothing, cpp:121-123, so
// the executed steps are a prefix of the chunk), and count[q] = how many.  Thread <-> (file, step): consecutive threads
// read consecutive arenas.
__global__ void k_gather_runs(const double *rec, long long B, int R, int K, int n_files, const int *ids, double *runs, int *count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_files * K) return;
  const int q = (int)(i % n_files), k = (int)(i / n_files);
  const long long a = ids[q];
  const double *src = rec + (long long)k * R * B + a;
  if (src[(long long)(R - 1) * B] == 0.0) return;
  atomicAdd(&count[q], 1);
  double *dst = runs + ((long long)q * K + k) * (R - 1);
  for (int r = 0; r < R - 1; ++r) dst[r] = src[(long long)r * B];
}

// Selftest of the device generator: one wavefront produces the first n normals of the stream seeded with `seed`
// (n <= window) and, per attempt of the polar loop, the two canonical uniforms it drew.
__global__ __launch_bounds__(64) void k_selftest_rng(unsigned x0, int n, int n_attempts, double *normals, double *canonicals,
                                                     unsigned *x_out, int *have_out) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < n_attempts; i += 64) {               // the uniforms of attempt i, straight from the jump formula
    const unsigned s = lcg_mulmod(x0, lcg_pow(4u * (unsigned)i));
    const unsigned d1 = lcg_mulmod(s, kLcgA), d2 = lcg_mulmod(d1, kLcgA), d3 = lcg_mulmod(d2, kLcgA), d4 = lcg_mulmod(d3, kLcgA);
    canonicals[2 * i] = lcg_canonical(d1, d2);
    canonicals[2 * i + 1] = lcg_canonical(d3, d4);
  }
  DevRng rng;
  rng.logtab = lds + n + kRngSlack;
  DevRng::load_log_table(lds + n + kRngSlack, lane, 64);
  wave_sync();
  rng.buf = lds; rng.cap = n + kRngSlack; rng.x = x0; rng.have = 0; rng.lane_jump = DevRng::lcg_pow_lane(lane);
  rng.ensure(n, lane);
  for (int i = lane; i < n; i += 64) normals[i] = rng.buf[i];
  if (lane == 0) { *x_out = rng.x; *have_out = rng.have; }
}

__global__ void k_selftest_divtau(const double *a, long long n, double tau, double inv, double *fast, double *ieee,
                                  double *one) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fast[i] = div_tau<1>(a[i], tau, inv);
  ieee[i] = div_tau<0>(a[i], tau, inv);
  one[i] = div_tau<3>(a[i], tau, inv);
}

__global__ void k_selftest(const double *a, const double *b, long long n, double *q, double *s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  q[i] = div_ieee(a[i], b[i]);
  s[i] = sqrt_ieee(a[i]);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------

struct GuardedBuffer { const char *name; char *base; size_t bytes; };

struct rgs_sim {
  std::vector<GuardedBuffer> guards;   // every device buffer of this handle (debug guard zones, see dev_alloc)
  long long B = 0;
  int n = 2, P = 4, E = 5, S = 22;
  int device = 0;
  int variant = RGS_VARIANT_AUTO;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double *state = nullptr, *tcur = nullptr, *damage = nullptr, *ctl = nullptr, *tprev = nullptr;
  unsigned long long *counters = nullptr;
  uint8_t *flags = nullptr;
  // reset() parameters, src/GameSim.cpp:27-46
  double dt = 0.05, dt_col = 0.05 / kGrid, tau_puck = 0.1, tau_player = 0.5, m_player = 1.0, m_puck = 0.5;
  int win = 3, seed = 0;
  bool fast_div = true;            // cleared by rgs_set_fast_division(0) or by unsuitable taus
  bool sparse_watch = true;        // cleared by rgs_set_sparse_watch(0)
  bool stamps = false;             // rgs_get_stamps(enable)
  bool one_step_ok = false;        // both taus pass one_step_certified (set by reset)
  // AUTO variant: contact regime estimate (see launch_steps)
  bool auto_wave = false;          // currently preferring the wave-per-arena kernel
  long long window_steps = 0;      // steps launched since the last look at the counters
  unsigned long long seen_busy = 0;  // events resolved at that look
  bool one_step_enabled = true;    // rgs_set_fast_division(2) = allow, (1) = two steps only
  // noise > 0: the reference's generator per arena, src/GameSim.cpp:3,41 -- the engine is
  // re-seeded by reset(), the distribution (with its cached second draw) deliberately is not.
  double sigma = 0.0;
  std::vector<std::default_random_engine> reng;
  std::vector<std::normal_distribution<double>> wdist;
  std::vector<double> tape_host;
  std::vector<int> consumed_host;
  std::vector<long long> tapeoff_host;                         // per-arena tape start / length (noise_step)
  std::vector<int> tapen_host;
  std::vector<std::default_random_engine> snap_eng;            // generator state after an event-free step's draws
  std::vector<std::normal_distribution<double>> snap_dist;
  long long *tapeoff_dev = nullptr;
  int *tapen_dev = nullptr;
  double *tape_dev = nullptr, *bak_state = nullptr, *bak_tcur = nullptr, *bak_damage = nullptr;
  int *consumed_dev = nullptr;
  unsigned long long *bak_counters = nullptr;
  size_t tape_cap = 0;             // doubles allocated in tape_dev
  // native log writer (rgs_log_open / rgs_step_device_log / rgs_log_close)
  std::vector<int> log_fds;
  int *log_ids = nullptr, *log_count = nullptr;      // device: arena of file q, executed records of file q in the last chunk
  double *log_rec = nullptr, *log_runs = nullptr;    // device: (cap, R, B) records of a chunk; (n_files, cap, R-1) per-file runs
  double *log_runs_host = nullptr;                   // pinned mirror of log_runs (lent by the process-wide staging buffer when free)
  bool log_runs_borrowed = false;
  int *log_count_host = nullptr;
  int log_cap = 0;                                   // steps the chunk buffers hold
  // rgs_set_noise_mode: 0 = draws from the host's <random> (tapes), 1 = the same generator on the device; 1 is the default
  // where it is exact as well (built with the host's log, gsim_hostlog.hpp)
  int noise_mode = GSIM_HOST_LOG ? 1 : 0;
  unsigned *rng_x = nullptr;       // device generator: engine state, left-over count and left-over normals per arena
  int *rng_left_n = nullptr;
  double *rng_left = nullptr;
  double *vgrid = nullptr;         // split-grid velocity rows (two-lane hybrid, n >= 4)
  size_t vgrid_cap = 0;
  int watch_every = 0;             // pair-watch interval in sub-steps (a divisor of 50); 0 = choose from dt
  int n_cus = 256;                 // hipDeviceProp_t::multiProcessorCount
  // Pooled kernels report a bounded wait that ran out in counters[CNT_POOL_TIMEOUT]: arenas may then be short of
  // their steps.  The error is sticky: every synchronising call checks it (finish()) and fails until the next reset.
  // Dealing hot arenas over the pooled kernels' workgroups (rebalance()): perm[slot] = arena, heat[arena] = arena-steps
  // handed to the wave pass in the pooled launches since the last deal.
  int *perm = nullptr, *hot_count = nullptr, *hot_list = nullptr, *cold_list = nullptr, *n_swapped = nullptr;
  unsigned *heat = nullptr;
  bool perm_valid = false;         // perm holds a deal (else the kernels get a null perm = identity)
  bool deal_enabled = true;        // rgs_set_rebalance
  long long heat_steps = 0;        // pooled steps counted into heat since the last deal
  unsigned long long *wave_log = nullptr;   // diagnostic: per-wavefront records of the asynchronous pool (stamps on)
  bool pooled_in_flight = false;   // a pooled launch has been queued since the counters were last looked at
  bool poisoned = false;
  unsigned long long *counters_pin = nullptr;   // page-locked mirror of the CNT_N counters
};

namespace {

// ---- debug guard zones --------------------------------------------------------------------------
// GPU AddressSanitizer is not available on this pool.  With RGS_DEBUG_GUARD_BYTES=<n> in the environment every
// device buffer of a handle is allocated with n bytes of 0xA5 in front of and behind it; rgs_debug_check_guards()
// reports the first buffer whose guard zone was written, and out-of-bounds READS land in mapped memory
// instead of faulting, which tells an over-read from an over-write.  Off (n = 0) in normal use.
size_t guard_bytes() {
  // rounded up to a multiple of 256 so the guarded buffers keep hipMalloc's alignment (same coalescing as unguarded)
  static const size_t g = [] { const char *e = std::getenv("RGS_DEBUG_GUARD_BYTES"); const size_t v = e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0; return (v + 255) / 256 * 256; }();
  return g;
}

template <typename T>
hipError_t dev_alloc(rgs_sim *s, T **out, size_t bytes, const char *name) {
  const size_t g = guard_bytes();
  char *base = nullptr;
  hipError_t e = hipMalloc((void **)&base, bytes + 2 * g);
  if (e != hipSuccess) return e;
  if (g) {
    e = hipMemset(base, 0xA5, bytes + 2 * g);
    if (e == hipSuccess) e = hipDeviceSynchronize();          // the fill runs on the null stream; the handle's stream does not wait for it
    if (e != hipSuccess) { (void)hipFree(base); return e; }
  }
  s->guards.push_back({name, base, bytes});
  *out = (T *)(base + g);
  return hipSuccess;
}

template <typename T>
void dev_free(rgs_sim *s, T *&ptr) {
  if (!ptr) return;
  const size_t g = guard_bytes();
  char *base = (char *)ptr - g;
  for (size_t i = 0; i < s->guards.size(); ++i)
    if (s->guards[i].base == base) { s->guards.erase(s->guards.begin() + i); break; }
  (void)hipFree(base);
  ptr = nullptr;
}

const char *kPoolTimeoutText = "pooled stepping kernel: a bounded wait timed out (internal protocol error); the arenas' state is invalid until the next reset";

// Waits for the handle's stream.  If a pooled kernel has run since the last look, the counters come along (one 64-byte
// copy into page-locked memory) and a protocol timeout poisons the handle: this call and every later synchronising call
// fail until rgs_reset.
int finish(rgs_sim *s) {
  if (s->pooled_in_flight && s->counters_pin)
    HIP_TRY(hipMemcpyAsync(s->counters_pin, s->counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->pooled_in_flight && s->counters_pin) {
    s->pooled_in_flight = false;
    if (s->counters_pin[CNT_POOL_TIMEOUT] != 0) s->poisoned = true;
  }
  if (s->poisoned) return fail(RGS_ERR_HIP, kPoolTimeoutText);
  return RGS_OK;
}
#define RGS_TRY(expr) do { int rc_ = (expr); if (rc_ != RGS_OK) return rc_; } while (0)

// div_tau<true>'s preconditions on the divisor (gsim_common.hpp): normal, moderate exponent,
// significand not ending in a run of ones.
bool fast_division_ok(double tau) {
  if (!std::isfinite(tau) || tau == 0.0) return false;
  const double m = std::fabs(tau);
  if (m < 0x1.0p-100 || m > 0x1.0p100) return false;
  uint64_t bits;
  std::memcpy(&bits, &tau, sizeof bits);
  const uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
  return frac < 0xFFFFFFFFFFFFFull - 64;
}

// +-2^k with a moderate exponent: 1/tau is exact and a*(1/tau) rounds exactly like a/tau.
bool is_power_of_two(double tau) {
  if (!fast_division_ok(tau)) return false;
  int e;
  return std::fabs(std::frexp(tau, &e)) == 0.5;
}

// x * 2^j is exact: the product is zero or a normal number and scales back to x.
bool scales_exactly(double x, double pow2) {
  const double y = x * pow2;
  if (y == 0.0) return x == 0.0;
  return std::isnormal(y) && y / pow2 == x;
}

// ---- certification of div_tau<3> (one correction step) for a given tau --------------------------
// See gsim_common.hpp: the one-step quotient can only be wrong for dividends with significand A such
// that |A*2^sh - M*B| <= 7 with M odd (sh = 53 for A >= B, 54 for A < B; B = tau's 53-bit significand).
// Those A are the solutions of A*2^sh = r (mod B), |r| <= 7: enumerate them and test each one.
typedef unsigned __int128 u128;

uint64_t mod_inverse(uint64_t c, uint64_t m) {         // c^-1 mod m, gcd(c, m) = 1, m > 1
  __int128 t = 0, nt = 1, r = m, nr = c % m;
  while (nr != 0) {
    const __int128 q = r / nr;
    const __int128 tt = t - q * nt; t = nt; nt = tt;
    const __int128 rr = r - q * nr; r = nr; nr = rr;
  }
  if (t < 0) t += m;
  return (uint64_t)t;
}

bool one_step_quotient_ok(uint64_t A, double Bd, double y) {
  const double a = (double)A;                            // exact: A < 2^53
  const double q0 = a * y;
  const double r0 = std::fma(-Bd, q0, a);
  return std::fma(r0, y, q0) == a / Bd;
}

// Collects (up to cap) the worst-case significands into out (may be null) and returns whether all pass.
bool one_step_certified(double tau, std::vector<double> *out = nullptr) {
  if (!fast_division_ok(tau)) return false;
  int e;
  const double m = std::frexp(std::fabs(tau), &e);
  const uint64_t B = (uint64_t)std::ldexp(m, 53);        // [2^52, 2^53)
  if (B == (1ull << 52)) return true;                    // power of two: r0 = 0, q1 = q0 exact
  const double Bd = (double)B, y = 1.0 / Bd;
  const int v2 = __builtin_ctzll(B);
  if (v2 > 12) return false;                             // too many solutions to enumerate; use two steps
  const uint64_t g = 1ull << v2, Bp = B >> v2;           // Bp odd
  bool ok = true;
  for (int sh = 53; sh <= 54; ++sh) {
    const uint64_t c = (uint64_t)((((u128)1) << (sh - v2)) % Bp);
    const uint64_t cinv = mod_inverse(c, Bp);
    const uint64_t lo = sh == 53 ? B : (1ull << 52), hi = sh == 53 ? (1ull << 53) : B;
    for (int r = -7; r <= 7; ++r) {
      if (r == 0 || (r % (int)g) != 0) continue;
      const long long rr = r / (long long)g;
      uint64_t rmod = (uint64_t)(rr >= 0 ? rr : -rr) % Bp;
      if (rr < 0 && rmod != 0) rmod = Bp - rmod;
      uint64_t A = (uint64_t)(((u128)rmod * cinv) % Bp);
      if (A < lo) A += ((lo - A + Bp - 1) / Bp) * Bp;
      for (; A < hi; A += Bp) {
        u128 num = ((u128)A) << sh;
        num = r > 0 ? num - (u128)r : num + (u128)(-r);
        if (num % B != 0) continue;                      // cannot happen; defensive
        if (((num / B) & 1) == 0) continue;              // M must be odd to be a midpoint
        if (out && out->size() < 65536) out->push_back(std::ldexp((double)A, e - 53 + (sh == 53 ? 0 : 0)));
        if (!one_step_quotient_ok(A, Bd, y)) ok = false;
      }
    }
  }
  return ok;
}

bool use_fast_division(const rgs_sim *s) {
  return s->fast_div && fast_division_ok(s->tau_player) && fast_division_ok(s->tau_puck);
}

// Largest double s with fl(sqrt(s)) <= R (strict = false) or < R (strict = true); host sqrt is IEEE.
double sqrt_threshold(double R, bool strict) {
  auto ok = [&](double s) { const double r = std::sqrt(s); return strict ? r < R : r <= R; };
  double s = R * R;
  while (!ok(s)) s = std::nextafter(s, 0.0);
  while (ok(std::nextafter(s, INFINITY))) s = std::nextafter(s, INFINITY);
  return s;
}

const GsimOps *gsim_ops(int n) {
  switch (n) {
    case 1: return gsim_ops_n1();
    case 2: return gsim_ops_n2();
    case 3: return gsim_ops_n3();
    case 4: return gsim_ops_n4();
    case 5: return gsim_ops_n5();
    case 6: return gsim_ops_n6();
    case 7: return gsim_ops_n7();
  }
  return nullptr;
}

int launch_kind(rgs_sim *s, const StepArgs &a, int kind, int mode) {
  const hipError_t e = gsim_ops(s->n)->step(a, s->stream, kind, mode);
  if (e != hipSuccess) return fail(RGS_ERR_HIP, std::string("stepping kernel launch: ") + hipGetErrorString(e));
  return RGS_OK;
}

constexpr long long kNoiseMaxArenas = 4096;     // tapes are host-generated and uploaded every step

// fn(lo, hi) over [0, n) on up to 16 host threads (the per-arena generators are independent).
template <typename F>
void parallel_arenas(long long n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const long long nt = std::max<long long>(1, std::min<long long>(std::min<long long>(16, hw ? hw : 1), n / 64));
  if (nt <= 1) { fn(0, n); return; }
  std::vector<std::thread> pool;
  for (long long t = 0; t < nt; ++t) pool.emplace_back([=] { fn(n * t / nt, n * (t + 1) / nt); });
  for (auto &th : pool) th.join();
}

// One step with w_stdev_ != 0.  The host draws a tape of N(0,1) values per arena from COPIES of the arena's
// generator, the wave kernel consumes them in the reference's call order and reports how many it used, and the
// real generators are then set to the state after exactly that many draws.  The first tape holds the draws of
// an event-free step plus one event; an arena whose step needs more gets a longer tape (6 events, then the
// 49-event worst case) appended to the buffer and the step is redone from a backup -- for everybody on the
// device (cheap), but only the overflowing arenas' tapes are drawn again (the host's draws are what this path
// costs).  The generator state after the event-free count (what nearly every arena-step consumes) is kept from
// the tape's own generation, so the common case draws each value once.
int noise_step(rgs_sim *s, StepArgs a) {
  const long long B = s->B;
  const int P = s->P;
  const int base_draws = 8 * kGrid * P;
  const int per_event = 8 * 2 * (kGrid + 3);
  const int lens[3] = {base_draws + per_event, base_draws + 6 * per_event, base_draws + (kLoopCap - 1) * per_event};
  HIP_TRY(hipMemcpyAsync(s->bak_state, s->state, sizeof(double) * s->S * B, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->bak_tcur, s->tcur, sizeof(double) * B, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->bak_damage, s->damage, sizeof(double) * P * P * B, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->bak_counters, s->counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToDevice, s->stream));
  s->tapeoff_host.resize(B); s->tapen_host.resize(B);
  s->snap_eng.resize(B); s->snap_dist.resize(B, std::normal_distribution<double>(0.0, 1.0));
  s->consumed_host.resize(B);
  std::vector<long long> todo(B);                              // arenas whose tape is (re)drawn in this attempt
  for (long long ar = 0; ar < B; ++ar) todo[ar] = ar;
  size_t used = 0;                                             // doubles of the buffer in use
  for (int attempt = 0; attempt < 3; ++attempt) {
    const int T = lens[attempt];
    const size_t first = used;
    used += (size_t)T * todo.size();
    s->tape_host.resize(used);
    parallel_arenas((long long)todo.size(), [&](long long lo, long long hi) {
      for (long long q = lo; q < hi; ++q) {
        const long long ar = todo[q];
        std::default_random_engine e = s->reng[ar];            // copies: the originals are set at the end
        std::normal_distribution<double> d = s->wdist[ar];
        double *t = s->tape_host.data() + first + (size_t)q * T;
        for (int k = 0; k < base_draws; ++k) t[k] = d(e);      // w_dist_(reng_), src/GameSim.cpp:66-67
        if (attempt == 0) { s->snap_eng[ar] = e; s->snap_dist[ar] = d; }   // the state after an event-free step
        for (int k = base_draws; k < T; ++k) t[k] = d(e);
        s->tapeoff_host[ar] = (long long)(first + (size_t)q * T);
        s->tapen_host[ar] = T;
      }
    });
    if (used > s->tape_cap) {                                  // grow: everything drawn so far is re-sent (rare)
      HIP_TRY(hipStreamSynchronize(s->stream));
      dev_free(s, s->tape_dev);
      s->tape_dev = nullptr; s->tape_cap = 0;
      HIP_TRY(dev_alloc(s, &s->tape_dev, sizeof(double) * used, "tape_dev"));
      s->tape_cap = used;
      HIP_TRY(hipMemcpyAsync(s->tape_dev, s->tape_host.data(), sizeof(double) * used, hipMemcpyHostToDevice, s->stream));
    } else {
      HIP_TRY(hipMemcpyAsync(s->tape_dev + first, s->tape_host.data() + first, sizeof(double) * (used - first),
                             hipMemcpyHostToDevice, s->stream));
    }
    HIP_TRY(hipMemcpyAsync(s->tapeoff_dev, s->tapeoff_host.data(), sizeof(long long) * B, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->tapen_dev, s->tapen_host.data(), sizeof(int) * B, hipMemcpyHostToDevice, s->stream));
    a.tape = s->tape_dev; a.tape_off = s->tapeoff_dev; a.tape_n = s->tapen_dev; a.sigma = s->sigma; a.consumed = s->consumed_dev;
    a.n_steps = 1; a.ctl_stride = 0;
    HIP_TRY(hipMemsetAsync(s->consumed_dev, 0, sizeof(int) * B, s->stream));   // decided arenas draw nothing
    int rc = launch_kind(s, a, LAUNCH_WAVE_NOISE, 0);
    if (rc != RGS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s->consumed_host.data(), s->consumed_dev, sizeof(int) * B, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    todo.clear();
    for (long long ar = 0; ar < B; ++ar)
      if (s->consumed_host[ar] > s->tapen_host[ar]) todo.push_back(ar);
    if (todo.empty()) {
      parallel_arenas(B, [&](long long lo, long long hi) {
        for (long long ar = lo; ar < hi; ++ar) {
          const int c = s->consumed_host[ar];
          if (c == base_draws) { s->reng[ar] = s->snap_eng[ar]; s->wdist[ar] = s->snap_dist[ar]; }
          else for (int k = 0; k < c; ++k) (void)s->wdist[ar](s->reng[ar]);
        }
      });
      return RGS_OK;
    }
    if (attempt == 2) return fail(RGS_ERR_HIP, "noise_step: worst-case tape still too short (internal error)");
    HIP_TRY(hipMemcpyAsync(s->state, s->bak_state, sizeof(double) * s->S * B, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->tcur, s->bak_tcur, sizeof(double) * B, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->damage, s->bak_damage, sizeof(double) * P * P * B, hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->counters, s->bak_counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToDevice, s->stream));
  }
  return RGS_OK;
}

// Pooled kernels bind 256 arenas to a workgroup (= a CU) for a whole launch, and an arena in lasting contact (a puck
// resting against a player, a disc pressed into a wall) needs the wave pass at EVERY step for tens to hundreds of steps:
// where such arenas cluster, their CU finishes last (driver window of bench.py: 12 to 36 of them per workgroup, the
// slowest wavefront 1.27x the average).  Between launches the arenas are therefore dealt afresh: an arena that went
// to the wave pass in at least 7/8 of the steps since the last deal is "hot"; workgroups above the average number
// of hot arenas swap their surplus against cold arenas of workgroups below it.  Only the swapped lanes (about 1 % of
// them) leave the coalesced arena order -- a full re-ordering by heat was measured in round 1 and lost 27 % to the
// scattered state accesses.  An arena's data never moves and its arithmetic does not depend on who runs it: results are
// bit-identical with and without (tests/test_gpu_parity.py::test_hot_arenas_are_dealt_evenly...).
int rebalance(rgs_sim *s, int n_steps) {
  const int n_wg = (int)(s->B / kDealWG);                     // full workgroups only; a ragged last one keeps its arenas
  if (!s->deal_enabled || n_wg < 2) return RGS_OK;
  if (!s->perm) {
    const size_t slots = (size_t)(n_wg + 1) * kDealWG;
    HIP_TRY(dev_alloc(s, &s->perm, sizeof(int) * slots, "perm"));
    HIP_TRY(dev_alloc(s, &s->heat, sizeof(unsigned) * slots, "heat"));
    HIP_TRY(dev_alloc(s, &s->hot_count, sizeof(int) * n_wg, "hot_count"));
    HIP_TRY(dev_alloc(s, &s->hot_list, sizeof(int) * (size_t)n_wg * kDealMax, "hot_list"));
    HIP_TRY(dev_alloc(s, &s->cold_list, sizeof(int) * (size_t)n_wg * kDealMax, "cold_list"));
    HIP_TRY(dev_alloc(s, &s->n_swapped, sizeof(int), "n_swapped"));
    HIP_TRY(hipMemsetAsync(s->heat, 0, sizeof(unsigned) * slots, s->stream));
    HIP_TRY(hipMemsetAsync(s->n_swapped, 0, sizeof(int), s->stream));
    s->heat_steps = 0; s->perm_valid = false;
  }
  // a new deal needs a few steps of history and is only worth its two small launches ahead of a fused launch
  if (s->heat_steps >= 4 && n_steps >= 8) {
    // "hot" = in the event pass at (nearly) every step of the window: measured on the driver's shape (tools/heat_probe.py),
    // arenas flagged at 5 of 5 warm-up steps carry 89 % of the next steps' event-pass work (5.6 of 7 steps each), those
    // flagged at 3 or 4 of 5 carry 0.4 of 7 -- counting them as hot (an earlier ">= half" rule) dealt the wrong arenas
    const unsigned thr = (unsigned)(s->heat_steps - s->heat_steps / 8);
    hipLaunchKernelGGL(k_deal_scan, dim3((unsigned)n_wg), dim3(kDealWG), 0, s->stream, s->heat, s->heat, s->perm, thr, s->hot_count,
                       s->hot_list, s->cold_list);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_deal_match, dim3(1), dim3(1024), sizeof(int) * 4 * (n_wg + 1), s->stream, s->hot_count, s->hot_list,
                       s->cold_list, n_wg, (n_wg + 1023) / 1024, s->perm, s->heat + (size_t)n_wg * kDealWG, s->n_swapped);
    HIP_TRY(hipGetLastError());
    s->heat_steps = 0;
    s->perm_valid = true;
  }
  return RGS_OK;
}

int launch_steps(rgs_sim *s, const double *ctl_dev, long long stride, int n_steps, double *traj = nullptr,
                 int traj_every = 0, double *rec = nullptr) {
  StepArgs a{};
  a.dt = s->dt; a.dt_col = s->dt_col;
  a.tau_player = s->tau_player; a.tau_puck = s->tau_puck;
  a.inv_tau_player = 1.0 / s->tau_player; a.inv_tau_puck = 1.0 / s->tau_puck;   // host IEEE division
  a.m_player = s->m_player; a.m_puck = s->m_puck;
  // the radii are compile-time constants: search the four thresholds once per process, not per launch
  static const double le_pp = sqrt_threshold(kRPlayer + kRPlayer, false), le_pk = sqrt_threshold(kRPlayer + kRPuck, false);
  static const double lt_pp = sqrt_threshold(kRPlayer + kRPlayer, true), lt_pk = sqrt_threshold(kRPlayer + kRPuck, true);
  a.le_pp = le_pp; a.le_pk = le_pk; a.lt_pp = lt_pp; a.lt_pk = lt_pk;
  {
    const double rp = kRPlayer, rk = kRPuck;
    a.push101_pp = (1.01 * rp) / (rp + rp); a.push101_plk = (1.01 * rp) / (rp + rk); a.push101_pkk = (1.01 * rk) / (rp + rk);
    a.push11_pp = (1.1 * rp) / (rp + rp);   a.push11_plk = (1.1 * rp) / (rp + rk);   a.push11_pkk = (1.1 * rk) / (rp + rk);
    a.ratio_pp = (2 * s->m_player) / (s->m_player + s->m_player);
    a.ratio_vs_puck = (2 * s->m_puck) / (s->m_player + s->m_puck);
    a.ratio_vs_player = (2 * s->m_player) / (s->m_player + s->m_puck);
  }
  a.win = (double)s->win;
  a.B = s->B;
  a.state = s->state; a.tcur = s->tcur; a.damage = s->damage;
  a.ctl = ctl_dev; a.ctl_stride = stride; a.n_steps = n_steps;
  {  // the sparse watch's travel bound needs the velocities to stay inside [v0, u]: small dt_col/tau
    const double zp = s->dt_col / s->tau_player, zk = 0.2 * s->dt_col / s->tau_puck;
    const bool ok = s->sparse_watch && zp > 0.0 && zp <= 0.1 && zk > 0.0 && zk <= 0.1;
    // Interval: the pair radius is widened by ~3.5*(we-1)*dt_col*vb, and every arena inside the widened
    // radius costs a wave pass.  Measured (profiles/r01): dt = 0.01 is fastest at 5, dt = 0.05 at 2, i.e.
    // the largest of {5, 2, 1} whose skipped time (we-1)*dt_col stays within 1 ms; rgs_set_sparse_watch
    // can pin it.
    int we = 1;
    if (ok) {
      if (s->watch_every > 0) we = s->watch_every;
      else we = 4.0 * s->dt_col <= 1.0e-3 ? 5 : (s->dt_col <= 1.0e-3 ? 2 : 1);
    }
    a.watch_every = we;
    a.wall_ends = ok ? 1 : 0;
  }
  a.counters = s->counters;
  a.n_state = (long long)s->S * s->B; a.n_tcur = s->B; a.n_damage = (long long)s->P * s->P * s->B;
  a.stamps = s->stamps ? 1 : 0;
  a.stamp_out = s->counters + CNT_N;
  a.sigma = 0.0; a.tape = nullptr; a.tape_off = nullptr; a.tape_n = nullptr; a.consumed = nullptr; a.vgrid = nullptr;
  a.rng_x = nullptr; a.rng_left_n = nullptr; a.rng_left = nullptr;
  a.perm = nullptr; a.heat = nullptr; a.wave_log = s->stamps ? s->wave_log : nullptr;
  a.traj = traj; a.traj_every = traj_every > 0 ? traj_every : 1;
  a.rec = rec; a.tprev = s->tprev;
  if (s->poisoned) return fail(RGS_ERR_HIP, kPoolTimeoutText);
  if (s->sigma != 0.0 && s->noise_mode == 1) {                 // noise from the device generator: fused launch, any batch size
    a.sigma = s->sigma;
    a.rng_x = s->rng_x; a.rng_left_n = s->rng_left_n; a.rng_left = s->rng_left;
    // division by the taus as in the noise-free wave kernel (the reciprocal forms equal '/' inside their magnitude window;
    // rgs_set_fast_division(0) forces IEEE)
    // (with the host's log the draws are the reference's bit for bit, gsim_hostlog.hpp, so the division has to be exact as
    // well: the reciprocal forms only inside their window, which for a noisy player needs a sigma of ordinary size)
    const bool sigma_ok = std::fabs(s->sigma) >= 0x1p-300 && std::fabs(s->sigma) <= 0x1p300;
    return launch_kind(s, a, LAUNCH_WAVE_RNG, !(use_fast_division(s) && sigma_ok) ? 0 : (s->one_step_ok ? 3 : 1));
  }
  if (s->sigma != 0.0) {                                       // noise: tape-driven wave kernel, one step per launch
    for (int k = 0; k < n_steps; ++k) {
      a.ctl = ctl_dev + (long long)k * stride;
      const bool snap = traj != nullptr && (k + 1) % a.traj_every == 0;   // one step per launch here
      StepArgs b = a;
      b.traj = snap ? traj + (long long)((k + 1) / a.traj_every - 1) * s->S * s->B : nullptr;
      b.traj_every = 1;
      b.rec = rec ? rec + (long long)k * rgs_record_size(s->n) * s->B : nullptr;   // one step per launch: record index 0, tprev kept current
      int rc = noise_step(s, b);
      if (rc != RGS_OK) return rc;
      if (rec) HIP_TRY(hipMemcpyAsync(s->tprev, s->tcur, sizeof(double) * s->B, hipMemcpyDeviceToDevice, s->stream));
    }
    return RGS_OK;
  }
  // Both hybrids assume that events are the exception.  They stay ahead of the plain wave-per-arena
  // kernel far into contact-rich play (2v2, random commands up to 100: 74 % of the arena-steps have an
  // event and the one-lane hybrid still does 8.5e7 against 4.7e7; 4v4 two-lane: level at 79 %), but not
  // when every step resolves a chain of events (everybody jammed together, 3-13 events per arena-step, 49-event
  // cap hits: wave kernel 4.4e7, hybrids 4.0e7 / 3.7e7 averaged over the build-up and further apart later):
  // there the wave kernel's four waves per SIMD hide the serial event chains.  So every >= 64 launched
  // steps AUTO looks at the events resolved per arena-step in the finished launches and switches with
  // hysteresis (> 3 -> wave kernel, < 2 -> hybrid).  Results are identical either way; only this
  // host-side choice depends on the history.
  if (s->variant == RGS_VARIANT_AUTO && s->window_steps >= 64) {
    unsigned long long c[CNT_N];
    HIP_TRY(hipMemcpyAsync(c, s->counters, sizeof(c), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (c[CNT_POOL_TIMEOUT] != 0) { s->poisoned = true; return fail(RGS_ERR_HIP, kPoolTimeoutText); }
    const unsigned long long resolved = c[CNT_RESOLVED];     // counted alike by every kernel
    const double per_step = (double)(resolved - s->seen_busy) / ((double)s->window_steps * (double)s->B);
    if (!s->auto_wave && per_step > 3.0) s->auto_wave = true;
    else if (s->auto_wave && per_step < 2.0) s->auto_wave = false;
    s->seen_busy = resolved;
    s->window_steps = 0;
  }
  if (s->variant == RGS_VARIANT_AUTO) s->window_steps += n_steps;
  // Which kernel: estimated arena-steps/s (x1e6) from the rates in auto_rates.h, a header GENERATED from a sweep on
  // MI355X (tools/sweep.py -> profiles/r02/sweep_auto.json -> tools/rates_to_header.py); nothing measured is typed in
  // here.  The hybrids are measured at 32 768, 65 536 and 131 072 arenas (linear in between and below the first, flat
  // above the last: the one-lane kernels need 65 536 arenas to put a wavefront on every SIMD, the two-lane one 32 768,
  // and beyond that the workgroups run in successive waves over the CUs); the wave-per-arena kernel spreads even a few
  // hundred arenas over the whole chip and is modelled as B / (t0 + B / saturated rate) through its 1 024- and
  // 131 072-arena points.  Launches of few steps are tail-dominated and have their own table ("short"); the
  // share of arena-steps with events grows with the step length, so the table is measured at dt = 0.01 and 0.05 and the
  // nearer one (in ratio) is used.
  // The pooled hybrids' workgroups (256 arenas, ~140 KB of LDS: one per CU) are self-contained; with more of them than
  // CUs the later ones start as the earlier ones end, like the blocks of the other kernels.
  const int regime = n_steps <= 64 ? 0 : 1;
  int dtc = 0;                                               // the measured step length nearest to this handle's (in ratio)
  for (int i = 1; i < kAutoDts; ++i)
    if (std::fabs(std::log(s->dt / kAutoDt[i])) < std::fabs(std::log(s->dt / kAutoDt[dtc]))) dtc = i;
  const double Bd = (double)s->B;
  auto hybrid_rate = [&](int kern) {
    const double r32 = kAutoRate[dtc][regime][kern][0][s->n], r65 = kAutoRate[dtc][regime][kern][1][s->n];
    const double r131 = kAutoRate[dtc][regime][kern][2][s->n];
    if (Bd <= 32768.0) return r32 * Bd / 32768.0;
    if (Bd >= 131072.0) return r131;
    if (Bd >= 65536.0) return r65 + (r131 - r65) * (Bd - 65536.0) / 65536.0;
    return r32 + (r65 - r32) * (Bd - 32768.0) / 32768.0;
  };
  const double est_one = hybrid_rate(1), est_two = hybrid_rate(2);
  const double est_pool = hybrid_rate(3), est_apool = hybrid_rate(4);
  double est_wave = 0.0;
  {
    const double sat = kAutoRate[dtc][regime][0][2][s->n], r1k = kAutoWave1k[dtc][regime][s->n];
    if (sat > 0.0 && r1k > 0.0) {
      const double t0 = 1024.0 / r1k - 1024.0 / sat;           // microseconds
      est_wave = Bd / ((t0 > 0.0 ? t0 : 0.0) + Bd / sat);
    }
  }
  int best = RGS_VARIANT_HYBRID;
  double best_rate = est_one;
  if (est_two > best_rate) { best = RGS_VARIANT_HYBRID2; best_rate = est_two; }
  if (est_pool > best_rate) { best = RGS_VARIANT_POOL; best_rate = est_pool; }
  if (est_apool > best_rate) { best = RGS_VARIANT_APOOL; best_rate = est_apool; }
  if (est_wave > best_rate) { best = RGS_VARIANT_WAVE; best_rate = est_wave; }
  const int variant = s->variant != RGS_VARIANT_AUTO ? s->variant : (s->auto_wave ? RGS_VARIANT_WAVE : best);
  const bool fast = use_fast_division(s);
  // The power-of-two modes fold 1/tau_player into the lane pass' step constants (gsim_lane.hpp): those
  // products must be exact and normal (or zero).
  const bool fold_ok = is_power_of_two(s->tau_player) && scales_exactly(s->dt_col / 2.0, 1.0 / s->tau_player) &&
                       scales_exactly(s->dt_col, 1.0 / s->tau_player) && scales_exactly(s->dt_col / 6.0, 1.0 / s->tau_player);
  const int div_mode = !fast ? DIV_IEEE
                       : (fold_ok ? (s->one_step_ok ? DIV_POW2_ONESTEP : DIV_PLAYER_POW2) : DIV_MARKSTEIN);
  if (rec != nullptr)                                          // log records: the one-lane hybrid instantiated with the record code
    return launch_kind(s, a, LAUNCH_HYBRID_REC, div_mode);     // (any variant gives the same values; the others do not carry that code)
  if (variant == RGS_VARIANT_HYBRID2) {
    const size_t need = gsim_ops(s->n)->vgrid_doubles(s->B);   // velocity rows of every block's grids
    if (need > s->vgrid_cap) {
      HIP_TRY(hipStreamSynchronize(s->stream));
      dev_free(s, s->vgrid);
      s->vgrid = nullptr; s->vgrid_cap = 0;
      HIP_TRY(dev_alloc(s, &s->vgrid, sizeof(double) * need, "vgrid"));
      s->vgrid_cap = need;
    }
    a.vgrid = s->vgrid;
    return launch_kind(s, a, LAUNCH_HYBRID2, div_mode);
  }
  if ((variant == RGS_VARIANT_POOL || variant == RGS_VARIANT_APOOL) && n_steps < (1 << 26)) {
    RGS_TRY(rebalance(s, n_steps));
    a.perm = s->perm_valid ? s->perm : nullptr;
    a.heat = s->heat;                                          // null while dealing is off or the batch is a single workgroup
    if (s->heat) s->heat_steps += n_steps;
    s->pooled_in_flight = true;
    return launch_kind(s, a, variant == RGS_VARIANT_POOL ? LAUNCH_POOL : LAUNCH_APOOL, div_mode);
  }
  if (variant != RGS_VARIANT_WAVE) return launch_kind(s, a, LAUNCH_HYBRID, div_mode);
  return launch_kind(s, a, LAUNCH_WAVE, !fast ? 0 : (s->one_step_ok ? 3 : 1));

}

int do_reset(rgs_sim *s, const double *x0_host, int batched) {
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = batched ? (size_t)4 * s->B : 4;
  double *x0_dev = nullptr;
  HIP_TRY(hipMalloc(&x0_dev, n * sizeof(double)));
  hipError_t e = hipMemcpyAsync(x0_dev, x0_host, n * sizeof(double), hipMemcpyHostToDevice, s->stream);
  if (e == hipSuccess) e = gsim_ops(s->n)->reset(s->state, s->tcur, s->damage, x0_dev, batched, s->B, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(s->counters, 0, (CNT_N + STAMP_N) * sizeof(unsigned long long), s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  hipFree(x0_dev);
  if (e != hipSuccess) return fail(RGS_ERR_HIP, std::string("reset: ") + hipGetErrorString(e));
  return RGS_OK;
}

}  // namespace

extern "C" {

const char *rgs_last_error(void) { return g_err.c_str(); }

int rgs_state_size(int n_per_team) { return 2 + 4 * (2 * n_per_team + 1); }

int rgs_create(int64_t n_arenas, int n_per_team, int device_id, rgs_sim **out) {
  if (!out || n_arenas < 1 || n_per_team < 1 || n_per_team > 7) return fail(RGS_ERR_ARG, "rgs_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(RGS_ERR_NO_DEVICE, "no HIP device visible");
  if (device_id < 0 || device_id >= ndev) return fail(RGS_ERR_ARG, "rgs_create: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RGS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  rgs_sim *s = new rgs_sim();
  s->n_cus = prop.multiProcessorCount;
  s->B = n_arenas;
  s->n = n_per_team;
  s->P = 2 * n_per_team;
  s->E = s->P + 1;
  s->S = 2 + 4 * s->E;
  s->device = device_id;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&s->ev0);
  if (e == hipSuccess) e = hipEventCreate(&s->ev1);
  if (e == hipSuccess) e = dev_alloc(s, &s->state, sizeof(double) * s->S * s->B, "state");
  if (e == hipSuccess) e = dev_alloc(s, &s->tcur, sizeof(double) * s->B, "tcur");
  if (e == hipSuccess) e = dev_alloc(s, &s->damage, sizeof(double) * s->P * s->P * s->B, "damage");
  if (e == hipSuccess) e = dev_alloc(s, &s->tprev, sizeof(double) * s->B, "tprev");
  if (e == hipSuccess) e = dev_alloc(s, &s->ctl, sizeof(double) * s->P * 2 * s->B, "ctl");
  if (e == hipSuccess) e = dev_alloc(s, &s->counters, sizeof(unsigned long long) * (CNT_N + STAMP_N), "counters");
  if (e == hipSuccess) e = dev_alloc(s, &s->flags, (size_t)s->B, "flags");
  if (e == hipSuccess) e = hipHostMalloc((void **)&s->counters_pin, sizeof(unsigned long long) * CNT_N, hipHostMallocDefault);
  if (e != hipSuccess) {
    std::string m = std::string("rgs_create: ") + hipGetErrorString(e);
    rgs_destroy(s);
    return fail(RGS_ERR_HIP, m);
  }
  const double zero[4] = {0, 0, 0, 0};
  s->one_step_ok = one_step_certified(s->tau_player) && one_step_certified(s->tau_puck);
  int rc = do_reset(s, zero, 0);                 // C++ defaults, src/GameSim.cpp:21-25
  if (rc == RGS_OK && s->B / kDealWG >= 2) {     // dealing buffers now, and one empty deal so that the first real one finds
    rc = rebalance(s, 0);                        // its two kernels' code object loaded (0.3 ms otherwise, inside somebody's timed launch)
    if (rc == RGS_OK) {
      const int n_wg = (int)(s->B / kDealWG);
      hipLaunchKernelGGL(k_deal_scan, dim3((unsigned)n_wg), dim3(kDealWG), 0, s->stream, s->heat, s->heat, s->perm, 0xFFFFFFFFu,
                         s->hot_count, s->hot_list, s->cold_list);
      hipLaunchKernelGGL(k_deal_match, dim3(1), dim3(1024), sizeof(int) * 4 * (n_wg + 1), s->stream, s->hot_count, s->hot_list,
                         s->cold_list, n_wg, (n_wg + 1023) / 1024, s->perm, s->heat + (size_t)n_wg * kDealWG, s->n_swapped);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(RGS_ERR_HIP, "rgs_create: dealing kernels");
    }
  }
  if (rc != RGS_OK) { rgs_destroy(s); return rc; }
  *out = s;
  return RGS_OK;
}

int rgs_destroy(rgs_sim *s) {
  if (!s) return RGS_OK;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  dev_free(s, s->state); dev_free(s, s->tcur); dev_free(s, s->tprev);
  dev_free(s, s->rng_x); dev_free(s, s->rng_left_n); dev_free(s, s->rng_left);
  dev_free(s, s->wave_log); dev_free(s, s->perm); dev_free(s, s->heat); dev_free(s, s->hot_count); dev_free(s, s->hot_list); dev_free(s, s->cold_list); dev_free(s, s->n_swapped);
  dev_free(s, s->vgrid); dev_free(s, s->tape_dev); dev_free(s, s->bak_state); dev_free(s, s->bak_tcur); dev_free(s, s->bak_damage); dev_free(s, s->consumed_dev); dev_free(s, s->tapeoff_dev); dev_free(s, s->tapen_dev); dev_free(s, s->bak_counters); dev_free(s, s->damage); dev_free(s, s->ctl); dev_free(s, s->counters); dev_free(s, s->flags);
  (void)rgs_log_close(s);
  if (s->counters_pin) hipHostFree(s->counters_pin);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
  return RGS_OK;
}

int rgs_reset(rgs_sim *s, double dt, int winning_score, const double *x0_ball_host, int x0_batched, double noise,
              int seed, double tau_puck, double tau_player, double player_mass, double puck_mass) {
  if (!s || !x0_ball_host) return fail(RGS_ERR_ARG, "rgs_reset: null argument");
  if (noise != 0.0 && s->noise_mode == 1) {                    // device generator: engine of arena a seeded with seed + a (cpp:41)
    HIP_TRY(hipSetDevice(s->device));
    if (!s->rng_x) {
      HIP_TRY(dev_alloc(s, &s->rng_x, sizeof(unsigned) * s->B, "rng_x"));
      HIP_TRY(dev_alloc(s, &s->rng_left_n, sizeof(int) * s->B, "rng_left_n"));
      HIP_TRY(dev_alloc(s, &s->rng_left, sizeof(double) * (size_t)kRngLeft * s->B, "rng_left"));
    }
    std::vector<unsigned> x0(s->B);
    for (long long ar = 0; ar < s->B; ++ar) x0[ar] = lcg_seed((long long)(seed + (int)ar));
    HIP_TRY(hipMemcpyAsync(s->rng_x, x0.data(), sizeof(unsigned) * s->B, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->rng_left_n, 0, sizeof(int) * s->B, s->stream));   // a re-seeded engine starts a new stream
    HIP_TRY(hipStreamSynchronize(s->stream));                  // x0 goes out of scope
  } else if (noise != 0.0) {
    if (s->B > kNoiseMaxArenas)
      return fail(RGS_ERR_UNSUPPORTED, "rgs_reset: noise != 0 with host-generated draw tapes (rgs_set_noise_mode 0) is limited to 4096 arenas per handle; the device generator (mode 1) has no limit");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->bak_state) {
      HIP_TRY(dev_alloc(s, &s->bak_state, sizeof(double) * s->S * s->B, "bak_state"));
      HIP_TRY(dev_alloc(s, &s->bak_tcur, sizeof(double) * s->B, "bak_tcur"));
      HIP_TRY(dev_alloc(s, &s->bak_damage, sizeof(double) * s->P * s->P * s->B, "bak_damage"));
      HIP_TRY(dev_alloc(s, &s->consumed_dev, sizeof(int) * s->B, "consumed_dev"));
      HIP_TRY(dev_alloc(s, &s->tapeoff_dev, sizeof(long long) * s->B, "tapeoff_dev"));
      HIP_TRY(dev_alloc(s, &s->tapen_dev, sizeof(int) * s->B, "tapen_dev"));
      HIP_TRY(dev_alloc(s, &s->bak_counters, sizeof(unsigned long long) * CNT_N, "bak_counters"));
    }
    s->reng.resize(s->B);
    s->wdist.resize(s->B, std::normal_distribution<double>(0.0, 1.0));   // kept across resets, cpp:41
    for (long long ar = 0; ar < s->B; ++ar) s->reng[ar] = std::default_random_engine(seed + (int)ar);   // cpp:41
  }
  s->sigma = noise;
  s->poisoned = false; s->pooled_in_flight = false;
  s->perm_valid = false; s->heat_steps = 0;                   // a reset game has no contact history: identity deal
  if (s->heat) HIP_TRY(hipMemsetAsync(s->heat, 0, sizeof(unsigned) * (size_t)(s->B / kDealWG + 1) * kDealWG, s->stream));
  s->auto_wave = false; s->window_steps = 0; s->seen_busy = 0;
  s->one_step_ok = s->one_step_enabled && one_step_certified(tau_player) && one_step_certified(tau_puck);
  s->dt = dt;
  s->dt_col = dt / kGrid;                        // cpp:28
  s->win = winning_score;
  s->seed = seed;
  s->tau_puck = tau_puck; s->tau_player = tau_player;
  s->m_player = player_mass; s->m_puck = puck_mass;
  return do_reset(s, x0_ball_host, x0_batched);
}

int rgs_step(rgs_sim *s, const double *controls_host) {
  if (!s || !controls_host) return fail(RGS_ERR_ARG, "rgs_step: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(s->ctl, controls_host, sizeof(double) * s->P * 2 * s->B, hipMemcpyHostToDevice, s->stream));
  int rc = launch_steps(s, s->ctl, 0, 1);
  if (rc != RGS_OK) return rc;
  RGS_TRY(finish(s));      // controls_host may be reused by the caller
  return RGS_OK;
}

int rgs_step_device(rgs_sim *s, const double *controls_dev, int64_t step_stride, int n_steps) {
  if (!s || !controls_dev || n_steps < 0 || step_stride < 0) return fail(RGS_ERR_ARG, "rgs_step_device: bad argument");
  if (n_steps == 0) return RGS_OK;
  HIP_TRY(hipSetDevice(s->device));
  return launch_steps(s, controls_dev, step_stride, n_steps);
}

int rgs_step_device_traj(rgs_sim *s, const double *controls_dev, int64_t step_stride, int n_steps, int rec_every,
                         double *traj_dev) {
  if (!s || !controls_dev || !traj_dev || n_steps < 0 || step_stride < 0 || rec_every < 1)
    return fail(RGS_ERR_ARG, "rgs_step_device_traj: bad argument");
  if (n_steps == 0) return RGS_OK;
  HIP_TRY(hipSetDevice(s->device));
  return launch_steps(s, controls_dev, step_stride, n_steps, traj_dev, rec_every);
}

int rgs_record_size(int n_per_team) { return 3 + 2 + 4 * n_per_team + 4 * n_per_team * n_per_team + 1; }

int rgs_step_device_record(rgs_sim *s, const double *controls_dev, int64_t step_stride, int n_steps, double *rec_dev) {
  if (!s || !controls_dev || !rec_dev || n_steps < 0 || step_stride < 0) return fail(RGS_ERR_ARG, "rgs_step_device_record: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  if (n_steps == 0) return RGS_OK;
  // ONE fused launch: every kernel writes the record of each step right after the step, in the kernel (snapshot_state,
  // gsim_common.hpp).  tprev = the clocks before the launch, for the "executed" row of its first step.
  HIP_TRY(hipMemcpyAsync(s->tprev, s->tcur, sizeof(double) * s->B, hipMemcpyDeviceToDevice, s->stream));
  return launch_steps(s, controls_dev, step_stride, n_steps, nullptr, 0, rec_dev);
}

// ---- native log writer: the reference's Logger (include/utils.h:55-75) opened by GameSim::reset (cpp:48-53)
// and fed by every executed step (cpp:160-170), for many arenas at once ----
namespace {
// Page-locking the staging buffer of the log writer costs more than a whole logged chunk (tens of MB: ~10 ms to pin, ~5 ms to
// release), and Monte-Carlo drivers create one handle per ensemble: ONE staging buffer per device is kept for the life of the
// process and lent to the handle that asks first; a second concurrent logger gets a private one.
struct PinnedLoan { double *ptr = nullptr; size_t bytes = 0; bool lent = false; };
std::mutex g_pin_mutex;
PinnedLoan g_pin[64];

hipError_t pinned_borrow(int device, size_t bytes, double **out, bool *borrowed) {
  *borrowed = false;
  if (device >= 0 && device < 64) {
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    PinnedLoan &L = g_pin[device];
    if (!L.lent) {
      if (L.bytes < bytes) {
        if (L.ptr) (void)hipHostFree(L.ptr);
        L.ptr = nullptr; L.bytes = 0;
        hipError_t e = hipHostMalloc((void **)&L.ptr, bytes, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        L.bytes = bytes;
      }
      L.lent = true; *out = L.ptr; *borrowed = true;
      return hipSuccess;
    }
  }
  return hipHostMalloc((void **)out, bytes, hipHostMallocDefault);
}
void pinned_return(int device, double *ptr, bool borrowed) {
  if (!ptr) return;
  if (borrowed) { std::lock_guard<std::mutex> lock(g_pin_mutex); g_pin[device].lent = false; }
  else (void)hipHostFree(ptr);
}

void log_release_buffers(rgs_sim *s) {
  dev_free(s, s->log_rec); dev_free(s, s->log_runs);
  pinned_return(s->device, s->log_runs_host, s->log_runs_borrowed);
  s->log_runs_host = nullptr; s->log_runs_borrowed = false;
  s->log_cap = 0;
}
int log_threads(size_t n_files) {
  unsigned hw = std::thread::hardware_concurrency();
  size_t t = hw == 0 ? 4 : (hw > 16 ? 16 : hw);
  return (int)(t < n_files ? t : (n_files ? n_files : 1));
}
}  // namespace

int rgs_log_close(rgs_sim *s) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_log_close: null handle");
  int bad = 0;
  for (int fd : s->log_fds) if (fd >= 0 && ::close(fd) != 0) bad = errno;
  s->log_fds.clear();
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  log_release_buffers(s);
  dev_free(s, s->log_ids); dev_free(s, s->log_count);
  if (s->log_count_host) { (void)hipHostFree(s->log_count_host); s->log_count_host = nullptr; }
  if (bad) return fail(RGS_ERR_ARG, std::string("rgs_log_close: close: ") + std::strerror(bad));
  return RGS_OK;
}

int rgs_log_open(rgs_sim *s, int n_files, const int *arena_ids, const char *const *paths) {
  if (!s || n_files < 1 || !arena_ids || !paths) return fail(RGS_ERR_ARG, "rgs_log_open: bad argument");
  for (int q = 0; q < n_files; ++q)
    if (arena_ids[q] < 0 || arena_ids[q] >= s->B || !paths[q]) return fail(RGS_ERR_ARG, "rgs_log_open: arena id out of range or null path");
  int rc = rgs_log_close(s);                                   // like Logger::open: a log that is still open is closed first
  if (rc != RGS_OK) return rc;
  {                                                            // every file stays open until rgs_log_close: mind the descriptor limit
    struct rlimit lim;
    if (getrlimit(RLIMIT_NOFILE, &lim) == 0 && lim.rlim_cur != RLIM_INFINITY && (rlim_t)n_files + 256 > lim.rlim_cur) {
      const rlim_t want = (rlim_t)n_files + 256;
      if (lim.rlim_max == RLIM_INFINITY || want <= lim.rlim_max) {
        lim.rlim_cur = want;
        (void)setrlimit(RLIMIT_NOFILE, &lim);                  // soft limit up to what is needed, inside the hard limit
      } else {
        return fail(RGS_ERR_UNSUPPORTED, "rgs_log_open: " + std::to_string(n_files) + " log files exceed the process's open-file limit (RLIMIT_NOFILE hard limit " +
                                             std::to_string((unsigned long long)lim.rlim_max) + "); log fewer arenas per handle or raise the limit");
      }
    }
  }
  HIP_TRY(hipSetDevice(s->device));
  s->log_fds.assign((size_t)n_files, -1);
  const int T = log_threads((size_t)n_files);
  std::vector<int> err((size_t)T, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t]() {
      for (int q = t; q < n_files; q += T) {
        const int fd = ::open(paths[q], O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);   // std::ofstream::open truncates
        if (fd < 0) { err[(size_t)t] = errno; return; }
        s->log_fds[(size_t)q] = fd;
      }
    });
  for (auto &th : pool) th.join();
  for (int e : err)
    if (e) { (void)rgs_log_close(s); return fail(RGS_ERR_ARG, std::string("rgs_log_open: open: ") + std::strerror(e)); }
  HIP_TRY(dev_alloc(s, &s->log_ids, sizeof(int) * (size_t)n_files, "log_ids"));
  HIP_TRY(dev_alloc(s, &s->log_count, sizeof(int) * (size_t)n_files, "log_count"));
  HIP_TRY(hipHostMalloc((void **)&s->log_count_host, sizeof(int) * (size_t)n_files, hipHostMallocDefault));
  HIP_TRY(hipMemcpyAsync(s->log_ids, arena_ids, sizeof(int) * (size_t)n_files, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));                    // arena_ids is the caller's memory
  return RGS_OK;
}

int rgs_step_device_log(rgs_sim *s, const double *controls_dev, int64_t step_stride, int n_steps) {
  if (!s || !controls_dev || n_steps < 0 || step_stride < 0) return fail(RGS_ERR_ARG, "rgs_step_device_log: bad argument");
  if (s->log_fds.empty()) return fail(RGS_ERR_ARG, "rgs_step_device_log: no log open (rgs_log_open)");
  HIP_TRY(hipSetDevice(s->device));
  if (n_steps == 0) return RGS_OK;
  const int R = rgs_record_size(s->n), W = R - 1;
  const size_t n_files = s->log_fds.size();
  if (n_steps > s->log_cap) {
    HIP_TRY(hipStreamSynchronize(s->stream));
    log_release_buffers(s);
    HIP_TRY(dev_alloc(s, &s->log_rec, sizeof(double) * (size_t)n_steps * R * s->B, "log_rec"));
    HIP_TRY(dev_alloc(s, &s->log_runs, sizeof(double) * n_files * (size_t)n_steps * W, "log_runs"));
    HIP_TRY(pinned_borrow(s->device, sizeof(double) * n_files * (size_t)n_steps * W, &s->log_runs_host, &s->log_runs_borrowed));
    s->log_cap = n_steps;
  }
  int rc = rgs_step_device_record(s, controls_dev, step_stride, n_steps, s->log_rec);
  if (rc != RGS_OK) return rc;
  HIP_TRY(hipMemsetAsync(s->log_count, 0, sizeof(int) * n_files, s->stream));
  const long long items = (long long)n_files * n_steps;
  hipLaunchKernelGGL(k_gather_runs, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s->stream, s->log_rec, s->B, R, n_steps,
                     (int)n_files, s->log_ids, s->log_runs, s->log_count);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->log_runs_host, s->log_runs, sizeof(double) * n_files * (size_t)n_steps * W, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(s->log_count_host, s->log_count, sizeof(int) * n_files, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  const int T = log_threads(n_files);
  std::vector<int> err((size_t)T, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&, t]() {
      for (size_t q = (size_t)t; q < n_files; q += (size_t)T) {
        const char *src = (const char *)(s->log_runs_host + q * (size_t)n_steps * W);
        size_t left = sizeof(double) * (size_t)W * (size_t)s->log_count_host[q];
        while (left) {
          const ssize_t w = ::write(s->log_fds[q], src, left);
          if (w < 0) { if (errno == EINTR) continue; err[(size_t)t] = errno; return; }
          src += w; left -= (size_t)w;
        }
      }
    });
  for (auto &th : pool) th.join();
  for (int e : err) if (e) return fail(RGS_ERR_ARG, std::string("rgs_step_device_log: write: ") + std::strerror(e));
  return RGS_OK;
}

int rgs_sync(rgs_sim *s) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_sync: null handle");
  HIP_TRY(hipSetDevice(s->device));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_get_state(rgs_sim *s, double *out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_get_state: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out, s->state, sizeof(double) * s->S * s->B, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_undecided(rgs_sim *s, uint8_t *out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_undecided: null argument");
  HIP_TRY(hipSetDevice(s->device));
  const int tb = 256;
  hipLaunchKernelGGL(k_undecided, dim3((unsigned)((s->B + tb - 1) / tb)), dim3(tb), 0, s->stream, s->state, s->B,
                     (double)s->win, s->flags);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, s->flags, (size_t)s->B, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_get_damage(rgs_sim *s, double *out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_get_damage: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out, s->damage, sizeof(double) * s->P * s->P * s->B, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_get_time(rgs_sim *s, double *out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_get_time: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out, s->tcur, sizeof(double) * s->B, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_set_state(rgs_sim *s, const double *state_host, const double *time_host) {
  if (!s || !state_host || !time_host) return fail(RGS_ERR_ARG, "rgs_set_state: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(s->state, state_host, sizeof(double) * s->S * s->B, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->tcur, time_host, sizeof(double) * s->B, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemsetAsync(s->damage, 0, sizeof(double) * s->P * s->P * s->B, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_set_damage(rgs_sim *s, const double *damage_host) {
  if (!s || !damage_host) return fail(RGS_ERR_ARG, "rgs_set_damage: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(s->damage, damage_host, sizeof(double) * s->P * s->P * s->B, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_state_device_ptr(rgs_sim *s, double **out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_state_device_ptr: null argument");
  *out = s->state;
  return RGS_OK;
}

int rgs_copy_state_to_device(rgs_sim *s, double *dst) {
  if (!s || !dst) return fail(RGS_ERR_ARG, "rgs_copy_state_to_device: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(dst, s->state, sizeof(double) * s->S * s->B, hipMemcpyDeviceToDevice, s->stream));
  return RGS_OK;
}

int rgs_stream(rgs_sim *s, void **out) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_stream: null argument");
  *out = (void *)s->stream;
  return RGS_OK;
}

int rgs_device_alloc(rgs_sim *s, int64_t n, double **out) {
  if (!s || !out || n < 1) return fail(RGS_ERR_ARG, "rgs_device_alloc: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(dev_alloc(s, out, sizeof(double) * (size_t)n, "rgs_device_alloc"));
  return RGS_OK;
}

int rgs_device_free(rgs_sim *s, double *dev) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_device_free: null handle");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  dev_free(s, dev);
  return RGS_OK;
}

int rgs_device_upload(rgs_sim *s, double *dst, const double *src, int64_t n) {
  if (!s || !dst || !src || n < 0) return fail(RGS_ERR_ARG, "rgs_device_upload: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_device_download(rgs_sim *s, double *dst, const double *src, int64_t n) {
  if (!s || !dst || !src || n < 0) return fail(RGS_ERR_ARG, "rgs_device_download: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

// Pinned host memory: lets rgs_step_pinned overlap nothing but at least run the PCIe copies at
// full rate (pageable numpy buffers are staged by the runtime).
int rgs_host_alloc(int64_t n_doubles, double **out_host) {
  if (!out_host || n_doubles < 1) return fail(RGS_ERR_ARG, "rgs_host_alloc: bad argument");
  HIP_TRY(hipHostMalloc((void **)out_host, sizeof(double) * (size_t)n_doubles, hipHostMallocDefault));
  return RGS_OK;
}

int rgs_host_free(double *host) {
  HIP_TRY(hipHostFree(host));
  return RGS_OK;
}

// One closed-loop step for a host-side controller (py/test_dmpc.py:43-47 shape): controls H2D,
// one step, state D2H, all asynchronous on the handle's stream, one synchronisation at the end.
int rgs_step_roundtrip(rgs_sim *s, const double *controls_host, double *state_host) {
  if (!s || !controls_host || !state_host) return fail(RGS_ERR_ARG, "rgs_step_roundtrip: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(s->ctl, controls_host, sizeof(double) * s->P * 2 * s->B, hipMemcpyHostToDevice, s->stream));
  int rc = launch_steps(s, s->ctl, 0, 1);
  if (rc != RGS_OK) return rc;
  HIP_TRY(hipMemcpyAsync(state_host, s->state, sizeof(double) * s->S * s->B, hipMemcpyDeviceToHost, s->stream));
  RGS_TRY(finish(s));
  return RGS_OK;
}

int rgs_timer_start(rgs_sim *s) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_timer_start: null handle");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
  return RGS_OK;
}

int rgs_timer_stop(rgs_sim *s, float *ms_out) {
  if (!s || !ms_out) return fail(RGS_ERR_ARG, "rgs_timer_stop: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(ms_out, s->ev0, s->ev1));
  return RGS_OK;
}

int rgs_get_counters(rgs_sim *s, int64_t *out8) {
  if (!s || !out8) return fail(RGS_ERR_ARG, "rgs_get_counters: null argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out8, s->counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->pooled_in_flight = false;
  if (out8[6] != 0) s->poisoned = true;
  out8[7] = s->auto_wave ? 1 : 0;
  if (s->poisoned) return fail(RGS_ERR_HIP, kPoolTimeoutText);
  return RGS_OK;
}

// The 8 device counters as they are (out[7] is not used), then out[8] = AUTO's current preference for the wave-per-arena
// kernel.  capacity >= 9.
int rgs_get_counters_ex(rgs_sim *s, int64_t *out, int capacity) {
  if (!s || !out || capacity < 9) return fail(RGS_ERR_ARG, "rgs_get_counters_ex: bad argument");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipMemcpyAsync(out, s->counters, sizeof(unsigned long long) * CNT_N, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->pooled_in_flight = false;
  if (out[CNT_POOL_TIMEOUT] != 0) s->poisoned = true;
  out[8] = s->auto_wave ? 1 : 0;
  if (s->poisoned) return fail(RGS_ERR_HIP, kPoolTimeoutText);
  return RGS_OK;
}

// Diagnostic: per-phase shader-cycle totals of the hybrid kernels since the last reset (or clear).
// enable: bit 0 = stamps on for later launches, bit 1 = clear the totals first.  out: min(capacity, STAMP_N = 17) values.
int rgs_get_stamps(rgs_sim *s, int enable, int64_t *out, int capacity) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_get_stamps: null handle");
  HIP_TRY(hipSetDevice(s->device));
  s->stamps = (enable & 1) != 0;
  if (enable & 2) HIP_TRY(hipMemsetAsync(s->counters + CNT_N, 0, sizeof(unsigned long long) * STAMP_N, s->stream));
  if (out && capacity > 0) {
    const int nv = capacity < (int)STAMP_N ? capacity : (int)STAMP_N;
    HIP_TRY(hipMemcpyAsync(out, s->counters + CNT_N, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  return RGS_OK;
}

// Debug (RGS_DEBUG_GUARD_BYTES=n): verifies the 0xA5 guard zones around every device buffer of the handle.
// Diagnostic (library built with STAMPS=1, stamps enabled): 8 values per wavefront of the last asynchronous-pool launch:
// cycles in the kernel, in lane passes, serving rounds, lane passes, rounds served, cycles until its own arenas were
// done, arena-steps it posted, arena-steps it ran for the pool.  out: [n_waves][8], n_waves = 4 * ceil(B / 256).
int rgs_debug_wave_log(rgs_sim *s, int64_t *out, int64_t capacity_waves) {
  if (!s || !out) return fail(RGS_ERR_ARG, "rgs_debug_wave_log: null argument");
  HIP_TRY(hipSetDevice(s->device));
  const long long n_waves = 4 * ((s->B + kDealWG - 1) / kDealWG);
  if (capacity_waves < n_waves) return fail(RGS_ERR_ARG, "rgs_debug_wave_log: buffer too small");
  if (!s->wave_log) {                                          // first call: allocate; later launches fill it
    HIP_TRY(dev_alloc(s, &s->wave_log, sizeof(unsigned long long) * 8 * n_waves, "wave_log"));
    HIP_TRY(hipMemsetAsync(s->wave_log, 0, sizeof(unsigned long long) * 8 * n_waves, s->stream));
  }
  HIP_TRY(hipMemcpyAsync(out, s->wave_log, sizeof(unsigned long long) * 8 * n_waves, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_debug_check_guards(rgs_sim *s) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_debug_check_guards: null handle");
  const size_t g = guard_bytes();
  if (g == 0) return RGS_OK;
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::vector<unsigned char> host(g);
  for (const GuardedBuffer &b : s->guards) {
    for (int side = 0; side < 2; ++side) {
      const char *zone = side == 0 ? b.base : b.base + g + b.bytes;
      HIP_TRY(hipMemcpy(host.data(), zone, g, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < g; ++i)
        if (host[i] != 0xA5) {
          const long long off = side == 0 ? (long long)i - (long long)g : (long long)(b.bytes + i);
          return fail(RGS_ERR_HIP, std::string("guard zone of device buffer '") + b.name + "' (" + std::to_string(b.bytes) +
                                       " bytes) overwritten at byte offset " + std::to_string(off));
        }
    }
  }
  return RGS_OK;
}

// Debug / test aid: makes the handle look as if a pooled kernel had reported a protocol timeout (device counter
// CNT_POOL_TIMEOUT = 1), so that the sticky-error path of the synchronising calls can be exercised.
int rgs_debug_inject_pool_timeout(rgs_sim *s) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_debug_inject_pool_timeout: null handle");
  HIP_TRY(hipSetDevice(s->device));
  const unsigned long long one = 1ull;
  HIP_TRY(hipMemcpyAsync(s->counters + CNT_POOL_TIMEOUT, &one, sizeof one, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->pooled_in_flight = true;
  return RGS_OK;
}

int rgs_set_variant(rgs_sim *s, int variant) {
  if (!s || variant < RGS_VARIANT_AUTO || variant > RGS_VARIANT_APOOL) return fail(RGS_ERR_ARG, "rgs_set_variant: bad argument");
  s->variant = variant;
  return RGS_OK;
}

// 1: the device generator's log() is the host's, bit for bit (gsim_hostlog.hpp accepted at build time): noise mode 1 then
// reproduces the reference's draws and trajectories exactly; 0: the device library's log, statistical parity only.
int rgs_noise_is_exact(void) { return GSIM_HOST_LOG ? 1 : 0; }

int rgs_set_noise_mode(rgs_sim *s, int mode) {
  if (!s || (mode != 0 && mode != 1)) return fail(RGS_ERR_ARG, "rgs_set_noise_mode: mode must be 0 (host tapes) or 1 (device generator)");
  s->noise_mode = mode;                                        // takes effect at the next rgs_reset (the engines are seeded there)
  return RGS_OK;
}

// The first n normals of the device generator's stream for `seed` (n <= 4096) and the 2 * n_attempts canonical
// uniforms of its first n_attempts polar attempts; *x_end = engine state after the attempts it evaluated, *have = the
// normals those produced (>= n).
int rgs_selftest_rng(int device_id, long long seed, int n, int n_attempts, double *normals_host, double *canonicals_host,
                     uint32_t *x_end, int *have) {
  if (!normals_host || !canonicals_host || n < 2 || n > 4096 || n_attempts < 1) return fail(RGS_ERR_ARG, "rgs_selftest_rng: bad argument");
  HIP_TRY(hipSetDevice(device_id));
  DeviceScratch scratch;
  double *nd = nullptr, *cd = nullptr, *misc = nullptr;
  HIP_TRY(scratch.alloc(&nd, sizeof(double) * n));
  HIP_TRY(scratch.alloc(&cd, sizeof(double) * 2 * (size_t)n_attempts));
  HIP_TRY(scratch.alloc(&misc, 16));
  hipLaunchKernelGGL(k_selftest_rng, dim3(1), dim3(64), sizeof(double) * (n + kRngSlack + kLogTabDoubles), 0, lcg_seed(seed), n, n_attempts, nd, cd,
                     (unsigned *)misc, (int *)misc + 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(normals_host, nd, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(canonicals_host, cd, sizeof(double) * 2 * (size_t)n_attempts, hipMemcpyDeviceToHost));
  unsigned two[2];
  HIP_TRY(hipMemcpy(two, misc, 8, hipMemcpyDeviceToHost));
  if (x_end) *x_end = two[0];
  if (have) *have = (int)two[1];
  return RGS_OK;
}

int rgs_set_rebalance(rgs_sim *s, int enable) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_set_rebalance: null handle");
  s->deal_enabled = enable != 0;
  if (!s->deal_enabled) s->perm_valid = false;
  return RGS_OK;
}

int rgs_get_heat(rgs_sim *s, uint32_t *heat_host) {
  if (!s || !heat_host) return fail(RGS_ERR_ARG, "rgs_get_heat: null argument");
  HIP_TRY(hipSetDevice(s->device));
  if (!s->heat) { for (long long i = 0; i < s->B; ++i) heat_host[i] = 0u; return RGS_OK; }
  HIP_TRY(hipMemcpyAsync(heat_host, s->heat, sizeof(unsigned) * (size_t)s->B, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_get_rebalance_info(rgs_sim *s, int *n_swapped, int *perm_host) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_get_rebalance_info: null handle");
  HIP_TRY(hipSetDevice(s->device));
  if (n_swapped) *n_swapped = 0;
  if (perm_host) for (long long i = 0; i < s->B; ++i) perm_host[i] = (int)i;
  if (!s->perm || !s->perm_valid) return RGS_OK;
  if (n_swapped) HIP_TRY(hipMemcpyAsync(n_swapped, s->n_swapped, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  if (perm_host) HIP_TRY(hipMemcpyAsync(perm_host, s->perm, sizeof(int) * (size_t)(s->B / kDealWG) * kDealWG, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RGS_OK;
}

int rgs_set_fast_division(rgs_sim *s, int enable) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_set_fast_division: null handle");
  s->fast_div = enable != 0;
  s->one_step_enabled = enable >= 2;
  s->one_step_ok = s->one_step_enabled && one_step_certified(s->tau_player) && one_step_certified(s->tau_puck);
  return RGS_OK;
}

int rgs_set_sparse_watch(rgs_sim *s, int enable) {
  if (!s) return fail(RGS_ERR_ARG, "rgs_set_sparse_watch: null handle");
  if (enable != 0 && enable != 1 && enable != 2 && enable != 5 && enable != 10)
    return fail(RGS_ERR_ARG, "rgs_set_sparse_watch: interval must be 0 (off), 1 (default interval), 2, 5 or 10");
  s->sparse_watch = enable != 0;
  s->watch_every = enable >= 2 ? enable : 0;                  // 1 = back to the default interval
  return RGS_OK;
}

int rgs_division_info(double tau, int *fast_ok, int *pow2, int *one_step, double *worst_out, int worst_cap,
                      int *worst_count) {
  if (fast_ok) *fast_ok = fast_division_ok(tau) ? 1 : 0;
  if (pow2) *pow2 = is_power_of_two(tau) ? 1 : 0;
  std::vector<double> w;
  const bool ok = one_step_certified(tau, &w);
  if (one_step) *one_step = ok ? 1 : 0;
  if (worst_count) *worst_count = (int)w.size();
  if (worst_out)
    for (int i = 0; i < worst_cap && i < (int)w.size(); ++i) worst_out[i] = w[i];
  return RGS_OK;
}

int rgs_selftest_divtau(int device_id, const double *a_host, int64_t n, double tau, double *fast_host,
                        double *ieee_host, double *one_host) {
  if (!a_host || !fast_host || !ieee_host || !one_host || n < 1) return fail(RGS_ERR_ARG, "rgs_selftest_divtau: bad argument");
  if (!fast_division_ok(tau)) return fail(RGS_ERR_UNSUPPORTED, "rgs_selftest_divtau: tau outside the fast-division domain");
  HIP_TRY(hipSetDevice(device_id));
  double *a = nullptr, *f = nullptr, *q = nullptr, *o = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  DeviceScratch scratch;
  HIP_TRY(scratch.alloc(&a, bytes));
  HIP_TRY(scratch.alloc(&f, bytes));
  HIP_TRY(scratch.alloc(&q, bytes));
  HIP_TRY(scratch.alloc(&o, bytes));
  HIP_TRY(hipMemcpy(a, a_host, bytes, hipMemcpyHostToDevice));
  const int tb = 256;
  hipLaunchKernelGGL(k_selftest_divtau, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, 0, a, (long long)n, tau,
                     1.0 / tau, f, q, o);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(fast_host, f, bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ieee_host, q, bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(one_host, o, bytes, hipMemcpyDeviceToHost));
  return RGS_OK;
}

int rgs_selftest_math(int device_id, const double *a_host, const double *b_host, int64_t n, double *quot_host,
                      double *sqrt_host) {
  if (!a_host || !b_host || !quot_host || !sqrt_host || n < 1) return fail(RGS_ERR_ARG, "rgs_selftest_math: bad argument");
  HIP_TRY(hipSetDevice(device_id));
  double *a = nullptr, *b = nullptr, *q = nullptr, *r = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n;
  DeviceScratch scratch;
  HIP_TRY(scratch.alloc(&a, bytes));
  HIP_TRY(scratch.alloc(&b, bytes));
  HIP_TRY(scratch.alloc(&q, bytes));
  HIP_TRY(scratch.alloc(&r, bytes));
  HIP_TRY(hipMemcpy(a, a_host, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b, b_host, bytes, hipMemcpyHostToDevice));
  const int tb = 256;
  hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, 0, a, b, (long long)n, q, r);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(quot_host, q, bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sqrt_host, r, bytes, hipMemcpyDeviceToHost));
  return RGS_OK;
}

}  // extern "C"
