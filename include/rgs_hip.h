/* rgs_hip.h -- C ABI of the MI355X batched robo-game-sim stepper (librgs_hip.so).
 *
 * Drop-in boundary for the reference's GameSim hot path.  The reference exposes
 * the path to Python through pybind11 (src/RoboGamePy.cpp:11-25, class GameSim:
 * reset / undecided / run, declared in include/GameSim.h:21-27).  This library
 * is what a replacement binding links instead of src/GameSim.cpp: plain
 * pointers and sizes, int status codes, no exceptions, caller-owned buffers.
 * One handle = one batch of independent arenas resident on one GPU.  Calls on a
 * handle must be serialised by the caller (the reference is single-threaded too).
 *
 * Array layouts (all IEEE binary64, arena index fastest = "SoA over arenas"):
 *   state     [S][B]        S = 2 + 4*(2n+1); rows in include/SimState.h:14-23
 *                           order: TAS, TBS, PK(px,py,vx,vy), A1(4)..An(4), B1(4)..Bn(4)
 *   controls  [P][2][B]     P = 2n players in disc order A1..An,B1..Bn
 *   x0_ball   [4][B]        or a single [4] broadcast to every arena
 *   damage    [P*P][B]      row-major (aggressor, victim), include/SimState.h:33-38
 *   time      [B]           the reference's t_ (src/GameSim.cpp:158)
 */
#ifndef RGS_HIP_H
#define RGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rgs_sim rgs_sim;

enum {
  RGS_OK = 0,
  RGS_ERR_ARG = -1,          /* bad argument (null, range, n_per_team not in 1..7) */
  RGS_ERR_HIP = -2,          /* a HIP runtime call failed; see rgs_last_error() */
  RGS_ERR_UNSUPPORTED = -3,  /* request outside this build's limits (e.g. noise > 0 with > 4096 arenas) */
  RGS_ERR_NO_DEVICE = -4     /* no gfx950 device visible */
};

/* Stepper variants (rgs_set_variant).  All give bit-identical results.  AUTO picks the kernel with the
 * highest measured rate for this team size, batch size, launch length and step length (csrc/auto_rates.h,
 * generated from a sweep on MI355X) and, looking at the event counters every >= 64 steps, falls back to the
 * wave-per-arena kernel while most arenas are in contact. */
enum {
  RGS_VARIANT_AUTO = 0,
  RGS_VARIANT_WAVE = 1,      /* one wavefront owns one arena, grid in LDS */
  RGS_VARIANT_HYBRID = 2,    /* lane-per-arena streaming pass + LDS event pass */
  RGS_VARIANT_HYBRID2 = 3,   /* the same with two lanes (x chains, y chains) per arena */
  RGS_VARIANT_POOL = 4,      /* HYBRID with four wavefronts per workgroup whose flagged arena-steps go through one queue in
                                LDS: fuller event-pass rounds, wavefronts help each other, early finishers keep serving */
  RGS_VARIANT_APOOL = 5      /* POOL with per-arena step counts: a flagged arena's lane sits out lane passes until its step
                                is back from the event pass, the rest of its wavefront goes on (bounded lag) */
};

/* Human-readable text for the most recent failure on this thread. */
const char *rgs_last_error(void);

/* Number of doubles in one arena's state vector (SimState::SIZE, include/SimState.h:22; 22 at n=2). */
int rgs_state_size(int n_per_team);

/* GameSim::GameSim (src/GameSim.cpp:3-17) for n_arenas independent arenas on `device_id`.
 * The arenas start as after reset() with the C++ default arguments (src/GameSim.cpp:21-25). */
int rgs_create(int64_t n_arenas, int n_per_team, int device_id, rgs_sim **out);
int rgs_destroy(rgs_sim *sim);

/* GameSim::reset (include/GameSim.h:21-24, src/GameSim.cpp:21-55).  log/logname are host-side
 * concerns and live in the Python layer.  x0_ball_host: [4] if !x0_batched else [4][B].
 * noise != 0 (w_stdev_, src/GameSim.cpp:30,66-67): arena a draws from the reference's own generator
 * (std::default_random_engine(seed + a) + std::normal_distribution, its cached draw surviving resets
 * like the reference's w_dist_); the draws are generated on the host and consumed by the one-wave-
 * per-arena kernel in the reference's call order, so results match the reference bit for bit.  This
 * mode steps one launch per step and is limited to 4096 arenas per handle (RGS_ERR_UNSUPPORTED
 * beyond).  With noise == 0 the seed has no effect on the physics. */
int rgs_reset(rgs_sim *sim, double dt, int winning_score, const double *x0_ball_host, int x0_batched,
              double noise, int seed, double tau_puck, double tau_player, double player_mass,
              double puck_mass);

/* GameSim::run (include/GameSim.h:26-27, src/GameSim.cpp:118-124) on every arena: arenas that are
 * still undecided advance one step, decided arenas are left untouched.  controls_host [P][2][B]. */
int rgs_step(rgs_sim *sim, const double *controls_host);

/* Same, controls already on the device: n_steps consecutive steps, step k reading
 * controls_dev + k*step_stride (doubles; stride 0 = hold one set).  Asynchronous on the
 * handle's stream; rgs_sync() or any rgs_get_* waits for it. */
int rgs_step_device(rgs_sim *sim, const double *controls_dev, int64_t step_stride, int n_steps);

/* rgs_step_device with snapshots: after every rec_every-th step of this call the state [S][B] is copied
 * to traj_dev[(k / rec_every) - 1] (device memory, floor(n_steps / rec_every) snapshots), inside the
 * same fused launch. */
int rgs_step_device_traj(rgs_sim *sim, const double *controls_dev, int64_t step_stride, int n_steps, int rec_every,
                         double *traj_dev);

/* Stepping with the reference's log record captured after every step (src/GameSim.cpp:161-170,
 * include/utils.h:64-76).  rec_dev: [n_steps][R][B] doubles on the device, R =
 * rgs_record_size(n) = 3 + 2 + 2P + P*P + 1: t, TAS, TBS, puck xy, players' xy, the P damage blocks in
 * the reference's byte order, and a final "executed" row (1.0 where the step really ran; the
 * reference writes no record for a decided game).  One fused launch: the records are written inside the stepping kernel
 * after each step (round 2: one step launch + one record launch per step). */
int rgs_record_size(int n_per_team);
int rgs_step_device_record(rgs_sim *sim, const double *controls_dev, int64_t step_stride, int n_steps,
                           double *rec_dev);

/* The reference's log FILES written by the library: replaces Logger::open / log / logVectors
 * (include/utils.h:55-75) as driven by GameSim::reset (src/GameSim.cpp:48-53) and GameSim::run (:160-170), for many
 * arenas of a batch at once.
 *   rgs_log_open         creates / truncates paths[q] (as std::ofstream::open does) for arena arena_ids[q], q < n_files;
 *                        a log that is still open is closed first;
 *   rgs_step_device_log  steps like rgs_step_device_record (one fused launch) and appends to each open file the
 *                        records of the steps its arena executed: R - 1 = 29 (2v2) native-endian doubles per step,
 *                        nothing for a decided game (cpp:121-123) -- byte for byte what the reference writes; the
 *                        per-file gather runs on the device, the writes on up to 16 host threads; returns when the
 *                        data is with the operating system;
 *   rgs_log_close        closes the files (also done by rgs_destroy). */
int rgs_log_open(rgs_sim *sim, int n_files, const int *arena_ids, const char *const *paths);
int rgs_step_device_log(rgs_sim *sim, const double *controls_dev, int64_t step_stride, int n_steps);
int rgs_log_close(rgs_sim *sim);

int rgs_sync(rgs_sim *sim);

/* The value GameSim::run / reset return (state_.vector(), src/GameSim.cpp:54,123), for all arenas. */
int rgs_get_state(rgs_sim *sim, double *state_host);
/* GameSim::undecided (src/GameSim.cpp:57-60) per arena: 1 = still running. */
int rgs_undecided(rgs_sim *sim, uint8_t *out_host);
/* SimState::damage (include/SimState.h:33), not reachable through the reference binding. */
int rgs_get_damage(rgs_sim *sim, double *damage_host);
int rgs_get_time(rgs_sim *sim, double *time_host);

/* Extension: overwrite state and time of every arena; damage is zeroed -- follow with
 * rgs_set_damage to restore a checkpoint taken with rgs_get_state / rgs_get_time / rgs_get_damage.  The event
 * counters are not part of a checkpoint, and with noise != 0 neither are the per-arena generators (a restored
 * noisy game continues with the handle's current generator state, not the checkpoint's). */
int rgs_set_state(rgs_sim *sim, const double *state_host, const double *time_host);
int rgs_set_damage(rgs_sim *sim, const double *damage_host);   /* [P*P][B], layout of rgs_get_damage */

/* Zero-copy access for device-side callers (torch tensors, RCCL all-gather). */
int rgs_state_device_ptr(rgs_sim *sim, double **out_dev);   /* [S][B] */
int rgs_copy_state_to_device(rgs_sim *sim, double *dst_dev); /* async on the handle's stream */
int rgs_stream(rgs_sim *sim, void **out_hip_stream);

/* Device memory helpers so a ctypes host can own control tables without another runtime. */
int rgs_device_alloc(rgs_sim *sim, int64_t n_doubles, double **out_dev);
int rgs_device_free(rgs_sim *sim, double *dev);
int rgs_device_upload(rgs_sim *sim, double *dst_dev, const double *src_host, int64_t n_doubles);
int rgs_device_download(rgs_sim *sim, double *dst_host, const double *src_dev, int64_t n_doubles);

/* Closed-loop use by a host-side controller (py/test_dmpc.py:43-47, py/run_sim.py:56-75): upload this
 * step's controls [P][2][B], advance one step, download the state [S][B]; one synchronisation.
 * rgs_host_alloc gives page-locked buffers so both copies run at PCIe rate. */
int rgs_step_roundtrip(rgs_sim *sim, const double *controls_host, double *state_host);
int rgs_host_alloc(int64_t n_doubles, double **out_host);
int rgs_host_free(double *host);

/* HIP-event timer on the handle's stream: start ... launches ... stop -> milliseconds. */
int rgs_timer_start(rgs_sim *sim);
int rgs_timer_stop(rgs_sim *sim, float *ms_out);

/* Run-time counters summed over arenas since the last reset:
 * out[0] events resolved, [1] steps ended by the 50-iteration cap (include/GameSim.h:7),
 * [2] base index past the grid (reference: out-of-bounds write; clamped here), [3] puck events
 * skipped after a goal (src/GameSim.cpp:230-231), [4] arena-steps whose first scan found an
 * event, [5] arena-steps the hybrid's lane pass handed to the wave pass, [6] bounded waits of the pooled
 * kernels that ran out (a protocol error: never seen; non-zero poisons the handle -- this call and every
 * synchronising call (rgs_sync, rgs_get_*, rgs_step, rgs_step_roundtrip, rgs_device_download) return
 * RGS_ERR_HIP until the next rgs_reset), [7] 1 if RGS_VARIANT_AUTO currently prefers the wave-per-arena
 * kernel (sustained-contact regime), else 0. */
int rgs_get_counters(rgs_sim *sim, int64_t *out8);
/* The same counters, raw: out[0..6] as above, out[7] unused, out[8] = the AUTO flag that rgs_get_counters reports in
 * [7].  capacity >= 9. */
int rgs_get_counters_ex(rgs_sim *sim, int64_t *out, int capacity);

/* Diagnostic in-kernel stamps (s_memtime) of the hybrid kernels; only in a library built with
 * `make STAMPS=1` (compiled out otherwise: all zeros), and off until enabled.  enable: bit 0 = stamps on
 * for later launches, bit 1 = clear the totals first.  out (may be NULL) receives up to 17 values since the
 * last reset/clear.  Summed over waves: lane-pass cycles, wave-pass cycles, wave-pass rounds,
 * whole-kernel cycles; for lane group 0 of every wave pass: fill, first-scan, event-loop, commit cycles
 * and the arena count; maxima over waves: whole-kernel cycles, wave-pass cycles, rounds, and the longest
 * single event loop; then, summed over group 0's carried-out events: cycles from the pop to the start of
 * the carry-out, of the carry-out, of the re-scan chunks, and the event count. */
int rgs_get_stamps(rgs_sim *sim, int enable, int64_t *out, int capacity);   /* writes min(capacity, 17) values */

/* Diagnostic (library built with STAMPS=1, stamps enabled): per-wavefront records of the last asynchronous-pool launch,
 * out [n_waves][8], n_waves = 4 * ceil(B / 256): cycles in the kernel, in lane passes, serving event-pass rounds, number of
 * lane passes, rounds served, cycles until the wavefront's own arenas were done, arena-steps posted, arena-steps run for the
 * pool.  The first call allocates the device buffer (zeros); launches after it fill it. */
int rgs_debug_wave_log(rgs_sim *sim, int64_t *out, int64_t capacity_waves);

/* Debug aid (GPU AddressSanitizer is not available here): with RGS_DEBUG_GUARD_BYTES=<n> in the environment every
 * device buffer of a handle carries n guard bytes on both sides; this call returns RGS_ERR_HIP and names the first
 * buffer whose guard zone was overwritten.  Without the variable it is a no-op returning RGS_OK. */
int rgs_debug_check_guards(rgs_sim *sim);

/* Debug / test aid: sets the pooled kernels' timeout counter as a failed launch would, so that callers can exercise
 * the sticky error (rgs_get_counters [6]). */
int rgs_debug_inject_pool_timeout(rgs_sim *sim);

int rgs_set_variant(rgs_sim *sim, int variant);

/* noise != 0 (w_stdev_, src/GameSim.cpp:30,66-67), two modes, chosen BEFORE rgs_reset:
 *   0 host tapes: the draws come from the reference's own std::default_random_engine / std::normal_distribution on the
 *     host (<= 4096 arenas per handle, one launch per step): bit-identical to the reference;
 *   1 device generator: the same engine (minstd_rand0, integer, bit-exact), the same generate_canonical and the same
 *     Marsaglia polar method run on the GPU (csrc/gsim_rng.hpp), normals in the reference's order.  Any batch size, fused
 *     launches.  Where rgs_noise_is_exact() = 1 (below) its log() is the host's too and the mode is bit-identical to the
 *     reference like mode 0 -- it is then the DEFAULT; otherwise log() is the device library's, the normals agree with the
 *     host's to ~1 ulp, a noisy game matches the reference statistically, and mode 0 is the default.
 * rgs_selftest_rng: the device generator's first n (<= 4096) normals for a seed and the canonical uniforms of its first
 * n_attempts polar attempts, for comparison with the host's <random>. */
int rgs_set_noise_mode(rgs_sim *sim, int mode);
/* 1 if this build's device generator uses the HOST's log() (glibc's table-driven log restated in csrc/gsim_hostlog.hpp
 * with the constants of the installed libm.so.6, accepted at build time only if bit-identical to the host's log() on 3e7
 * arguments): the normals are then std::normal_distribution's bit for bit and noise mode 1 reproduces the reference's
 * noisy trajectories exactly at any batch size.  0: the device library's log (normals within 2 ulp, statistical parity). */
int rgs_noise_is_exact(void);
int rgs_selftest_rng(int device_id, long long seed, int n, int n_attempts, double *normals_host, double *canonicals_host,
                     uint32_t *x_end, int *have);

/* The pooled variants (POOL, APOOL) bind 256 arenas to a workgroup for a launch.  Between fused launches (>= 8 steps,
 * after >= 4 pooled steps of history) the library deals arenas that spent at least 7/8 of those steps in the event pass
 * ("hot": lasting contact) evenly over the workgroups by swapping them against cold arenas; results do not depend on
 * it.  enable = 0 switches the dealing off (default on).  rgs_get_rebalance_info: *n_swapped = hot arenas moved by the
 * last deal, perm_host (may be NULL) [B] = the arena each lane slot steps (identity when no deal is active). */
int rgs_set_rebalance(rgs_sim *sim, int enable);
int rgs_get_rebalance_info(rgs_sim *sim, int *n_swapped, int *perm_host);
/* heat_host [B]: per arena, the steps of the pooled launches since the last deal that went through the event pass. */
int rgs_get_heat(rgs_sim *sim, uint32_t *heat_host);

/* Divisions by tau_player / tau_puck use reciprocal sequences that round exactly like IEEE '/'
 * ("Division by tau"): a multiply when tau is a power of two; reciprocal + ONE correction
 * step when the host has certified tau for it (all worst-case dividends enumerated and checked at
 * reset); reciprocal + two correction steps (proven for any tau in the domain) otherwise.
 * level: 0 = IEEE '/' everywhere, 1 = never the one-step form, 2 = everything allowed (default).
 * Taus outside the domain select IEEE automatically.  All levels give identical results. */
int rgs_set_fast_division(rgs_sim *sim, int level);

/* What the host decides for a given tau: *fast_ok (in the reciprocal domain), *pow2, *one_step
 * (certified).  worst_out (may be NULL) receives up to worst_cap of the enumerated worst-case
 * dividends (quotients within 2^-51 ulp of a rounding midpoint), *worst_count their number. */
int rgs_division_info(double tau, int *fast_ok, int *pow2, int *one_step, double *worst_out, int worst_cap,
                      int *worst_count);

/* The lane pass evaluates the disc-pair part of its (conservative) overlap watch every 5th sub-step
 * against radii widened by the largest possible travel in between; enable = 0 evaluates it at every
 * sub-step (slower, same results).  Batches with dt_col/tau > 0.1 use every sub-step automatically. */
int rgs_set_sparse_watch(rgs_sim *sim, int enable);

/* a[i]/tau on the GPU three ways: two-step reciprocal (fast), IEEE '/', one-step reciprocal (one). */
int rgs_selftest_divtau(int device_id, const double *a_host, int64_t n, double tau, double *fast_host,
                        double *ieee_host, double *one_host);

/* Device arithmetic self-test: computes a[i]/b[i] and sqrt(a[i]) on the GPU with the same
 * code paths the stepper uses, for comparison with the host's IEEE results. */
int rgs_selftest_math(int device_id, const double *a_host, const double *b_host, int64_t n,
                      double *quot_host, double *sqrt_host);

#ifdef __cplusplus
}
#endif
#endif /* RGS_HIP_H */
