#!/usr/bin/env python3
"""bench.py -- arena-steps/sec of the MI355X batched robo-game-sim stepper.

    python bench.py --gpus N --steps K --warmup W          (N > 1 starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 65 536 independent 2v2 arenas per GPU,
dt = 0.01, puck starts and per-player circle controls u = A*(cos(t+phi), sin(t+phi)) drawn from
np.random.default_rng(20260); one "step" advances every arena by one GameSim::run.  The control
table for all W+K steps is resident in HBM before the timed region.  Weak scaling: every rank
owns its own 65 536 arenas, no per-step communication, one RCCL all-gather of the final states
at the end of the timed region (N > 1).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      HBM view of the stepping kernel (algorithmic bytes / HIP-event time)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's algorithm) timed on this
                host's cores on a bounded sample of the same workload
  parity        the GPU result of the sampled arenas (taken from the all-gathered ensemble when
                N > 1: arenas of the first AND of the last rank) against that oracle run
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 lanes/2 (fp64 half rate) x 2 flop x 2.4 GHz
PMC_FILE = os.path.join(ROOT, "profiles", "r03", "pmc.json")

# config -> (n per team, arenas per GPU, dt, winning score, default timed steps, default warm-up steps)
PRESETS = {"c2": (2, 65536, 0.01, 1000000, 1000, 10), "c3": (7, 65536, 0.01, 1000000, 1000, 10),
           "c4": (2, 131072, 0.05, 4, 201, 0), "c5": (4, 32768, 0.05, 1000000, 1000, 10)}


def algorithmic_bytes_per_arena_step(n):
    """state read + controls read + state write = 8*(2S + C)."""
    S = 2 + 4 * (2 * n + 1)
    return 8 * (2 * S + 4 * n)


def model_flops_per_arena_step(n):
    """SURVEY.md 8d secondary figure (event-free step)."""
    E = 2 * n + 1
    return 50 * (2 * n * 68 + 76) + 50 * E * (4 + 7 * (E - 1))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps K (default 1000; 201 for c4 = 1 + ceil(T/dt))")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps W (default 10; c4: 0, its 201 steps "
                    "run from reset like py/MonteCarlo.py, the kernels are warmed on a scratch run instead)")
    ap.add_argument("--config", default="c2", choices=sorted(PRESETS),
                    help="BASELINE.json workload: c2 = 65 536 x 2v2 (the headline metric, default); c3 = 65 536 x 7v7; "
                         "c4 = Monte-Carlo 2v2, mc_example_1.py parameters, 131 072 arenas per GPU; c5 = 4v4 with "
                         "per-step host-streamed controls, 32 768 arenas per GPU (also reports the PCIe-inclusive rate)")
    ap.add_argument("--n-per-team", type=int, default=None)
    ap.add_argument("--arenas", type=int, default=None, help="arenas per GPU")
    ap.add_argument("--steps-per-launch", type=int, default=0,
                    help="steps fused into one kernel launch (0 = all K: waves only re-synchronise at launch ends, so "
                         "fewer launches = less tail)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 wave-per-arena, 2 hybrid, 3 two-lane hybrid, 4 pooled hybrid, 5 asynchronous pooled hybrid")
    ap.add_argument("--cpu-sample-arenas", type=int, default=2048)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the cpu_baseline leg (a 1-GPU box's CPU share)")
    ap.add_argument("--cpu-seconds", type=float, default=2.5, help="target duration of each cpu_baseline leg (all threads, "
                    "one thread, reference-faithful RNG); never below 0.5 s of measured work")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--no-streamed", action="store_true", help="c5: skip the host-streamed leg (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a single GPU)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--watch-every", type=int, default=-1, help="diagnostic: pair-watch interval (0 dense, 2, 5, 10)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic: in-kernel phase stamps (slightly perturbs timing)")
    ap.add_argument("--jam", action="store_true", help="diagnostic: drive every player towards the arena centre "
                    "(sustained contact, many 49-event steps): the hybrid's worst case")
    ap.add_argument("--calm", action="store_true", help="diagnostic: shrink all motion so no arena ever has an event")
    ap.add_argument("--arena-offset", type=int, default=0, help="rehearsal / tests: this job's arenas start at this index of the "
                    "global ensemble (c4: run numbers offset+1 ...), so that one process can re-run exactly another job's rank r shard")
    ap.add_argument("--ensemble-arenas", type=int, default=0, help="size of the global ensemble the inputs are drawn for "
                    "(default: arenas x ranks + offset)")
    ap.add_argument("--no-rebalance", action="store_true", help="diagnostic: pooled variants keep every arena on its own workgroup")
    ap.add_argument("--dt", type=float, default=None, help="diagnostic / tools/sweep.py: override the preset's step length")
    ap.add_argument("--noise", type=float, default=0.0,
                    help="w_stdev_ of the players' dynamics (src/GameSim.cpp:66-67; not a BASELINE config, one rank only): the "
                         "reference's generator runs on the device, arena a seeded --noise-seed + a like cpp:41; the sampled arenas "
                         "are checked against the oracle's <random> draws")
    ap.add_argument("--noise-seed", type=int, default=1)
    args = ap.parse_args(argv)
    preset = PRESETS[args.config]
    if args.n_per_team is None:
        args.n_per_team = preset[0]
    if args.arenas is None:
        args.arenas = preset[1]
    args.dt, args.win = (preset[2] if args.dt is None else args.dt), preset[3]
    if args.steps is None:
        args.steps = preset[4]
    if args.warmup is None:
        args.warmup = preset[5]
    if args.config == "c4":
        args.warmup = 0                                        # the 1 + 200 steps run from reset (py/MonteCarlo.py:28-45)
    return args


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) under
    torch.distributed.run and relay rank 0's JSON line.  This process never touches the GPU."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(orc, w, args, n, x0, amp, ph, cores):
    """The CPU oracle timed on this host on a bounded sample of the SAME workload (same generator, same parameters;
    SURVEY 8d / BASELINE.md section 3): independent of --steps, every leg runs for >= 0.5 s.  The sample is the first
    arenas of this rank for max(W+K, 200) steps from reset, controls computed on the host (numpy) -- what the CPU
    computes is the workload, not the GPU's table, so its values need not be the device's to the last ulp (parity
    above uses the device's own table).  Legs: all `cores` threads with the dead RNG draws elided; ONE thread; all
    threads with the reference's 2 normal draws per f_player call (src/GameSim.cpp:66-67).  Each leg's arena count is
    grown from a short calibration run until the leg runs for about `args.cpu_seconds`."""
    P = 2 * n
    steps0 = min(max(args.warmup + args.steps, 200), 1010)
    cap = min(16384, x0.shape[1])                                  # arenas; a longer leg then runs more steps (<= 1010)

    def leg(arena_steps, threads, draw_always):
        arenas = max(1, min(cap, int(arena_steps / steps0)))
        steps = steps0 if arenas < cap else int(min(1010, max(steps0, arena_steps / cap)))
        tab = w.circle_controls(amp[:, :arenas], ph[:, :arenas], w.step_times(steps, args.dt))
        if args.config == "c4":
            tab[0] = 0.0                                           # py/MonteCarlo.py:33
        xs = np.ascontiguousarray(x0[:, :arenas])
        t1 = time.perf_counter()
        orc.batch_run(n, xs, tab, steps, dt=args.dt, winning_score=args.win, n_threads=threads, draw_always=draw_always)
        el = time.perf_counter() - t1
        return (arenas, steps), el, arenas * steps / el

    out = {}
    total = 0.0
    for name, threads, draw in (("all", cores, False), ("one", 1, False), ("faithful", cores, True)):
        shape, el, rate = leg(8 * threads * steps0, threads, draw)  # calibration (also starts the thread pool)
        total += el
        for _ in range(3):                                         # grow the sample until it runs for about cpu_seconds
            if el >= 0.6 * args.cpu_seconds or shape == (cap, 1010):
                break
            shape, el, rate = leg(rate * args.cpu_seconds * 1.1, threads, draw)
            total += el
        out[name] = (shape, el, rate)
    a_all, s_all, r_all = out["all"]
    a_one, s_one, r_one = out["one"]
    a_f, s_f, r_f = out["faithful"]
    return {"value": round(r_all, 1), "unit": "arena-steps/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_string(), "host_cpus": os.cpu_count(),
            "sample": "%d arenas x %d steps of this workload from reset (host-computed controls of the same generator), "
                      "oracle/gamesim_oracle.cpp (g++ -O2 -ffp-contract=off, OpenMP over arenas), dead RNG draws elided"
                      % a_all,
            "sample_seconds": round(s_all, 3),
            "one_thread_value": round(r_one, 1), "one_thread_sample": "%d arenas x %d steps" % a_one,
            "one_thread_seconds": round(s_one, 3),
            "reference_faithful_rng_value": round(r_f, 1),
            "reference_faithful_rng_sample": "%d arenas x %d steps, %d threads, 2 normal draws per f_player call as "
                                             "src/GameSim.cpp:66-67" % (a_f + (cores,)),
            "reference_faithful_rng_seconds": round(s_f, 3), "total_cpu_leg_seconds": round(total, 2)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    DT, WIN = args.dt, args.win
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the stepper has no CPU path")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    on_dev = args.backend == "nccl"

    import robo_game_py as rgp
    from robo_game_py import workloads as w
    from robo_game_py.ensemble import all_gather_states, partition

    n, B, K, W = args.n_per_team, args.arenas, args.steps, args.warmup
    if args.steps_per_launch <= 0:
        args.steps_per_launch = max(K, 1)
    P = 2 * n
    S = rgp.state_size(n)
    dev = torch.device("cuda", local_rank)
    lo, hi = args.arena_offset + rank * B, args.arena_offset + (rank + 1) * B
    if args.noise != 0.0 and world > 1:
        raise SystemExit("bench.py --noise: one rank only (the sampled arenas of the last rank would need their own seed base)")
    total_ensemble = args.ensemble_arenas if args.ensemble_arenas > 0 else B * world + args.arena_offset

    # ---- inputs, resident in HBM before timing ------------------------------------------
    times = w.step_times(W + K, DT)
    if args.config == "c4":
        # py/MonteCarlo.py:23-45 with mc_example_1.py's parameters: run i (1-based) is reset with seed i, which
        # seeds that run's control generator; every run starts from x0_ball = (-1, 0, 0, 0), makes
        # one step with zero commands (MonteCarlo.py:33) and then ceil(T/dt) = 200 commanded steps.
        x0 = np.repeat(np.array([[-1.0], [0.0], [0.0], [0.0]]), B, axis=1)
        amp, ph = w.mc_inputs(np.arange(lo + 1, hi + 1), n)
    else:
        x0_all, amp_all, ph_all = w.c2_inputs(total_ensemble, n)   # one global ensemble, sliced per rank
        if args.calm:
            amp_all = amp_all * 0.05
            x0_all = x0_all * np.array([0.2, 0.2, 0.05, 0.05])[:, None]
        x0, amp, ph = x0_all[:, lo:hi], amp_all[:, lo:hi], ph_all[:, lo:hi]
    stride = P * 2 * B
    table = w.device_circle_table(torch, dev, amp, ph, times, zero_first=(args.config == "c4"))
    if args.jam:                                               # constant pull to the centre, strength from amp
        s0 = rgp.BatchedGameSim(B, n, device=local_rank)
        st0 = torch.from_numpy(s0.reset(DT, WIN, np.ascontiguousarray(x0))).to(dev)
        s0.close()
        amp_d = torch.from_numpy(np.ascontiguousarray(amp)).to(dev)
        for e in range(P):
            d = -st0[6 + 4 * e:8 + 4 * e]                      # spawn position -> centre
            d = d / torch.clamp(torch.sqrt((d * d).sum(0)), min=1e-9)
            table[:, e, :, :] = (amp_d[e] * d)[None]
    torch.cuda.synchronize()

    sim = rgp.BatchedGameSim(B, n, device=local_rank)
    sim.set_variant(args.variant)
    if args.no_rebalance:
        sim.set_rebalance(False)
    if args.watch_every >= 0:
        sim.set_sparse_watch(args.watch_every)
    mine = torch.empty((S, B), dtype=torch.float64, device=dev)

    def run_steps(k_begin, k_count):
        launches = 0
        k = k_begin
        while k < k_begin + k_count:
            m = min(args.steps_per_launch, k_begin + k_count - k)
            sim.step_device(table.data_ptr() + 8 * k * stride, stride, m)
            launches += 1
            k += m
        return launches

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if on_dev else dist.barrier()
        torch.cuda.synchronize()
        sim.sync()

    def reset():
        if args.noise != 0.0:
            sim.reset(DT, WIN, np.ascontiguousarray(x0), args.noise, False, "", args.noise_seed + lo)
        else:
            sim.reset(DT, WIN, np.ascontiguousarray(x0))

    if W == 0:                                                 # no warm-up steps: load the kernels / wake the clocks on a scratch run
        reset()
        run_steps(0, min(K, 8))
        sim.sync()
    reset()
    if args.stamps:
        sim.stamps(True)
        sim.wave_log()                                         # arm the per-wavefront log (asynchronous pool)
    if world > 1:                                             # RCCL channel setup happens on first use: not in the timed region
        all_gather_states(mine if on_dev else mine.cpu(), B * world)
    # ---- warm-up, then the timed region ----------------------------------------------------
    run_steps(0, W)
    barrier()
    if args.stamps:
        sim.stamps(True, clear=True)                 # the timed launches only
    t0 = time.perf_counter()
    sim.timer_start()
    launches = run_steps(W, K)
    kernel_ms = sim.timer_stop()                 # HIP events on the stepper's own stream
    gathered = None
    if world > 1:                                # the job's only collective: final states to every rank
        sim.copy_state_to(mine.data_ptr())
        sim.sync()
        gathered = all_gather_states(mine if on_dev else mine.cpu(), B * world)    # (S, B*world) on every rank
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if on_dev else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_arena_steps = float(B) * world * K
    value = total_arena_steps / elapsed
    counters = sim.counters()
    counters["hot_arenas_dealt_to_other_workgroups"] = sim.rebalance_info()[0]
    if args.stamps:
        counters["stamps"] = sim.stamps(True)
        wl = sim.wave_log()
        if wl[:, 0].any():
            np.save(os.path.join(ROOT, "gpurun_out", "wave_log.npy"), wl)
            wg = wl.reshape(-1, 4, 8)
            tot = wl[:, 0].astype(float)
            counters["wave_log"] = {"waves": int(len(wl)), "total_cycles_mean": float(tot.mean()), "total_cycles_max": float(tot.max()),
                                    "p50": float(np.percentile(tot, 50)), "p90": float(np.percentile(tot, 90)), "p99": float(np.percentile(tot, 99)),
                                    "own_done_mean": float(wl[:, 5].mean()), "own_done_max": float(wl[:, 5].max()),
                                    "lane_passes_mean": float(wl[:, 3].mean()), "lane_passes_max": int(wl[:, 3].max()),
                                    "posted_per_wg_mean": float(wg[:, :, 6].sum(1).mean()), "posted_per_wg_max": int(wg[:, :, 6].sum(1).max()),
                                    "rounds_per_wg_mean": float(wg[:, :, 4].sum(1).mean()), "rounds_per_wg_max": int(wg[:, :, 4].sum(1).max()),
                                    "served_per_wg_mean": float(wg[:, :, 7].sum(1).mean()), "served_per_wg_max": int(wg[:, :, 7].sum(1).max()),
                                    "corr_total_vs_wg_served": float(np.corrcoef(wg[:, :, 0].max(1), wg[:, :, 7].sum(1))[0, 1])}

    # ---- sampled arenas: the first cs0 of rank 0 and (N > 1) the first cs1 of the LAST rank, read from the
    # all-gathered ensemble so that the shard -> arena-id mapping of the collective is part of what is checked
    check = not args.no_cpu
    cs = min(args.cpu_sample_arenas, B)
    cs0 = cs if world == 1 else max(1, cs // 2)
    cs1 = 0 if world == 1 else max(1, cs // 2)
    got = sub = x0s = ids = None
    if check:
        last_lo = partition(B * world, world)[-1][0]
        ids = np.concatenate([np.arange(cs0), last_lo + np.arange(cs1)]).astype(np.int64)
        sub0 = table[:, :, :, :cs0].contiguous()                 # the very values the GPU consumed
        if world > 1:
            sub1 = table[:, :, :, :cs1].contiguous() if rank == world - 1 else torch.empty(
                (W + K, P, 2, cs1), dtype=torch.float64, device=dev)
            x1 = torch.from_numpy(np.ascontiguousarray(x0[:, :cs1])).to(dev) if rank == world - 1 else torch.empty(
                (4, cs1), dtype=torch.float64, device=dev)
            if not on_dev:
                sub1, x1 = sub1.cpu(), x1.cpu()
            dist.broadcast(sub1, src=world - 1)
            dist.broadcast(x1, src=world - 1)
        if rank == 0:
            if world == 1:
                got = sim.state()[:, :cs0].copy()                  # before anything else steps the arenas
                sub, x0s = sub0.cpu().numpy(), np.ascontiguousarray(x0[:, :cs0])
            else:
                got = gathered[:, torch.from_numpy(ids).to(gathered.device)].cpu().numpy()
                sub = np.concatenate([sub0.cpu().numpy(), sub1.cpu().numpy()], axis=3)
                x0s = np.concatenate([x0[:, :cs0], x1.cpu().numpy()], axis=1)

    # ---- C5: the same steps driven from the host, one control upload + one state download per step ----
    streamed = None
    streamed_state = None
    if args.config == "c5" and not args.no_streamed:
        ks = min(W + K, 200)
        ctl_pin, st_pin = sim.pinned((ks, P, 2, B)), sim.pinned((S, B))
        ctl_pin[...] = table[:ks].cpu().numpy()                  # the controller's outputs, already in pinned memory
        sim.reset(DT, WIN, np.ascontiguousarray(x0))             # from reset, so the streamed result can be checked too
        barrier()
        t1 = time.perf_counter()
        for k in range(ks):
            sim.step_roundtrip(ctl_pin[k], st_pin)
        el = time.perf_counter() - t1
        streamed_state = np.array(st_pin[:, :cs0], copy=True)
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev if on_dev else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        streamed = {"value": round(float(B) * world * ks / el, 1), "unit": "arena-steps/s", "steps": ks,
                    "ms_per_step": round(el * 1e3 / ks, 4),
                    "bytes_per_step_per_gpu": {"h2d_controls": P * 2 * B * 8, "d2h_state": S * B * 8},
                    "note": "PCIe-inclusive closed-loop rate from reset: per step one H2D of the controls, one stepping "
                            "launch, one D2H of the state, pinned host buffers; never the headline value"}

    # ---- roofline of the stepping kernel (per launch) ------------------------------------------
    bytes_per_launch = algorithmic_bytes_per_arena_step(n) * B * (K / launches)
    launch_s = kernel_ms * 1e-3 / launches
    achieved = bytes_per_launch / launch_s / 1e9
    # HBM bytes and VALU instructions of this launch shape from the committed rocprofv3 PMC passes (tools/prof.sh:
    # FETCH_SIZE, WRITE_SIZE and the SQ counters in separate --pmc runs of this very command line; KB -> bytes,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only a profile of the same config, arena
    # count, K, W and launch count is used; anything else reports null.
    pmc = None
    pmc_key = "%s:n%d:B%d:K%d:W%d:L%d:v%d" % (args.config, n, B, K, W, launches, args.variant)
    try:
        pmc = json.load(open(PMC_FILE)).get(pmc_key)
    except Exception:
        pmc = None
    traffic = None
    valu_issue = None
    if pmc and not (args.jam or args.calm):
        traffic = round((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0 / 1e9, 4)
        per_wave = pmc["SQ_INSTS_VALU"] / max(1.0, pmc["SQ_WAVES"])
        peak = 1024 * 2.4e9 / 4.0
        rate = pmc["SQ_INSTS_VALU"] / launch_s
        valu_issue = {"achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s",
                      "frac": round(rate / peak, 4), "valu_wave_instr_per_wave": round(per_wave, 1),
                      "source": "SQ_INSTS_VALU of the timed launch, %s key %s" % (os.path.relpath(PMC_FILE, ROOT), pmc_key),
                      "note": "peak = one wave64 instruction per 4 cycles per SIMD; a lone wave per SIMD issues FP64 "
                              "every 5.2-5.8 cycles (profiles/r01/fp64_issue_rate.txt)"}
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "traffic_unit": "GB per launch (PMC of the same command line, %s)" % os.path.relpath(PMC_FILE, ROOT),
                "algorithmic_gb_per_launch": round(bytes_per_launch / 1e9, 4),
                "kernel_ms_per_launch": round(launch_s * 1e3, 4), "launches": launches,
                "bytes_per_arena_step": algorithmic_bytes_per_arena_step(n),
                "note": "path is FP64-VALU/latency bound, not HBM bound; see fp64_model and DESIGN.md"}
    model_tflops = model_flops_per_arena_step(n) * B * K / (kernel_ms * 1e-3) / 1e12
    fp64_model = {"flops_per_arena_step": model_flops_per_arena_step(n), "achieved": round(model_tflops, 4),
                  "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(model_tflops / FP64_VECTOR_PEAK_TFLOPS, 5)}

    # ---- parity of the sampled arenas + CPU baseline (rank 0, every N) ------------------------
    cpu = None
    parity = None
    if rank == 0 and check:
        from oracle import oracle as orc
        nc = cs0 + cs1
        cores = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0))))
        ref = orc.batch_run(n, x0s, sub, W + K, dt=DT, winning_score=WIN, n_threads=cores, noise=args.noise,
                            seed_base=args.noise_seed + lo)
        delta = float(np.max(np.abs(got - ref["state"])))
        import hashlib
        sha = lambda arr: hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
        parity = {"arenas_checked": nc, "steps": W + K, "max_abs_delta_vs_oracle": delta,
                  "first_rank_states_sha256": sha(got[:, :cs0]), "last_rank_states_sha256": sha(got[:, cs0:]) if cs1 else None,
                  "global_arena_ids": "%d..%d" % (lo, lo + cs0 - 1) + ("" if world == 1 else " and %d..%d" % (
                      args.arena_offset + (world - 1) * B, args.arena_offset + (world - 1) * B + cs1 - 1)),
                  "scores_equal": bool(np.array_equal(got[:2], ref["state"][:2])),
                  "bit_identical": bool(np.array_equal(got, ref["state"])),
                  "arena_ids": "0..%d" % (cs0 - 1) + ("" if world == 1 else " (rank 0) and %d..%d (rank %d), read from the "
                                                       "all-gathered ensemble" % (ids[cs0], ids[-1], world - 1))}
        if streamed_state is not None:
            ks = streamed["steps"]
            rs = orc.batch_run(n, np.ascontiguousarray(x0s[:, :cs0]), np.ascontiguousarray(sub[:ks, :, :, :cs0]), ks, dt=DT,
                               winning_score=WIN, n_threads=cores)
            parity["host_streamed_bit_identical"] = bool(np.array_equal(streamed_state, rs["state"]))
            parity["host_streamed_arenas_x_steps"] = "%d x %d" % (cs0, ks)
        cpu = cpu_baseline(orc, w, args, n, x0, amp, ph, cores)

    if rank == 0:
        controls = ("per-run seeds %d..%d drive the control generator, x0_ball=(-1,0,0,0), 1 zero-command step + %d "
                    "commanded steps from reset" % (1, B * world, K - 1)) if args.config == "c4" else "circle controls, rng 20260"
        out = {
            "metric": "arena-steps/sec (whole node), 2v2" if n == 2 else "arena-steps/sec (whole node), %dv%d" % (n, n),
            "value": round(value, 1), "unit": "arena-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed * 1e3 / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d parallel %dv%d arenas per GPU, dt=%g, winning_score=%d, %s"
                                   % (args.config.upper(), B, n, n, DT, WIN, controls)
                                   + ("" if args.noise == 0.0 else ", noise %g (generator on the device, %s)" % (
                                       args.noise, "exact" if rgp.BatchedGameSim.noise_is_exact() else "statistical parity")),
                       "arenas_per_gpu": B, "n_per_team": n, "steps_per_launch": args.steps_per_launch,
                       "variant": args.variant, "parallelism": "arena-sharded x%d, end-of-run all-gather" % world},
            "collective": None if world == 1 else {"backend": args.backend, "ranks": dist.get_world_size(),
                                                   "gathered_shape": list(gathered.shape)},
            "roofline": roofline, "fp64_model": fp64_model, "valu_issue": valu_issue, "cpu_baseline": cpu, "parity": parity,
            "host_streamed": streamed, "events": counters,
        }
        print(json.dumps(out), flush=True)

    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
