#!/usr/bin/env python3
"""bench.py -- arena-steps/sec of the MI355X batched robo-game-sim stepper.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): 65 536 independent 2v2 arenas per GPU,
dt = 0.01, puck starts and per-player circle controls u = A*(cos(t+phi), sin(t+phi)) drawn from
np.random.default_rng(20260); one "step" advances every arena by one GameSim::run.  The control
table for all W+K steps is resident in HBM before the timed region.  Weak scaling: every rank
owns its own 65 536 arenas, no per-step communication, one RCCL all-gather of the final states
at the end of the timed region (N > 1).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM view of the stepping kernel (algorithmic bytes / HIP-event time)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's algorithm) timed on this
                host's cores on a bounded sample of the same workload, and used to check the
                GPU result of those arenas.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))

ARENAS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 lanes/2 (fp64 half rate) x 2 flop x 2.4 GHz


def algorithmic_bytes_per_arena_step(n):
    """state read + controls read + state write = 8*(2S + C)."""
    S = 2 + 4 * (2 * n + 1)
    return 8 * (2 * S + 4 * n)


def model_flops_per_arena_step(n):
    """SURVEY.md 8d secondary figure (event-free step)."""
    E = 2 * n + 1
    return 50 * (2 * n * 68 + 76) + 50 * E * (4 + 7 * (E - 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps K (default 1000; 201 for c4 = 1 + ceil(T/dt))")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json workload: c2 = 65 536 x 2v2 (the headline metric, default); c3 = 65 536 x 7v7; "
                         "c4 = Monte-Carlo 2v2, mc_example_1.py parameters, 131 072 arenas per GPU; c5 = 4v4 with "
                         "per-step host-streamed controls, 32 768 arenas per GPU (also reports the PCIe-inclusive rate)")
    ap.add_argument("--n-per-team", type=int, default=None)
    ap.add_argument("--arenas", type=int, default=None, help="arenas per GPU")
    ap.add_argument("--steps-per-launch", type=int, default=0,
                    help="steps fused into one kernel launch (0 = all K: waves only re-synchronise at launch ends, so "
                         "fewer launches = less tail)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 wave-per-arena, 2 hybrid")
    ap.add_argument("--cpu-sample-arenas", type=int, default=2048)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the cpu_baseline leg (a 1-GPU box's CPU share)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank path on a single GPU)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--watch-every", type=int, default=-1, help="diagnostic: pair-watch interval (0 dense, 2, 5, 10)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic: in-kernel phase stamps (slightly perturbs timing)")
    ap.add_argument("--jam", action="store_true", help="diagnostic: drive every player towards the arena centre "
                    "(sustained contact, many 49-event steps): the hybrid's worst case")
    ap.add_argument("--calm", action="store_true", help="diagnostic: shrink all motion so no arena ever has an event")
    args = ap.parse_args()
    preset = {"c2": (2, 65536, 0.01, 1000000), "c3": (7, 65536, 0.01, 1000000), "c4": (2, 131072, 0.05, 4),
              "c5": (4, 32768, 0.05, 1000000)}[args.config]
    if args.n_per_team is None:
        args.n_per_team = preset[0]
    if args.arenas is None:
        args.arenas = preset[1]
    DT, WIN = preset[2], preset[3]
    if args.steps is None:
        args.steps = 201 if args.config == "c4" else 1000

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
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

    import robo_game_py as rgp
    from robo_game_py import workloads as w

    n, B, K, W = args.n_per_team, args.arenas, args.steps, args.warmup
    if args.steps_per_launch <= 0:
        args.steps_per_launch = max(K, 1)
    P = 2 * n
    S = rgp.state_size(n)

    # ---- inputs, resident in HBM before timing ------------------------------------------
    x0_all, amp_all, ph_all = w.c2_inputs(B * world, n)          # one global ensemble, sliced per rank
    if args.calm:
        amp_all = amp_all * 0.05
        x0_all = x0_all * np.array([0.2, 0.2, 0.05, 0.05])[:, None]
    lo, hi = rank * B, (rank + 1) * B
    x0, amp, ph = x0_all[:, lo:hi], amp_all[:, lo:hi], ph_all[:, lo:hi]
    times = w.step_times(W + K, DT)
    dev = torch.device("cuda", local_rank)
    stride = P * 2 * B
    table = torch.empty((W + K, P, 2, B), dtype=torch.float64, device=dev)
    amp_d, ph_d = torch.from_numpy(np.ascontiguousarray(amp)).to(dev), torch.from_numpy(np.ascontiguousarray(ph)).to(dev)
    t_d = torch.from_numpy(times).to(dev)
    CH = 64
    for k0 in range(0, W + K, CH):
        arg = t_d[k0:k0 + CH, None, None] + ph_d[None]
        table[k0:k0 + CH, :, 0, :] = amp_d[None] * torch.cos(arg)
        table[k0:k0 + CH, :, 1, :] = amp_d[None] * torch.sin(arg)
    del arg
    if args.jam:                                               # constant pull to the centre, strength from amp
        import robo_game_py as _r
        s0 = _r.BatchedGameSim(B, n, device=local_rank)
        st0 = torch.from_numpy(s0.reset(DT, WIN, np.ascontiguousarray(x0))).to(dev)
        s0.close()
        for e in range(P):
            d = -st0[6 + 4 * e:8 + 4 * e]                    # spawn position -> centre
            d = d / torch.clamp(torch.sqrt((d * d).sum(0)), min=1e-9)
            table[:, e, :, :] = (amp_d[e] * d)[None]
    torch.cuda.synchronize()

    sim = rgp.BatchedGameSim(B, n, device=local_rank)
    sim.set_variant(args.variant)
    if args.watch_every >= 0:
        sim.set_sparse_watch(args.watch_every)
    if args.config == "c4":                                    # mc_example_1.py:13-30: every run starts from x0_ball
        x0 = np.repeat(np.array([[-1.0], [0.0], [0.0], [0.0]]), B, axis=1)
    sim.reset(DT, WIN, np.ascontiguousarray(x0))
    from robo_game_py.ensemble import all_gather_states
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
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()
        sim.sync()

    if args.stamps:
        sim.stamps(True)
    if world > 1:                                             # RCCL channel setup happens on first use: not in the timed region
        src0 = mine if args.backend == "nccl" else mine.cpu()
        all_gather_states(src0, B * world)
    # ---- warm-up, then the timed region ----------------------------------------------------
    run_steps(0, W)
    barrier()
    if args.stamps:
        sim.stamps(True, clear=True)                 # the timed launches only
    t0 = time.perf_counter()
    sim.timer_start()
    launches = run_steps(W, K)
    kernel_ms = sim.timer_stop()                 # HIP events on the stepper's own stream
    if world > 1:                                # the job's only collective: final states to every rank
        sim.copy_state_to(mine.data_ptr())
        sim.sync()
        src = mine if args.backend == "nccl" else mine.cpu()
        gathered = all_gather_states(src, B * world)           # (S, B*world) on every rank
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_arena_steps = float(B) * world * K
    value = total_arena_steps / elapsed
    counters = sim.counters()
    if args.stamps:
        counters["stamps"] = sim.stamps(True)
    sample_state = None
    if rank == 0 and world == 1 and not args.no_cpu:
        sample_state = sim.state()[:, :min(args.cpu_sample_arenas, B)].copy()      # before anything else steps the arenas

    # ---- C5: the same steps driven from the host, one control upload + one state download per step ----
    streamed = None
    if args.config == "c5":
        ks = min(K, 200)
        ctl_pin, st_pin = sim.pinned((ks, P, 2, B)), sim.pinned((S, B))
        ctl_pin[...] = table[W:W + ks].cpu().numpy()             # the controller's outputs, already in pinned memory
        barrier()
        t1 = time.perf_counter()
        for k in range(ks):
            sim.step_roundtrip(ctl_pin[k], st_pin)
        el = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        streamed = {"value": round(float(B) * world * ks / el, 1), "unit": "arena-steps/s", "steps": ks,
                    "ms_per_step": round(el * 1e3 / ks, 4),
                    "bytes_per_step_per_gpu": {"h2d_controls": P * 2 * B * 8, "d2h_state": S * B * 8},
                    "note": "PCIe-inclusive closed-loop rate: per step one H2D of the controls, one stepping launch, one "
                            "D2H of the state, pinned host buffers; never the headline value"}

    # ---- roofline of the stepping kernel (per launch) ------------------------------------------
    bytes_per_launch = algorithmic_bytes_per_arena_step(n) * B * (K / launches)
    launch_s = kernel_ms * 1e-3 / launches
    achieved = bytes_per_launch / launch_s / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this kernel (profiles/r01/hybrid_pmc.json:
    # FETCH_SIZE and WRITE_SIZE in separate --pmc runs, KB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    # prescribes for gfx950), scaled to this launch's arena-steps.  Not collected live.
    traffic = None
    try:
        if n == 2 and args.variant != 1 and args.config == "c2":
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "hybrid_pmc.json")))
            per = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 / (260 * 65536)
            traffic = round(per * B * (K / launches) / 1e9, 4)
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_unit": "GB per launch (PMC, profiles/r01)",
                "algorithmic_gb_per_launch": round(bytes_per_launch / 1e9, 4),
                "kernel_ms_per_launch": round(launch_s * 1e3, 4), "launches": launches,
                "bytes_per_arena_step": algorithmic_bytes_per_arena_step(n),
                "note": "path is FP64-VALU/latency bound, not HBM bound; see fp64_model and DESIGN.md"}
    # VALU issue view (the bound that actually binds): wave-instructions per wave-step from the committed PMC
    # pass (profiles/r01/hybrid_pmc.json, SQ_INSTS_VALU) x measured wave-steps/s, against one fp64 wave64
    # instruction per 4 cycles per SIMD-32 (1024 SIMDs x 2.4 GHz / 4).
    valu_issue = None
    try:
        if n == 2 and args.variant != 1 and args.config == "c2":
            per_wave_step = pmc["_per_arena_step"]["valu_insts_per_wave_step"]
            wave_steps_per_s = (B / 64.0) * K / (kernel_ms * 1e-3)
            peak = 1024 * 2.4e9 / 4.0
            valu_issue = {"achieved": round(per_wave_step * wave_steps_per_s / 1e9, 2), "peak": round(peak / 1e9, 1),
                          "unit": "G wave-instr/s", "frac": round(per_wave_step * wave_steps_per_s / peak, 4),
                          "valu_wave_instr_per_wave_step": round(per_wave_step, 1),
                          "source": "SQ_INSTS_VALU from profiles/r01/hybrid_pmc.json (not collected live)",
                          "note": "peak = one wave64 instruction per 4 cycles per SIMD; a lone wave per SIMD issues FP64 "
                                  "every 5.2-5.8 cycles (profiles/r01/fp64_issue_rate.txt)"}
    except Exception:
        valu_issue = None
    model_tflops = model_flops_per_arena_step(n) * B * K / (kernel_ms * 1e-3) / 1e12
    fp64_model = {"flops_per_arena_step": model_flops_per_arena_step(n), "achieved": round(model_tflops, 4),
                  "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(model_tflops / FP64_VECTOR_PEAK_TFLOPS, 5)}

    # ---- CPU baseline + parity of the sampled arenas (rank 0, N = 1 only) ---------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as orc
        cs = min(args.cpu_sample_arenas, B)
        cores = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0))))
        sub = table[:, :, :, :cs].contiguous().cpu().numpy()     # the very values the GPU consumed
        x0s = np.ascontiguousarray(x0[:, :cs])
        t1 = time.perf_counter()
        ref = orc.batch_run(n, x0s, sub, W + K, dt=DT, winning_score=WIN, n_threads=cores)
        cpu_s = time.perf_counter() - t1
        cf = max(1, cs // 4)
        t1 = time.perf_counter()
        orc.batch_run(n, np.ascontiguousarray(x0s[:, :cf]), np.ascontiguousarray(sub[:, :, :, :cf]), W + K, dt=DT,
                      winning_score=WIN, n_threads=cores, draw_always=True)
        cpu_faithful_s = time.perf_counter() - t1
        got = sample_state
        delta = float(np.max(np.abs(got - ref["state"])))
        parity = {"arenas_checked": cs, "steps": W + K, "max_abs_delta_vs_oracle": delta,
                  "scores_equal": bool(np.array_equal(got[:2], ref["state"][:2])),
                  "bit_identical": bool(np.array_equal(got, ref["state"]))}
        cpu = {"value": round(cs * (W + K) / cpu_s, 1), "unit": "arena-steps/s", "cores": cores, "kind": "port",
               "sample": "first %d arenas x %d steps of this workload, oracle/gamesim_oracle.cpp (g++ -O2 "
                         "-ffp-contract=off, OpenMP over arenas), dead RNG draws elided" % (cs, W + K),
               "reference_faithful_rng_value": round(cf * (W + K) / cpu_faithful_s, 1),
               "reference_faithful_rng_sample": "%d arenas x %d steps, 2 normal draws per f_player call as "
                                                "src/GameSim.cpp:66-67" % (cf, W + K)}

    if rank == 0:
        out = {
            "metric": "arena-steps/sec (whole node), 2v2" if n == 2 else "arena-steps/sec (whole node), %dv%d" % (n, n),
            "value": round(value, 1), "unit": "arena-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed * 1e3 / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d parallel %dv%d arenas per GPU, dt=%g, winning_score=%d, circle controls, rng 20260"
                                   % (args.config.upper(), B, n, n, DT, WIN),
                       "arenas_per_gpu": B, "n_per_team": n, "steps_per_launch": args.steps_per_launch,
                       "variant": args.variant, "parallelism": "arena-sharded x%d, end-of-run all-gather" % world},
            "roofline": roofline, "fp64_model": fp64_model, "valu_issue": valu_issue, "cpu_baseline": cpu, "parity": parity, "host_streamed": streamed,
            "events": counters,
        }
        print(json.dumps(out), flush=True)

    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
