"""The sharded path with the real HIP stepper in every rank: two processes share the one GPU of the
test box (gloo carries the all-gather; RCCL cannot put two ranks on one device) and must reproduce
the single-process GPU run arena for arena."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(n_total):
    from robo_game_py import workloads as w
    x0, amp, ph = w.c2_inputs(n_total, 2, seed=23)
    return x0, w.circle_controls(amp * 2.0, ph, w.step_times(90, 0.05))


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robo_game_py.ensemble import ShardedEnsemble
    x0, table = _inputs(n_total)
    ens = ShardedEnsemble(n_total, 2, device=0)
    ens.reset(0.05, 3, x0)
    ens.run_table(table)
    full = ens.gather_states(on_device=False)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    ens.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpu_ranks_equal_one(rgp, tmp_path):
    n_total, world = 3000, 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    x0, table = _inputs(n_total)
    sim = rgp.BatchedGameSim(n_total, 2)
    sim.reset(0.05, 3, x0)
    single = sim.run_table(table)
    sim.close()
    for r in range(world):
        got = np.load(str(tmp_path / ("rank%d.npy" % r)))
        assert np.array_equal(got, single), "rank %d's gathered ensemble differs from the single-process GPU run" % r


def _worker_rccl(rank, world, port, n_total, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from robo_game_py.ensemble import ShardedEnsemble
    x0, table = _inputs(n_total)
    ens = ShardedEnsemble(n_total, 2)                       # device = LOCAL_RANK
    ens.reset(0.05, 3, x0)
    ens.run_table(table)
    full = ens.gather_states(on_device=True)                # RCCL all_gather_into_tensor of device-resident shards
    np.save(os.path.join(out_dir, "rccl_rank%d.npy" % rank), full.cpu().numpy())
    ens.close()
    dist.barrier(device_ids=[rank])
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL all-gather needs one GPU per rank (the test box has one)")
def test_two_ranks_one_gpu_each_over_rccl(rgp, tmp_path):
    """The production N > 1 path: one process per GPU, shards gathered on the device over RCCL."""
    n_total, world = 3000, 2
    mp.spawn(_worker_rccl, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    x0, table = _inputs(n_total)
    sim = rgp.BatchedGameSim(n_total, 2)
    sim.reset(0.05, 3, x0)
    single = sim.run_table(table)
    sim.close()
    for r in range(world):
        got = np.load(str(tmp_path / ("rccl_rank%d.npy" % r)))
        assert np.array_equal(got, single), "rank %d's RCCL-gathered ensemble differs from the single-process GPU run" % r
