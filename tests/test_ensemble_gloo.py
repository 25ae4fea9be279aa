"""Multi-rank path on CPU: two gloo ranks each step their own arena shard and all-gather the final
states; the result must equal the single-process run arena for arena.  The stepping itself is
done by the oracle here (no GPU in this test); what is under test is the product's partition +
all-gather code (robo_game_py/ensemble.py), which bench.py uses unchanged with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_arenas, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "robo-game-sim_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from robo_game_py import workloads as w
    from robo_game_py.ensemble import all_gather_states, partition, shard_columns
    n, K, dt = 2, 120, 0.05
    x0, amp, ph = w.c2_inputs(n_arenas, n, seed=17)
    table = w.circle_controls(amp * 2.0, ph, w.step_times(K, dt))
    lo, hi = partition(n_arenas, world)[rank]
    local = orc.batch_run(n, shard_columns(x0, lo, hi), shard_columns(table, lo, hi), K, dt=dt, winning_score=3)
    full = all_gather_states(torch.from_numpy(local["state"]), n_arenas)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_arenas", [64, 37])
def test_two_ranks_equal_one(oracle, tmp_path, n_arenas):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_arenas, str(tmp_path)), nprocs=world, join=True)
    from robo_game_py import workloads as w
    n, K, dt = 2, 120, 0.05
    x0, amp, ph = w.c2_inputs(n_arenas, n, seed=17)
    table = w.circle_controls(amp * 2.0, ph, w.step_times(K, dt))
    single = oracle.batch_run(n, x0, table, K, dt=dt, winning_score=3)["state"]
    for r in range(world):
        got = np.load(str(tmp_path / ("rank%d.npy" % r)))
        assert got.shape == single.shape
        assert np.array_equal(got, single), "rank %d's gathered ensemble differs from the single-process run" % r
