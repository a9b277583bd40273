"""N>1 path on CPU: world_size 2, gloo — shard assignment covers every unit exactly once and the end-of-run
reduction (sum of counters, max of time) is what bench.py / a sharded call_mods run rely on."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ccsmeth_amd.sharding import reduce_run_stats, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = list(shard_indices(n_units, rank, world))
    sites = sum(100 + u for u in mine)          # unit u carries 100 + u sites
    total_sites, total_reads, tmax = reduce_run_stats(sites, len(mine), 1.0 + rank)
    dist.barrier()
    q.put((rank, mine, total_sites, total_reads, tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world, n_units = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted(u for _, mine, *_ in res for u in mine)
    assert covered == list(range(n_units))
    exp_sites = sum(100 + u for u in range(n_units))
    for _, _, ts, tr, tmax in res:
        assert ts == exp_sites and tr == n_units and tmax == 2.0


def test_shard_indices_edges():
    assert list(shard_indices(0, 0, 8)) == []
    assert list(shard_indices(3, 2, 8)) == [2]
    assert list(shard_indices(3, 5, 8)) == []
    assert sorted(sum((list(shard_indices(17, r, 8)) for r in range(8)), [])) == list(range(17))
