"""Read sharding across the GPUs of one node: independent units, no exchange on the data path.

The reference hands hole-batches (50 reads) to call-workers through one shared queue, worker i on GPU i mod n
(call_modifications.py:465-471, 561-578).  With one process per GPU the static equivalent is round-robin by hole-batch
index; the only communication is an end-of-run reduction of counters / timings (torch.distributed: RCCL on GPUs, gloo in
the CPU tests)."""


def shard_indices(n_units, rank, world_size):
    """Indices of the units (hole-batches / site batches) rank `rank` processes: rank, rank + W, rank + 2W, ..."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return range(rank, n_units, world_size)


def reduce_run_stats(local_sites, local_reads, local_seconds, group=None):
    """All ranks -> (total sites, total reads, max seconds).  No-op without an initialised process group."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return local_sites, local_reads, local_seconds
    if not (dist.is_available() and dist.is_initialized()):
        return local_sites, local_reads, local_seconds
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cnt = torch.tensor([local_sites, local_reads], dtype=torch.int64, device=dev)
    sec = torch.tensor([local_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(sec, op=dist.ReduceOp.MAX, group=group)
    return int(cnt[0].item()), int(cnt[1].item()), float(sec[0].item())
