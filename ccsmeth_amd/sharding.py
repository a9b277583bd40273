"""Read sharding across the GPUs of one node: independent units, no exchange on the data path.

The reference hands hole-batches (50 reads) to its call-workers through ONE shared queue, worker i on GPU i mod n
(call_modifications.py:465-471, 561-578), filled by one reader process that inflates the whole input
(extract_features.py:129-177).  Here, with one process per GPU, the queue is a counter and nobody reads the file for anybody else:
  * the input is cut into chunks of `chunk_bytes` COMPRESSED bytes; chunk k = the records that start in a BGZF block whose file
    offset lies in [k * chunk_bytes, (k + 1) * chunk_bytes);
  * every rank claims the next unclaimed chunk number (an atomic counter on a torch.distributed.TCPStore: the shared queue;
    `dispatch="static"`: chunk k goes to rank k mod world, deterministic shares for the tests), finds the chunk's first record
    itself (libccsm_bam: ccsm_bam_seek_chunk) and inflates only its own chunk: the input is inflated exactly once in total;
  * a site's device-drawn initial states are keyed by (hash of the read name, position of the C in the read), so every
    probability is independent of who computes the read, of the chunking and of the batching;
  * record-start detection inside a BGZF stream is a heuristic, so the ranks' reports are chained at the end: where chunk k's last
    record ended must be where the next non-empty chunk was found to begin (verify_chain) - any miss is an error, never a silently
    dropped or duplicated read.
The only other communication is the end-of-run gather of the output runs, index tables and counters and two barriers around the
stitch, all on the same store (ChunkQueue.rendezvous: a failing rank releases the waiting ones with its error)."""


def shard_indices(n_units, rank, world_size):
    """Indices of the units rank `rank` processes under static dispatch: rank, rank + W, rank + 2W, ..."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return range(rank, n_units, world_size)


def n_chunks_of(file_size, chunk_bytes):
    if chunk_bytes <= 0:
        raise ValueError("chunk size must be positive")
    return max(1, -(-int(file_size) // int(chunk_bytes)))


def _to_wire(obj):
    """json.dumps hook of ChunkQueue.rendezvous: run tables travel as plain data (bamnative.IndexRun.to_wire), NumPy scalars as numbers."""
    if hasattr(obj, "to_wire"):
        return obj.to_wire()
    if hasattr(obj, "item") and getattr(obj, "shape", None) == ():
        return obj.item()
    raise TypeError("call_mods: %r does not travel between ranks" % type(obj).__name__)


def _from_wire(d):
    if "__index_run__" in d:
        from .bamnative import IndexRun
        return IndexRun.from_wire(d)
    return d


class ChunkQueue:
    """The shared work queue: chunk numbers 0 .. n_chunks - 1, each handed out once."""

    def __init__(self, store, world, rank, n_chunks, dispatch="dynamic", prefix="ccsm"):
        if dispatch not in ("dynamic", "static"):
            raise ValueError("dispatch must be 'dynamic' or 'static'")
        self.store, self.world, self.rank, self.n_chunks, self.dispatch, self.prefix = store, int(world), int(rank), int(n_chunks), dispatch, prefix
        self._static = iter(shard_indices(self.n_chunks, self.rank, self.world))
        self.claimed = []

    def claim(self):
        """Next chunk number of this rank, or None when the input is handed out.  Raises when another rank has reported a failure
        (so that nobody keeps working towards a gather that will never complete)."""
        self.check()
        if self.dispatch == "dynamic":
            k = int(self.store.add("%s/next" % self.prefix, 1)) - 1
        else:
            k = next(self._static, self.n_chunks)
        if k >= self.n_chunks:
            return None
        self.claimed.append(k)
        return k

    def fail(self, message):
        """Tell the other ranks that this one is giving up."""
        try:
            # first failure wins: a rank that fails BECAUSE another one did (check() / rendezvous() raised the published error in it)
            # must not replace the cause with its own echo of it
            if self.store.add("%s/error_count" % self.prefix, 1) == 1:
                self.store.set("%s/error" % self.prefix, ("rank %d: %s" % (self.rank, message)).encode("utf-8", "replace"))
        except Exception:   # noqa: BLE001 - the store may be what failed
            pass

    def check(self):
        if self.store.check(["%s/error" % self.prefix]):
            raise RuntimeError("call_mods aborted: " + self.store.get("%s/error" % self.prefix).decode("utf-8", "replace"))

    def rendezvous(self, tag, payload=None, poll_s=0.02, timeout_s=1800.0):
        """Barrier + gather on the store, with the error key polled while waiting: every rank publishes `payload` under `tag` and gets
        the list of all ranks' payloads once all are there.  A rank that has called fail() releases the others at once with the
        error (a collective all_gather_object / barrier would keep them until the process group's timeout, 30 min by default)."""
        import json
        import time
        self.check()
        # JSON, not pickle: the payloads are numbers, strings and lists, and nothing read from a TCP store should be executable
        self.store.set("%s/%s/%d" % (self.prefix, tag, self.rank), json.dumps(payload, default=_to_wire).encode("utf-8"))
        keys = ["%s/%s/%d" % (self.prefix, tag, r) for r in range(self.world)]
        t0 = time.time()
        while not self.store.check(keys):
            self.check()
            if time.time() - t0 > timeout_s:
                raise RuntimeError("call_mods: rank %d waited %.0f s at '%s' for ranks that never arrived" % (self.rank, timeout_s, tag))
            time.sleep(poll_s)
        self.check()
        return [json.loads(self.store.get(k).decode("utf-8"), object_hook=_from_wire) for k in keys]



def device_identity(index=None):
    """What names the PHYSICAL device behind this rank's cuda index: its UUID where the runtime exposes one, else PCI domain:bus:device.
    N ranks are N GPUs only if these differ (ranks can share a device: CUDA/HIP_VISIBLE_DEVICES, a launcher that maps every rank to cuda:0)."""
    import torch
    if not torch.cuda.is_available():
        return None                                 # (CPU stand-ins of the tests: no device to name)
    i = torch.cuda.current_device() if index is None else index
    pr = torch.cuda.get_device_properties(i)
    # both names where both exist: two ranks share a device only if EVERYTHING the runtime says about theirs agrees (logical partitions of
    # one package must not be refused as "one device" because one of the two names happens to be the package's)
    parts = []
    u = getattr(pr, "uuid", None)
    if u is not None and str(u).strip("0-") != "":
        parts.append("uuid:%s" % u)
    ids = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if all(v is not None for v in ids):
        parts.append("pci:%04x:%02x:%02x" % tuple(int(v) for v in ids))
    if parts:
        return " ".join(parts)
    return "index:%d@%s" % (i, __import__("socket").gethostname())      # no identity available: distinct indices on one host count as distinct


def device_census(identities, n_expected):
    """-> dict(ranks_seen, distinct_devices, ok): `identities` = one device_identity() per rank (gathered); ok iff every one of the
    n_expected ranks reported and they sit on n_expected DISTINCT devices.  The caller refuses to report an N-GPU number when not ok."""
    ids = [str(x) for x in identities if x]
    distinct = len(set(ids))
    return dict(ranks_seen=len(ids), distinct_devices=distinct, ok=(len(ids) == n_expected and distinct == n_expected), devices=sorted(set(ids)))


def collective_library():
    """The collective library torch.distributed's "nccl" backend is bound to on this build (RCCL on ROCm), as a string for reports."""
    try:
        import torch
        v = torch.cuda.nccl.version()
        name = "rccl" if getattr(torch.version, "hip", None) else "nccl"
        return "%s %s" % (name, ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v))
    except Exception as e:      # noqa: BLE001
        return "unavailable (%s)" % type(e).__name__

def verify_chain(first_voffset, chunks, n_chunks=None, eof_voffset=None):
    """chunks: [(k, voffset of the chunk's first record, voffset behind its last record, n_records)] of every rank, empty chunks with
    n_records == 0.  The first non-empty chunk must begin at the file's first record and every later one where its predecessor ended;
    with n_chunks every chunk number 0 .. n_chunks - 1 must have been reported exactly once, and with eof_voffset (NativeBamReader.
    eof_voffset: where a reader stands behind the file's last record) the chain must END there - a last chunk whose record search ran
    off the end of a damaged file reports "empty", and only the end of the chain shows that the file's tail was never read (the
    sequential reader raises "truncated BAM record" on such a file; so does this).  Returns the total number of records."""
    prev_end, total, seen = int(first_voffset), 0, set()
    for k, v0, v1, n in sorted(chunks):
        if k in seen:
            raise RuntimeError("chunk %d was processed twice" % k)
        seen.add(k)
        if n == 0:
            continue
        if int(v0) != prev_end:
            raise RuntimeError("chunk %d begins at virtual offset %#x but the records before it end at %#x: record-start detection "
                               "failed on this input (re-run with one GPU, or report the file)" % (k, int(v0), prev_end))
        prev_end = int(v1)
        total += int(n)
    if n_chunks is not None:
        missing = sorted(set(range(int(n_chunks))) - seen)
        if missing or len(seen) != int(n_chunks):
            raise RuntimeError("chunks %s of %d were never reported (a rank stopped early?)" % (missing[:8], int(n_chunks)))
    if eof_voffset is not None and prev_end != int(eof_voffset):
        raise RuntimeError("the records read end at virtual offset %#x but the file's data ends at %#x: truncated BAM record at the end "
                           "of the input (or record-start detection failed in the last chunk)" % (prev_end, int(eof_voffset)))
    return total


def open_board_store(world, rank, port=None, host=None, timeout_s=1800):
    """The queue's store: a TCPStore on MASTER_ADDR hosted by rank 0, on CCSM_BOARD_PORT or, by default, a free port that rank 0
    picks and broadcasts through the (already initialised) process group."""
    import datetime
    import os
    import socket
    import torch.distributed as dist
    host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None and os.environ.get("CCSM_BOARD_PORT"):
        port = int(os.environ["CCSM_BOARD_PORT"])
    if port is None:
        box = [None]
        if rank == 0:
            with socket.socket() as s:
                s.bind((host if host not in ("localhost",) else "127.0.0.1", 0))
                box[0] = s.getsockname()[1]
        dist.broadcast_object_list(box, src=0)
        port = box[0]
    tmo = datetime.timedelta(seconds=timeout_s)
    return dist.TCPStore(host, int(port), int(world), is_master=(rank == 0), timeout=tmo, wait_for_workers=True)
