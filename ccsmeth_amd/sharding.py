"""Read sharding across the GPUs of one node: independent units, no exchange on the data path.

The reference hands hole-batches (50 reads) to its call-workers through ONE shared queue, worker i on GPU i mod n
(call_modifications.py:465-471, 561-578), filled by one reader process (extract_features.py:129-177).  Here, with one
process per GPU:
  * ONE rank scans the input once (a background thread with its own reader) and publishes, per hole-batch, where it lies in the
    file (BGZF virtual offsets of its first record and of the byte behind its last), how many records it holds and the running
    site index of its first site (the Philox counter of the initial states: every probability is independent of who computes
    the batch) on a key-value board (torch.distributed.TCPStore);
  * every rank claims the next unclaimed batch index (an atomic counter on the board: the shared queue), SEEKS its own reader
    to the batch and inflates only that range (libccsm_bam: ccsm_bam_seek), so the input is inflated twice in total — once by the
    scan, once by whoever processes a batch — instead of once per rank;
  * `dispatch="static"` hands batch i to rank i mod world instead (deterministic shares, used by the tests).
The only other communication is the end-of-run gather of the output runs and counters (gloo)."""
import struct
import threading

_DESC = struct.Struct("<QQIQ")      # voffset_start, voffset_end, n_reads, site_base
_END = b"END"


def shard_indices(n_units, rank, world_size):
    """Indices of the units rank `rank` processes under static dispatch: rank, rank + W, rank + 2W, ..."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return range(rank, n_units, world_size)


class BatchBoard:
    """Descriptors of consecutive hole-batches on a torch.distributed store + the claim counter."""

    def __init__(self, store, world, rank, dispatch="dynamic", prefix="ccsm"):
        if dispatch not in ("dynamic", "static"):
            raise ValueError("dispatch must be 'dynamic' or 'static'")
        self.store, self.world, self.rank, self.dispatch, self.prefix = store, int(world), int(rank), dispatch, prefix
        self._next_static = self.rank
        self.claimed = []

    def _key(self, i):
        return "%s/b%d" % (self.prefix, i)

    # ---- scanning rank
    def publish(self, i, voffset_start, voffset_end, n_reads, site_base):
        self.store.set(self._key(i), _DESC.pack(voffset_start, voffset_end, n_reads, site_base))

    def finish(self, n_batches):
        """No more batches: every rank reads at most one index behind the end."""
        for k in range(self.world + 1):
            self.store.set(self._key(n_batches + k), _END)

    # ---- every rank
    def claim(self):
        """(index, voffset_start, voffset_end, n_reads, site_base) of the next batch of this rank, or None when the input is done.
        Blocks until the scan has published the claimed index."""
        if self.dispatch == "dynamic":
            i = int(self.store.add("%s/next" % self.prefix, 1)) - 1
        else:
            i = self._next_static
            self._next_static += self.world
        raw = self.store.get(self._key(i))
        if raw == _END:
            return None
        self.claimed.append(i)
        return (i,) + _DESC.unpack(raw)


def scan_hole_batches(reader, holes_batch, sites_of_batch, board, on_error=None):
    """The scanning rank's pass over the whole input: publish every hole-batch, then the end marks.  `sites_of_batch(batch)` =
    CpG sites of the batch that will be called (after the read filters).  Returns (batches, sites)."""
    i, site_base = 0, 0
    try:
        while True:
            b = reader.next_batch(holes_batch)
            if b is None:
                break
            board.publish(i, b.voffset_start, b.voffset_end, b.n_reads, site_base)
            site_base += int(sites_of_batch(b))
            b.close()
            i += 1
    except BaseException as e:      # noqa: BLE001 - the claimers must not wait forever
        if on_error is not None:
            on_error(e)
        raise
    finally:
        board.finish(i)
    return i, site_base


def start_scan_thread(make_reader, holes_batch, sites_of_batch, board):
    """Run scan_hole_batches in a daemon thread with its own reader; returns (thread, result dict: batches, sites, inflated_bytes, error)."""
    res = {}

    def run():
        try:
            with make_reader() as rd:
                res["batches"], res["sites"] = scan_hole_batches(rd, holes_batch, sites_of_batch, board)
                res["inflated_bytes"] = rd.inflated_bytes
        except BaseException as e:      # noqa: BLE001
            res["error"] = e
    th = threading.Thread(target=run, name="ccsm-scan", daemon=True)
    th.start()
    return th, res


def open_board_store(world, rank, port=None, host=None, timeout_s=1800):
    """The board's store: a TCPStore on MASTER_ADDR hosted by rank 0, on CCSM_BOARD_PORT or, by default, a free port that rank 0
    picks and broadcasts through the (already initialised) process group.  Returns (store, connect): `connect()` opens one more
    client connection — a store client serialises its calls, so the scanning thread must not share the connection on which the
    same process's claimer blocks in get() (observed: the scan's set() waits behind that get() forever)."""
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
    store = dist.TCPStore(host, int(port), int(world), is_master=(rank == 0), timeout=tmo, wait_for_workers=True)

    def connect():
        return dist.TCPStore(host, int(port), None, is_master=False, timeout=tmo, wait_for_workers=False)
    return store, connect
