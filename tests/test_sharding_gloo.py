"""N > 1 path of `call_mods` on CPU (gloo, no GPU): the chunk queue of ccsmeth_amd/sharding.py — every rank claims chunk numbers,
finds its chunks' first records itself and inflates only its own chunks — with the model replaced by a stand-in whose
"probabilities" are a function of (read-name hash, position of the C) only (what the Philox counter of the real initial states is):
the stitched 8-rank output must be identical to the single-process output, the index must equal the one made by re-reading the file,
and the ranks together must inflate the file once."""
import gzip
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from ccsmeth_amd.extract_features import motif_locs_cg
from ccsmeth_amd.sharding import shard_indices

ARGV = ["ccsmeth_amd", "call_mods", "--test"]      # the @PG line quotes sys.argv: pinned so that both runs write the same header


class StubPipe:
    """CallModsPipeline's native-batch interface without a GPU: site locations from the sequence, prob = f(read key, location)."""

    def run_native_batch(self, batch, skip=None):
        nr = batch.n_reads
        cnt = np.where(batch.length > 0, batch.n_sites, 0).astype(np.int64)
        if skip is not None:
            cnt[np.asarray(skip, bool)] = 0
        first = np.zeros(nr + 1, np.int32)
        np.cumsum(cnt, out=first[1:])
        locs = np.empty(int(first[-1]), np.int32)
        for r in np.flatnonzero(cnt > 0):
            n = int(batch.length[r])
            sb = batch.seq[int(batch.offset[r]):int(batch.offset[r]) + n]
            lc = motif_locs_cg(sb)
            rl = n - 1 - (lc + 1)
            lc = lc[(lc >= 10) & (lc < n - 10) & (rl >= 10) & (rl < n - 10)]
            assert len(lc) == cnt[r]
            locs[first[r]:first[r + 1]] = lc
        key = np.repeat(batch.name_hash.astype(np.uint64), cnt)
        with np.errstate(over="ignore"):
            idx = ((key ^ (key >> np.uint64(29))) + locs.astype(np.uint64)) * np.uint64(2654435761) % np.uint64(2 ** 32)
        prob1 = np.round(idx.astype(np.float64) / 2 ** 32, 6).astype(np.float32)
        return first, locs, prob1, (cnt > 0).astype(np.uint8), 0

    def close(self):
        pass


def _args(inp, out, dispatch="dynamic"):
    from ccsmeth_amd.call_mods import build_parser
    return build_parser().parse_args(["-i", inp, "-m", "unused.ckpt", "-o", out, "--holes_batch", "10", "--threads", "2",
                                      "--dispatch", dispatch, "--chunk_mb", "0.08"])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, ports, inp, out, dispatch, q):
    from ccsmeth_amd.call_mods import call_mods
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[0]),
                      **({"CCSM_BOARD_PORT": str(ports[1])} if dispatch == "static" else {}))     # both ways of choosing the store's port
    sys.argv = list(ARGV)
    res = call_mods(_args(inp, out, dispatch), log=open(os.devnull, "w"), pipe=StubPipe())
    q.put((rank, res))


def _payload(path):
    """inflated BAM stream (BGZF members concatenated): header and records, independent of the block boundaries"""
    with open(path, "rb") as fh:
        return gzip.decompress(fh.read())


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    from ccsmeth_amd.utils.benchdata import write_synthetic_hifi_bam
    d = tmp_path_factory.mktemp("shard")
    path = str(d / "in.bam")
    write_synthetic_hifi_bam(path, n_reads=203, read_len=6000, cpg=0.02, seed=11)
    return str(d), path


@pytest.mark.parametrize("dispatch", ["static", "dynamic"])
def test_eight_ranks_equal_single_process(bam, dispatch, monkeypatch):
    from ccsmeth_amd.call_mods import call_mods
    d, inp = bam
    monkeypatch.setattr(sys, "argv", list(ARGV))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = call_mods(_args(inp, os.path.join(d, "one")), log=open(os.devnull, "w"), pipe=StubPipe())
    assert one["reads"] == 203 and one["tagged"] > 150
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = (_free_port(), _free_port())
    out = os.path.join(d, "eight_" + dispatch)
    procs = [ctx.Process(target=_worker, args=(r, world, ports, inp, out, dispatch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = res[0]
    assert (r0["reads"], r0["tagged"], r0["failed"]) == (one["reads"], one["tagged"], one["failed"])
    assert _payload(r0["output"]) == _payload(one["output"])              # every record, tag and probability byte
    # the index came from the writers' run tables in both runs: equal to each other and to a fresh pass over the stitched file
    from ccsmeth_amd import bamnative
    assert open(r0["output"] + ".bai", "rb").read() == open(one["output"] + ".bai", "rb").read()
    ok, n = bamnative.index_build(r0["output"], r0["output"] + ".check.bai", threads=2)
    assert ok and n == 203 and open(r0["output"] + ".check.bai", "rb").read() == open(r0["output"] + ".bai", "rb").read()
    chunk = int(0.08 * (1 << 20))
    n_chunks = -(-os.path.getsize(inp) // chunk)
    assert n_chunks > 30 and 0 < sum(r0["rank_chunks"]) <= n_chunks
    total = len(_payload(inp))                                             # inflated size of the input
    slack = 2 * 65536 + 2 * 6000 * 6                                       # per chunk: a block either side + the record that overhangs
    shares = r0["rank_inflated_bytes"]
    assert total * 0.99 <= sum(shares) <= total + n_chunks * slack        # the ranks together inflate the file once; nobody scans
    if dispatch == "static":
        mine = [len(shard_indices(n_chunks, r, world)) for r in range(world)]
        for r in range(world):                                             # a rank's share of the file + epsilon
            assert shares[r] <= total * mine[r] / n_chunks * 1.25 + mine[r] * slack


def test_two_ranks_many_small_batches_per_block(tmp_path, monkeypatch):
    """Reads much smaller than a BGZF block and chunks of a few hundred bytes: most chunks hold no record start, the rest a handful of
    records that end in later chunks' blocks."""
    from ccsmeth_amd import bamio
    from ccsmeth_amd.call_mods import build_parser, call_mods
    rng = np.random.default_rng(77)
    inp = str(tmp_path / "small.bam")
    with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i in range(23):
            n = int(rng.integers(200, 3000))
            seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n)
            pos = rng.integers(0, n - 1, n // 40)
            seq[pos], seq[pos + 1] = ord("C"), ord("G")
            kin = lambda: rng.integers(0, 256, n).astype(np.uint8)  # noqa: E731
            tags = [("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 11)]
            w.write(bamio.BamRecord("z/%d/ccs" % i, flag=16 if i % 5 == 0 else 4, seq=seq.tobytes().decode(), tags=tags[1:] if i == 7 else tags))
    monkeypatch.setattr(sys, "argv", list(ARGV))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    mk = lambda out: build_parser().parse_args(["-i", inp, "-m", "x", "-o", out, "--holes_batch", "3", "--threads", "2", "--no_sort",  # noqa: E731
                                                "--chunk_mb", "0.0005"])
    one = call_mods(mk(str(tmp_path / "one")), log=open(os.devnull, "w"), pipe=StubPipe())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = (_free_port(), _free_port())
    procs = [ctx.Process(target=_worker_small, args=(r, 2, ports, inp, str(tmp_path / "two"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert _payload(res[0]["output"]) == _payload(one["output"])
    assert sum(res[0]["rank_chunks"]) >= 2 and res[0]["sites"] == one["sites"] and res[0]["reads"] == 23


class FakeProbeModel:
    """What call_mods needs of DeviceModel for the probe on its own input: rank 0's probe "finds" split-mx unclean on the input."""

    def __init__(self):
        self.auto_precision, self.precision = True, 4
        self.data_probe_error, self.data_probe_q999, self.data_probe_sites = -1.0, -1.0, 0
        self.probed, self.told = 0, []

    def data_probe(self, run, max_sites=65536):
        self.probed = sum(len(p) for p in run())
        self.data_probe_error, self.data_probe_q999, self.data_probe_sites = 3e-5, 1e-5, self.probed
        self.precision = 3
        return 3

    def set_precision(self, p):
        self.told.append(p)
        self.precision = p


class StubPipeWithProbe(StubPipe):
    def __init__(self):
        self.data_probe_model = FakeProbeModel()

    def probs_of_native_batch(self, batch, skip=None):
        _, _, p1, _, _ = self.run_native_batch(batch, skip)
        return np.stack([1 - p1, p1], 1).astype(np.float32)


def _worker_small(rank, world, ports, inp, out, q, with_probe=False):
    from ccsmeth_amd.call_mods import build_parser, call_mods
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[0]))
    sys.argv = list(ARGV)
    a = build_parser().parse_args(["-i", inp, "-m", "x", "-o", out, "--holes_batch", "3", "--threads", "2", "--no_sort", "--chunk_mb", "0.0005"])
    pipe = StubPipeWithProbe() if with_probe else StubPipe()
    res = call_mods(a, log=open(os.devnull, "w"), pipe=pipe)
    if with_probe:
        m = pipe.data_probe_model
        res = dict(res or {}, probe=(m.precision, m.probed, list(m.told)))
    q.put((rank, res))


def test_data_probe_verdict_travels_from_rank_0_to_the_other_ranks(tmp_path, monkeypatch):
    """The probe of the arithmetic on the input itself (call_mods) under several ranks: rank 0 alone reads the head of
    the file and decides; the verdict reaches every other rank over the chunk board BEFORE any of them calls a site - one arithmetic for
    the whole run, so the output does not depend on the sharding."""
    from ccsmeth_amd import bamio
    rng = np.random.default_rng(78)
    inp = str(tmp_path / "p.bam")
    with bamio.BamWriter(inp, "@HD\tVN:1.5\tSO:unknown\n", []) as w:
        for i in range(17):
            n = int(rng.integers(400, 2500))
            seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n)
            pos = rng.integers(0, n - 1, n // 40)
            seq[pos], seq[pos + 1] = ord("C"), ord("G")
            kin = lambda: rng.integers(0, 256, n).astype(np.uint8)  # noqa: E731
            w.write(bamio.BamRecord("y/%d/ccs" % i, flag=4, seq=seq.tobytes().decode(),
                                    tags=[("fi", "BC", kin()), ("fp", "BC", kin()), ("ri", "BC", kin()), ("rp", "BC", kin()), ("fn", "C", 9), ("rn", "C", 11)]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = (_free_port(), _free_port())
    procs = [ctx.Process(target=_worker_small, args=(r, 3, ports, inp, str(tmp_path / "three"), q, True)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["probe"][0] == 3 and res[0]["probe"][1] > 0 and res[0]["probe"][2] == []          # rank 0 probed (every called site of the head) and decided
    for r in (1, 2):
        assert res[r]["probe"] == (3, 0, [3])                                                      # the others probed nothing and were told
    assert res[0]["reads"] == 17


def test_shard_indices_edges():
    assert list(shard_indices(0, 0, 8)) == []
    assert list(shard_indices(3, 2, 8)) == [2]
    assert list(shard_indices(3, 5, 8)) == []
    assert sorted(sum((list(shard_indices(17, r, 8)) for r in range(8)), [])) == list(range(17))


def test_bench_refuses_more_gpus_than_visible():
    """python bench.py --gpus 2 without torch.distributed.run must not fall back to fewer devices (here: none)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "6"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "refusing" in p.stderr and "{" not in p.stdout


def test_device_census_refuses_shared_devices():
    """N ranks count as N GPUs only on N distinct devices (bench.py refuses its line, rc 2, otherwise; call_mods reports both numbers)."""
    from ccsmeth_amd import sharding
    ok = sharding.device_census(["uuid:a", "uuid:b", "uuid:c", "uuid:d"], 4)
    assert ok["ok"] and ok["ranks_seen"] == 4 and ok["distinct_devices"] == 4
    shared = sharding.device_census(["uuid:a", "uuid:a", "uuid:b", "uuid:b"], 4)
    assert not shared["ok"] and shared["ranks_seen"] == 4 and shared["distinct_devices"] == 2
    missing = sharding.device_census(["uuid:a", None, "uuid:b"], 4)
    assert not missing["ok"] and missing["ranks_seen"] == 2
    assert sharding.device_census(["pci:0000:05:00"], 1)["ok"]
    assert isinstance(sharding.collective_library(), str)


def test_device_identity_uses_every_name_the_runtime_gives(monkeypatch):
    """UUID and PCI address together: ranks share a device only if both agree (two logical partitions that carry their package's UUID but sit
    behind different functions are two devices; one device seen twice is one); with neither, distinct indices on one host count as distinct."""
    import types
    import torch
    from ccsmeth_amd import sharding
    props = {0: types.SimpleNamespace(uuid="aa-bb", pci_domain_id=0, pci_bus_id=5, pci_device_id=0),
             1: types.SimpleNamespace(uuid="aa-bb", pci_domain_id=0, pci_bus_id=6, pci_device_id=0),
             2: types.SimpleNamespace(uuid="00000000-0000", pci_domain_id=0, pci_bus_id=7, pci_device_id=0),
             3: types.SimpleNamespace()}
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props[i])
    ids = [sharding.device_identity(i) for i in range(4)]
    assert ids[0] == "uuid:aa-bb pci:0000:05:00" and ids[1] == "uuid:aa-bb pci:0000:06:00"
    assert ids[2] == "pci:0000:07:00"                                   # an all-zero UUID names nothing
    assert ids[3].startswith("index:3@")
    assert sharding.device_identity() == ids[0]
    assert sharding.device_census(ids, 4)["ok"]
    assert not sharding.device_census([ids[0], sharding.device_identity(0)], 2)["ok"]
