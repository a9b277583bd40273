"""Host side of `trainm` (ccsmeth_amd/trainm.py) without a GPU: CLI parity with the reference's parser, the sharded sampler
against torch's DistributedSampler, the initial parameters against the reference's own construction (checksums in
tests/golden/train_golden.json), the feature-table reader, the schedulers, and the gradient averaging over gloo (world 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from ccsmeth_amd import trainm

META = json.load(open(os.path.join(GOLDEN, "train_golden.json")))
CLI = json.load(open(os.path.join(GOLDEN, "cli_golden.json")))


def _same_flags(ours_parser, gold):
    ours = {a.dest: a for a in ours_parser._actions if a.dest != "help"}
    for dest, g in gold.items():
        assert dest in ours, dest
        assert sorted(ours[dest].option_strings) == sorted(g["options"]), dest
        assert ours[dest].default == g["default"], dest
        assert bool(ours[dest].required) == g["required"], dest


def test_trainm_and_call_freqb_cli_match_reference_parsers():
    from ccsmeth_amd.call_mods_freq_bam import build_freqb_parser
    _same_flags(trainm.build_parser(), CLI["trainm"])
    _same_flags(build_freqb_parser(), CLI["call_freqb"])
    _same_flags(trainm.build_train_parser(), CLI["train"])
    assert not any(a.dest in ("nodes", "node_rank", "dist_url") for a in trainm.build_train_parser()._actions)
    base = ["--train_file", "t", "--valid_file", "v", "--model_dir", "d"]
    trainm.check_scope(trainm.build_parser().parse_args(base))
    for extra in (["--model_type", "attbilstm2s"], ["--optim_type", "SGD"], ["--is_sn", "yes"], ["--hid_rnn", "128"],
                  ["--lr_scheduler", "Cosine"], ["--use_compile", "yes"]):
        with pytest.raises(ValueError):
            trainm.check_scope(trainm.build_parser().parse_args(base + extra))


def test_shard_indices_equal_torch_distributed_sampler():
    import torch
    from torch.utils.data.distributed import DistributedSampler
    for n, world in ((10, 1), (11, 2), (37, 4), (5, 8), (100, 3)):
        ds = list(range(n))
        for epoch in (0, 3):
            for rank in range(world):
                s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True)
                s.set_epoch(epoch)
                assert list(s) == trainm.shard_indices(n, world, rank, epoch).tolist(), (n, world, rank, epoch)
    assert sorted(np.concatenate([trainm.shard_indices(11, 2, r, 0, shuffle=False) for r in range(2)]).tolist()) == sorted(list(range(11)) + [0])


def test_init_state_dict_is_the_reference_initialisation():
    from ccsmeth_amd.train import PARAM_NAMES, PARAM_SHAPES
    sd = trainm.init_state_dict(1234)
    gold = META["init_tseed_1234"]
    assert list(sd.keys()) == PARAM_NAMES and sorted(PARAM_NAMES) == sorted(gold.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == PARAM_SHAPES[k]
        assert np.allclose(v.ravel()[:4], gold[k]["head"], rtol=0, atol=0), k
        assert abs(float(v.astype(np.float64).sum()) - gold[k]["sum"]) <= 1e-9 * max(1.0, abs(gold[k]["sum"])), k
    assert not np.array_equal(trainm.init_state_dict(1)["fc1.weight"], sd["fc1.weight"])


def test_read_feature_file_and_metrics(tmp_path):
    rng = np.random.default_rng(0)
    rows, want = [], []
    for i in range(7):
        fk = "".join(rng.choice(list("ACGTN"), 21))
        rk = "".join(rng.choice(list("ACGT"), 21))
        fi, fp, ri, rp = (np.round(rng.normal(size=21), 6) for _ in range(4))
        c = lambda a: ",".join(str(x) for x in a)  # noqa: E731
        rows.append("\t".join([".", "-1", ".", "m/%d/ccs" % i, str(10 + i), fk, str(5 + i), c(fi), ".", c(fp), ".", ".", ".",
                               rk, str(9 + i), c(ri), ".", c(rp), ".", ".", ".", str(i % 2)]))
        want.append((fk, rk, fi, fp, ri, rp))
    p = tmp_path / "f.tsv"
    p.write_text("\n".join(rows) + "\n")
    d = trainm.read_feature_file(str(p))
    assert d["kmer1"].shape == (7, 21) and d["kmer1"].dtype == np.uint8 and d["labels"].tolist() == [0, 1, 0, 1, 0, 1, 0]
    for i, (fk, rk, fi, fp, ri, rp) in enumerate(want):
        assert d["kmer1"][i].tolist() == [trainm.BASE2CODE[c] for c in fk] and d["kmer2"][i].tolist() == [trainm.BASE2CODE[c] for c in rk]
        assert np.array_equal(d["ipd1"][i], fi.astype(np.float32)) and np.array_equal(d["pw2"][i], rp.astype(np.float32))
        assert d["npass1"][i] == 5 + i and d["npass2"][i] == 9 + i
    (tmp_path / "bad.tsv").write_text("a\tb\n")
    with pytest.raises(ValueError):
        trainm.read_feature_file(str(tmp_path / "bad.tsv"))
    assert trainm.binary_metrics([1, 1, 0, 0], [1, 0, 1, 0]) == (0.5, 0.5, 0.5)
    assert trainm.binary_metrics([0, 0], [0, 0]) == (1.0, 0.0, 0.0)


def test_schedulers_follow_torch():
    import torch
    from torch.optim.lr_scheduler import ReduceLROnPlateau, StepLR
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=0.001)
    ref = StepLR(opt, step_size=2, gamma=0.1)
    ours = trainm.StepLR(0.001, 2, 0.1)
    for _ in range(7):
        assert abs(ours.lr - opt.param_groups[0]["lr"]) < 1e-15
        opt.step(); ref.step(); ours.step()
    opt = torch.optim.SGD(p, lr=0.01)
    ref = ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=1)
    ours = trainm.ReduceLROnPlateau(0.01, 0.5, 1)
    for m in (1.0, 0.9, 0.95, 0.97, 0.96, 0.5, 0.6, 0.7, 0.8, 0.49999, 0.6):
        ref.step(m); ours.step(m)
        assert abs(ours.lr - opt.param_groups[0]["lr"]) < 1e-15, m


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ccsmeth_amd import trainm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
g = torch.from_numpy(np.random.default_rng(100 + rank).normal(size=3043114).astype(np.float32))
mine = g.clone()
trainm.average_gradients(g, world)
want = sum(np.random.default_rng(100 + r).normal(size=3043114).astype(np.float32) for r in range(world)) / world
assert np.abs(g.numpy() - want).max() < 1e-6
# the two ranks' shards of an epoch partition the padded permutation
idx = trainm.shard_indices(1001, world, rank, 5)
both = [torch.zeros(len(idx), dtype=torch.int64) for _ in range(world)]
dist.all_gather(both, torch.from_numpy(idx))
allidx = torch.stack(both, 1).reshape(-1).numpy()
assert len(allidx) == 1002 and set(allidx.tolist()) == set(range(1001))
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gradient_averaging_over_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
