"""`python -m ccsmeth_amd trainm` — the reference's multi-GPU training command (ccsmeth/train_multigpu.py, flags of
ccsmeth.py `trainm`) on libccsm_train.

One process per GPU.  Launch N ranks with `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
-m ccsmeth_amd trainm ...` (RANK / WORLD_SIZE / LOCAL_RANK from the environment; the reference's own mp.spawn flags --nodes,
--ngpus_per_node, --dist-url, --node_rank are accepted and ignored in favour of that launcher).  Per step every rank runs
forward + backward on its shard of the epoch's permutation (torch DistributedSampler semantics, train_multigpu.py:186-193,
:256), the flat 12.2 MB gradient buffer is summed over RCCL with one all_reduce and divided by the world size (DDP, :171-172),
and clip_grad_norm_(0.5) + Adam then run identically on every rank (:309-312), so the replicas stay bit-identical without a
broadcast.

Note on the reference: train_multigpu.py:142 starts a second `if` chain, so its `else: raise ValueError("--model_type not
right!")` fires for attbigru2s; this command implements the evident intent (attbigru2s trains)."""
import argparse
import os
import re
import sys
import time
from collections import OrderedDict

import numpy as np

BASE2CODE = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4, 'W': 4, 'S': 4, 'M': 4, 'K': 4, 'R': 4, 'Y': 4, 'B': 4, 'V': 4, 'D': 4, 'H': 4,
             'Z': 4}                                                    # utils/process_utils.py:26-29


def build_parser():
    """ccsmeth.py `trainm` flags and defaults."""
    p = argparse.ArgumentParser(prog="ccsmeth_amd trainm", description="train an attbigru2s model, one process per GPU")
    p.add_argument('--train_file', type=str, required=True)
    p.add_argument('--valid_file', type=str, required=True)
    p.add_argument('--model_dir', type=str, required=True)
    p.add_argument('--model_type', type=str, default="attbigru2s")
    p.add_argument('--seq_len', type=int, default=21)
    p.add_argument('--is_npass', type=str, default="yes")
    p.add_argument('--is_sn', type=str, default="no")
    p.add_argument('--is_map', type=str, default="no")
    p.add_argument('--is_stds', type=str, default="no")
    p.add_argument('--class_num', type=int, default=2)
    p.add_argument('--dropout_rate', type=float, default=0.5)
    p.add_argument('--layer_rnn', type=int, default=3)
    p.add_argument('--hid_rnn', type=int, default=256)
    p.add_argument('--layer_trans', type=int, default=6)
    p.add_argument('--nhead', type=int, default=4)
    p.add_argument('--d_model', type=int, default=256)
    p.add_argument('--dim_ff', type=int, default=512)
    p.add_argument('--optim_type', type=str, default="Adam")
    p.add_argument('--batch_size', type=int, default=512)
    p.add_argument('--lr_scheduler', type=str, default='StepLR')
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--lr_decay', type=float, default=0.1)
    p.add_argument('--lr_decay_step', type=int, default=1)
    p.add_argument('--lr_patience', type=int, default=0)
    p.add_argument('--lr_mode_strategy', type=str, default="last")
    p.add_argument('--max_epoch_num', type=int, default=50)
    p.add_argument('--min_epoch_num', type=int, default=10)
    p.add_argument('--pos_weight', type=float, default=1.0)
    p.add_argument('--step_interval', type=int, default=500)
    p.add_argument('--dl_num_workers', type=int, default=0)
    p.add_argument('--init_model', type=str, default=None)
    p.add_argument('--tseed', type=int, default=1234)
    p.add_argument('--use_compile', type=str, default="no")
    p.add_argument('--nodes', type=int, default=1)
    p.add_argument('--ngpus_per_node', type=int, default=2)
    p.add_argument('--dist-url', dest="dist_url", type=str, default="tcp://127.0.0.1:12315")
    p.add_argument('--node_rank', type=int, default=0)
    p.add_argument('--epoch_sync', action="store_true", default=False)
    return p


def build_train_parser():
    """ccsmeth.py `train` (single process; the reference wraps the model in nn.DataParallel, train.py:128-130): the same flags
    without the launcher options, plus --dl_offsets (how the reference's data loader seeks lines; the table is loaded whole here).
    Runs the same trainer on one GPU."""
    p = build_parser()
    p.prog = "ccsmeth_amd train"
    for dest in ("nodes", "ngpus_per_node", "dist_url", "node_rank", "epoch_sync"):
        act = next(a for a in p._actions if a.dest == dest)
        p._remove_action(act)
        for o in act.option_strings:
            p._option_string_actions.pop(o, None)
    p.add_argument('--dl_offsets', action="store_true", default=False)
    return p


def _yes(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def check_scope(args):
    if args.model_type != "attbigru2s":
        raise ValueError("--model_type not right!")                                       # train_multigpu.py:152
    if not _yes(args.is_npass) or _yes(args.is_stds) or _yes(args.is_sn) or _yes(args.is_map):
        raise ValueError("this build implements --is_npass yes --is_stds no --is_sn no --is_map no")
    if (args.layer_rnn, args.hid_rnn, args.class_num, args.seq_len) != (3, 256, 2, 21):
        raise ValueError("this build implements --seq_len 21 --layer_rnn 3 --hid_rnn 256 --class_num 2")
    if args.optim_type != "Adam":
        raise ValueError("this build implements --optim_type Adam")
    if args.lr_scheduler not in ("StepLR", "ReduceLROnPlateau"):
        raise ValueError("--lr_scheduler is not right!")                                  # :244
    if _yes(args.use_compile):
        raise ValueError("--use_compile applies to the reference's torch model only")


# ------------------------------------------------------------------------------------------------- data
def read_feature_file(path):
    """The 22-column feature table `ccsmeth extract` writes (dataloader.py:15-47 parse_a_line), loaded whole:
    dict of kmer1/2 (N,21) uint8, ipd1/2, pw1/2 (N,21) float32, npass1/2 (N,) float32 and labels (N,) int64."""
    k1, k2, i1, i2, p1, p2, n1, n2, lab = [], [], [], [], [], [], [], [], []
    lut = np.full(256, 4, np.uint8)
    for ch, code in BASE2CODE.items():
        lut[ord(ch)] = code
    with open(path, "r") as rf:
        for line in rf:
            w = line.rstrip("\n").split("\t")
            if len(w) < 22:
                if line.strip():
                    raise ValueError("%s: a feature line has %d columns, expected 22" % (path, len(w)))
                continue
            k1.append(lut[np.frombuffer(w[5].encode("ascii"), np.uint8)])
            n1.append(int(w[6]))
            i1.append(np.array(w[7].split(","), dtype=np.float64))
            p1.append(np.array(w[9].split(","), dtype=np.float64))
            k2.append(lut[np.frombuffer(w[13].encode("ascii"), np.uint8)])
            n2.append(int(w[14]))
            i2.append(np.array(w[15].split(","), dtype=np.float64))
            p2.append(np.array(w[17].split(","), dtype=np.float64))
            lab.append(int(w[21]))
    f = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)  # noqa: E731
    if not lab:
        raise ValueError("%s holds no samples" % path)
    return dict(kmer1=np.stack(k1), kmer2=np.stack(k2), ipd1=f(i1), ipd2=f(i2), pw1=f(p1), pw2=f(p2),
                npass1=np.asarray(n1, np.float32), npass2=np.asarray(n2, np.float32), labels=np.asarray(lab, np.int64))


def take(data, idx):
    return {k: v[idx] for k, v in data.items()}


def shard_indices(n, world, rank, epoch, shuffle=True, seed=0):
    """torch.utils.data.distributed.DistributedSampler(shuffle=True, seed=0, drop_last=False): the epoch's permutation from
    torch.Generator().manual_seed(seed + epoch), padded by wrapping around to a multiple of `world`, then every world-th index."""
    import torch
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return np.asarray(idx[rank:total:world], dtype=np.int64)


def init_state_dict(tseed):
    """The parameters ModelAttRNN.__init__ draws after torch.manual_seed(tseed) (train_multigpu.py:470, models.py:32-75):
    modules constructed in the reference's order so that the generator is consumed the same way."""
    import torch
    from torch import nn
    torch.manual_seed(tseed)
    embed = nn.Embedding(5, 8)
    rnn = nn.GRU(11, 256, 3, dropout=0, batch_first=True, bidirectional=True)
    wa, ua, va = nn.Linear(512, 256, bias=False), nn.Linear(512, 256, bias=False), nn.Linear(256, 1, bias=False)   # attention.py:43-45
    fc1 = nn.Linear(1024, 2)
    nn.init.uniform_(embed.weight, -0.1, 0.1)
    nn.init.zeros_(fc1.bias)
    nn.init.uniform_(fc1.weight, -0.1, 0.1)
    sd = OrderedDict()
    sd["embed.weight"] = embed.weight.detach().numpy().copy()
    for k, v in rnn.state_dict().items():
        sd["rnn." + k] = v.detach().numpy().copy()
    sd["_att3.Wa.weight"], sd["_att3.Ua.weight"], sd["_att3.va.weight"] = (m.weight.detach().numpy().copy() for m in (wa, ua, va))
    sd["fc1.weight"], sd["fc1.bias"] = fc1.weight.detach().numpy().copy(), fc1.bias.detach().numpy().copy()
    return sd


class StepLR:
    def __init__(self, lr, step_size, gamma):
        self.base, self.step_size, self.gamma, self.epoch = lr, max(1, step_size), gamma, 0

    @property
    def lr(self):
        return self.base * self.gamma ** (self.epoch // self.step_size)

    def step(self, metric=None):
        self.epoch += 1


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold=1e-4 relative, cooldown 0, min_lr 0, eps 1e-8)."""

    def __init__(self, lr, factor, patience):
        self.lr, self.factor, self.patience, self.best, self.bad = lr, factor, patience, float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1 - 1e-4):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = self.lr * self.factor
            if self.lr - new > 1e-8:
                self.lr = new
            self.bad = 0


def average_gradients(flat, world):
    """DDP's gradient averaging as one collective: sum over ranks (RCCL all_reduce on the flat buffer), divide by world."""
    import torch.distributed as dist
    if world > 1:
        import torch
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        if flat.is_cuda:
            # libccsm_train runs on its own stream and does not know torch's: the averaged gradients must be complete before
            # trainer.step() reads them
            torch.cuda.current_stream(flat.device).synchronize()
    return flat


def binary_metrics(labels, pred):
    """sklearn accuracy / precision / recall of class 1 (0 when undefined)."""
    labels, pred = np.asarray(labels), np.asarray(pred)
    tp = int(((pred == 1) & (labels == 1)).sum())
    acc = float((pred == labels).mean()) if len(labels) else 0.0
    prec = tp / float((pred == 1).sum()) if (pred == 1).any() else 0.0
    rec = tp / float((labels == 1).sum()) if (labels == 1).any() else 0.0
    return acc, prec, rec


def train(args, log=sys.stderr):
    import torch
    import torch.distributed as dist
    from .train import Trainer
    check_scope(args)
    for p in (args.train_file, args.valid_file):
        if not os.path.exists(p):
            raise ValueError("%s does not exist" % p)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "12315")
        # nccl = RCCL over xGMI; CCSM_DIST_BACKEND=gloo lets several ranks share one GPU (tests on a 1-GPU box)
        dist.init_process_group(backend=os.environ.get("CCSM_DIST_BACKEND", "nccl"), init_method="env://", rank=rank, world_size=world)
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    say = (lambda m: (log.write(m + "\n"), log.flush())) if rank == 0 else (lambda m: None)

    model_dir = os.path.abspath(args.model_dir).rstrip("/") if args.model_dir != "/" else "/"
    if rank == 0:
        if not os.path.exists(model_dir):
            os.makedirs(model_dir)
        else:                                                            # train_multigpu.py:108-112
            rx = re.compile(r"" + args.model_type + r"\..*b\d+_epoch\d+\.ckpt*")
            for f in os.listdir(model_dir):
                if rx.match(f) is not None:
                    os.remove(os.path.join(model_dir, f))
    if args.init_model is not None:
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in torch.load(args.init_model, map_location="cpu").items())
    else:
        sd = init_state_dict(args.tseed)
    if world > 1:
        dist.barrier()
    grads = torch.zeros(3043114, dtype=torch.float32, device="cuda:%d" % local_rank)
    trainer = Trainer(sd, device=local_rank, max_sites=args.batch_size, grads_tensor=grads)
    train_data, valid_data = read_feature_file(args.train_file), read_feature_file(args.valid_file)
    n_train, n_valid = len(train_data["labels"]), len(valid_data["labels"])
    sched = StepLR(args.lr, args.lr_decay_step, args.lr_decay) if args.lr_scheduler == "StepLR" else \
        ReduceLROnPlateau(args.lr, args.lr_decay, args.lr_patience)
    total_step = -(-len(shard_indices(n_train, world, rank, 0)) // args.batch_size)
    say("training_process-%d total_step: %d (%d train / %d valid samples, world %d)" % (os.getpid(), total_step, n_train, n_valid, world))
    best_acc, best_loc, lowest_loss, acc_hist, gstep = 0.0, 0, 10000.0, [], 0
    feats = ("kmer1", "kmer2", "ipd1", "ipd2", "pw1", "pw2", "npass1", "npass2")
    for epoch in range(args.max_epoch_num):
        idx = shard_indices(n_train, world, rank, epoch)
        no_best, tlosses, start = True, [], time.time()
        for i in range(total_step):
            b = idx[i * args.batch_size:(i + 1) * args.batch_size]
            sites = {k: train_data[k][b] for k in feats}
            # device-drawn initial states: the generator's counter is the running site index of this rank, so every forward draws fresh
            # values as the reference's torch.randn does (a per-step offset of 1 made consecutive steps share all but one site's window)
            loss, _ = trainer.forward_backward(sites, train_data["labels"][b], h0=None, pos_weight=args.pos_weight,
                                               dropout_rate=args.dropout_rate, seed=args.tseed * 1009 + rank, step=gstep,
                                               h0_offset=gstep * args.batch_size)
            average_gradients(grads, world)
            trainer.step(sched.lr, max_norm=0.5)
            gstep += 1
            tlosses.append(loss)
            if (i + 1) % args.step_interval == 0 or (i + 1) == total_step:
                say("Epoch [%d/%d], Step [%d/%d]; TrainLoss: %.4f; Time: %.2fs" % (epoch + 1, args.max_epoch_num, i + 1, total_step,
                                                                                   np.mean(tlosses), time.time() - start))
                start, tlosses = time.time(), []
        # validation (train_multigpu.py:326-420): every rank its shard of the (epoch-0) permutation, loss averaged over ranks
        vidx = shard_indices(n_valid, world, rank, 0)
        vlosses, vlabels, vpred = [], [], []
        for i in range(-(-len(vidx) // args.batch_size)):
            b = vidx[i * args.batch_size:(i + 1) * args.batch_size]
            vloss, logits = trainer.evaluate({k: valid_data[k][b] for k in feats}, valid_data["labels"][b], h0=None,
                                             pos_weight=args.pos_weight, seed=args.tseed * 1009 + rank + 500009,      # validation: its own stream
                                             h0_offset=(gstep + i) * args.batch_size)
            if world > 1:
                t = torch.tensor([vloss], device="cuda:%d" % local_rank)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                vloss = float(t.item()) / world
            vlosses.append(vloss)
            vlabels += valid_data["labels"][b].tolist()
            vpred += logits.argmax(1).tolist()
        v_acc, v_prec, v_rec = binary_metrics(vlabels, vpred)
        v_meanloss = float(np.mean(vlosses))
        if v_acc > best_acc - 0.0002:
            if rank == 0:
                _save(trainer, os.path.join(model_dir, "%s.b%d_epoch%d.ckpt" % (args.model_type, args.seq_len, epoch + 1)))
            if v_acc > best_acc:
                best_acc, best_loc = v_acc, epoch + 1
            if acc_hist and v_acc > acc_hist[-1] and rank == 0:
                _save(trainer, os.path.join(model_dir, "%s.betterthanlast.b%d_epoch%d.ckpt" % (args.model_type, args.seq_len, epoch + 1)))
        if v_meanloss < lowest_loss:
            lowest_loss, no_best = v_meanloss, False
        acc_hist.append(v_acc)
        say("Epoch [%d/%d]; LR: %.4e; ValidLoss: %.4f, Acc: %.4f, Prec: %.4f, Reca: %.4f, Best_acc: %.4f; Time: %.2fs"
            % (epoch + 1, args.max_epoch_num, sched.lr, v_meanloss, v_acc, v_prec, v_rec, best_acc, time.time() - start))
        if no_best and epoch >= args.min_epoch_num - 1:
            say("training_process-%d early stop!" % os.getpid())
            break
        sched.step(v_meanloss)
    say("best model is in epoch %d (Acc: %s)" % (best_loc, best_acc))
    checksum = float(sum(np.abs(v.astype(np.float64)).sum() for v in trainer.state_dict().values()))
    trainer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    res = dict(best_acc=best_acc, best_epoch=best_loc, epochs=len(acc_hist), valid_loss=lowest_loss, acc_hist=acc_hist, rank=rank, world=world,
               param_checksum=checksum, steps=gstep)
    if os.environ.get("CCSM_TRAINM_REPORT"):
        import json
        with open(os.environ["CCSM_TRAINM_REPORT"] + ".rank%d.json" % rank, "w") as wf:
            json.dump(res, wf)
    return res


def _save(trainer, path):
    import torch
    torch.save(OrderedDict((k, torch.from_numpy(v)) for k, v in trainer.state_dict().items()), path)


def main(argv=None):
    train(build_parser().parse_args(argv))


def main_train(argv=None):
    train(build_train_parser().parse_args(argv))
