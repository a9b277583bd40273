"""Run `trainm` on synthetic feature tables (label = sign of the centre IPDs) and print the validation curve.  usage: trainm_probe.py [trainm flags]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_train import _make_tables
from ccsmeth_amd import trainm
import pathlib
d = pathlib.Path(tempfile.mkdtemp())
_make_tables(d, int(os.environ.get("NTRAIN", "4096")), 1024)
args = trainm.build_parser().parse_args(["--train_file", str(d / "train.tsv"), "--valid_file", str(d / "valid.tsv"), "--model_dir", str(d / "m")] + sys.argv[1:])
res = trainm.train(args)
print({k: v for k, v in res.items() if k != "acc_hist"}); print(np.round(res["acc_hist"], 3).tolist())
