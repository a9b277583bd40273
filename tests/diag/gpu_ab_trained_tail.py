"""Is a failure of test_trained_checkpoint_tail_over_8192_sites the checkpoint or the build?  Trains the test's checkpoints ONCE, saves them,
and evaluates the SAME weights in subprocesses with different libccsm builds (CCSM_LIB_PATH): per build and arithmetic max |dprob| against the
C oracle over the test's 8192 sites, sites beyond 1e-5 / 5e-5, and what the probe selects.
usage: python tests/diag/gpu_ab_trained_tail.py <lib_a.so> <lib_b.so> [repeats=2]      (child: ... --eval <weights.npz>)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ccsmeth_amd.utils import synth

if len(sys.argv) > 2 and sys.argv[1] == "--eval":
    from ccsmeth_amd.models import DeviceModel
    from oracle import c_oracle
    wt = dict(np.load(sys.argv[2]))
    m = 8192
    sv = synth.synth_sites(m, 143)
    h1, h2 = synth.synth_h0(m, 144)
    args = (sv["kmer1"], sv["ipd1"], sv["pw1"], sv["npass1"], sv["kmer2"], sv["ipd2"], sv["pw2"], sv["npass2"])
    _, ref = c_oracle.forward(wt, *args, h1, h2, threads=c_oracle.usable_threads())
    out = {}
    for prec in (0, 4, 6, 5, 3):
        dm = DeviceModel(wt, device=0, precision=prec)
        ws = dm.workspace(m)
        _, probs = ws.forward_host(*args, h0=(h1, h2))
        ws.close()
        d = np.abs(probs - ref)[:, 1]
        out[str(prec)] = dict(used=dm.precision, max=float(d.max()), n1=int((d > 1e-5).sum()), n5=int((d > 5e-5).sum()),
                              probe=[dm.probe_error, dm.probe_tail, dm.probe_error_mxd, dm.probe_tail_mxd, dm.probe_error_hybrid, dm.probe_tail_hybrid])
        dm.close()
    print("RESULT " + json.dumps(out))
    sys.exit(0)

from ccsmeth_amd.train import Trainer
libs = sys.argv[1:3]
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = 512
pool = synth.synth_sites(n * 8, 42)
lab = lambda q: (q["ipd1"][:, 10] + q["ipd2"][:, 10] > 0).astype(np.int64)  # noqa: E731
for rep in range(repeats):
    for wseed, steps in ((41, 960), (5, 320)):
        tr = Trainer(synth.synth_weights(wseed), device=0, max_sites=n)
        for k in range(steps):
            i = (k % 8) * n
            q = {key: v[i:i + n] for key, v in pool.items()}
            tr.forward_backward(q, lab(q), h0=None, dropout_rate=0.5, seed=wseed, step=k)
            tr.step(1e-3)
        wt = tr.state_dict()
        tr.close()
        path = "/tmp/ab_%d_%d_%d.npz" % (wseed, steps, rep)
        np.savez(path, **wt)
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--eval", path], env=dict(os.environ, CCSM_LIB_PATH=os.path.abspath(lib)),
                               capture_output=True, text=True)
            res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if not res:
                print("checkpoint %d/%d run %d, %s: FAILED rc %d %s" % (wseed, steps, rep, os.path.basename(lib), r.returncode, r.stderr[-400:]))
                continue
            o = json.loads(res[0][7:])
            line = "  ".join("%s->%d max %.2e n1 %d n5 %d" % (p, o[p]["used"], o[p]["max"], o[p]["n1"], o[p]["n5"]) for p in ("0", "4", "6", "5", "3"))
            t = o["0"]
            verdict = "PASS" if (t["max"] < 1e-4 and t["n5"] <= 2 and t["n1"] <= 81) else "FAIL"
            print("checkpoint %d/%d run %d, %-22s test rule on the selected arithmetic: %s | %s" % (wseed, steps, rep, os.path.basename(lib), verdict, line), flush=True)
