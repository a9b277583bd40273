"""Wall time of ONE blocking forward call through the Python mirror of the reference's operator (features in host memory, probabilities
back in host memory): what a caller that cannot coalesce sees.   usage: python tools/time_forward_call.py [precision=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dm = DeviceModel(synth.synth_weights(7), 0, precision=prec)
for n in (64, 512, 2048, 4096, 12288):
    s = synth.synth_sites(n, 5)
    ws = dm.workspace(n)
    args = (s["kmer1"], s["ipd1"], s["pw1"], s["npass1"], s["kmer2"], s["ipd2"], s["pw2"], s["npass2"])
    for _ in range(5):
        ws.forward_host(*args)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        ws.forward_host(*args)
    dt = (time.perf_counter() - t0) / reps
    print("%5d sites per call: %.3f ms per call = %.3g sites/s" % (n, dt * 1e3, n / dt))
    ws.close()
