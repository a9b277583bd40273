import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
w = synth.synth_weights(7)
for prec in (0, 4, 6, 5, 3, 0):
    t0 = time.perf_counter(); dm = DeviceModel(w, device=0, precision=prec); t1 = time.perf_counter()
    print("precision %d: ccsm_create %.3f s (selected %d)" % (prec, t1 - t0, dm.precision)); dm.close()
