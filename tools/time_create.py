"""Wall time of ccsm_create per arithmetic (0 = the probe decides): weight-stream packing on 16 threads, only for the arithmetics the
probe reaches.  The first call also pays the device initialisation.   usage: python tools/time_create.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
w = synth.synth_weights(7)
for prec in (0, 4, 6, 5, 3, 0):
    t0 = time.perf_counter(); dm = DeviceModel(w, device=0, precision=prec); t1 = time.perf_counter()
    print("precision %d: ccsm_create %.3f s (selected %d)" % (prec, t1 - t0, dm.precision)); dm.close()
