import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch
from ccsmeth_amd.models import DeviceModel
from ccsmeth_amd.utils import synth
n=2048; dev=torch.device("cuda:0")
dm=DeviceModel(synth.synth_weights(7),0,precision=int(os.environ.get("PREC","3")))
s=synth.synth_sites(n,8); t={k:torch.from_numpy(v).to(dev) for k,v in s.items()}
args=(t["kmer1"],t["ipd1"],t["pw1"],t["npass1"],t["kmer2"],t["ipd2"],t["pw2"],t["npass2"])
ws=dm.workspace(n); ws.set_timing(True)
for _ in range(3): ws.forward_torch(*args)
torch.cuda.synchronize()
tm=np.mean([(ws.forward_torch(*args), torch.cuda.synchronize(), ws.last_timing())[2] for _ in range(8)],axis=0)
print("GRUV=%%s XVAR=%%s PREC=%%s ms [gru0 gru1 gru2 attn misc] %%s" %% (os.environ.get("CCSM_GRU_VERSION","2"), os.environ.get("CCSM_XVAR","0"), os.environ.get("PREC","3"), np.round(tm,4)))
''' % ROOT
for env in sys.argv[1:]:
    e = dict(os.environ)
    for kv in env.split(","):
        if kv:
            k, v = kv.split("=")
            e[k] = v
    subprocess.run([sys.executable, "-c", code], env=e)
