import sys, numpy as np, torch
sys.path[:0]=['.']
from uit_mobile_amd.synth import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev=torch.device("cuda",0)
for name,B in (("uit_xs",8192),("uit_xxxs",4096)):
    sd={k: torch.from_numpy(np.asarray(v)) for k,v in make_state_dict(name,13).items()}
    xs=[synth_batch(B,16000,i*B,dev,False) for i in range(3)]
    m=getattr(models,name)(outputdim=537,target_length=102,precision="f16").to(dev); m.load_state_dict(sd); m.eval()
    m.set_option("fe_asym_pm",590)
    for rep in range(2):
        for skip in (1,0):
            m.set_option("fe_tail_skip",skip)
            N=80; evs=[[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(N)]
            for i in range(40): m.frontend(xs[i%3])
            for i in range(N):
                evs[i][0].record(); m.frontend(xs[i%3]); evs[i][1].record()
            torch.cuda.synchronize()
            print(name,B,"fe_tail_skip",skip,"front-end median %.4f ms"%np.median([e[0].elapsed_time(e[1]) for e in evs]))
