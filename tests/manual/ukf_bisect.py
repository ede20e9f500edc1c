import sys, glob, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repository root
import torch
import awesomeslam_amd.core as ac
k=sys.argv[1]
if k!='base': ac._CORE = ac._CORE.replace('libaslam_core.so','libaslam_core_v%s.so'%k)
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core
from oracle.c_oracle import CFilter
from util import rel_err
L,T=5,150
tr=tg.make_traces(L,T,B=2,seed=21)
core=Core('ukf',tg.dim_cap(L),batch=2,max_obs=tr.max_obs,max_wait=512)
core.set_trace(tr)
p=torch.zeros((2,T,3),dtype=torch.float64,device='cuda')
core.replay(0,T,p.data_ptr(),None); torch.cuda.synchronize()
o=CFilter('ukf',tg.dim_cap(L)); po,do=o.replay(tr[0]); Xo,Zo,Po=o.state(); X,Z,P=core.state(0)
print('variant',k,'rel err pose/X/P %.2e %.2e %.2e'%(rel_err(p.cpu().numpy()[0],po),rel_err(X,Xo),rel_err(P,Po)))
