import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repository root
import torch
from awesomeslam_amd.core import Core
from oracle.c_oracle import CFilter
from test_gpu_large import synth
for n in (29, 43, 61, 79, 83, 99, 115, 131, 143):
    X,Z,P=synth(n,n)
    o=CFilter('ekf',n+1); o.set_state(n,X,Z,P,0.07,-0.03)
    c=Core('ekf',n+1,batch=1,max_obs=4,max_wait=4); c.set_state(0,n,X,Z,P)
    Xg=c.ekf_step(0,0.2,0.1,1.0,Z,0.07,-0.03); o.slam(0.2,0.1,1.0)
    Xo,_,Po=o.state(); Pg=c.state(0)[2]
    E=np.abs(Pg-Po)/np.abs(Po).max()
    bad=np.argwhere(E>1e-9)
    print('n',n,'nt',(n+15)//16,'errX %.2e errP %.2e'%(np.abs(Xg-Xo).max()/np.abs(Xo).max(),E.max()),'bad entries',len(bad), 'rows',sorted(set(bad[:,0]//16))[:10],'cols',sorted(set(bad[:,1]//16))[:10], 'status',c.status(0))
