"""Front-end phase cycles of the single-CU EKF (needs libaslam_core_stamps.so built with -DASLAM_STAMPS=1 -DASLAM_FE_STAMPS=1)."""
import sys, ctypes, numpy as np
sys.path.insert(0,'.')
import torch
import awesomeslam_amd.core as ac
ac._CORE = ac._CORE.replace('libaslam_core.so','libaslam_core_stamps.so')
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core
L,B=64,1
tr=tg.make_traces(L,264,B=B,seed=1)
core=Core('ekf',tg.dim_cap(L),batch=B,max_obs=tr.max_obs,max_wait=256); core.set_trace(tr)
core.replay(0,64); torch.cuda.synchronize()
lib=ac.core_lib(); a=(ctypes.c_ulonglong*6)(); lib.aslam_debug_fe_stamps(core._h,a); base=np.array(list(a),float)
core.replay(64,200); torch.cuda.synchronize(); lib.aslam_debug_fe_stamps(core._h,a); cyc=(np.array(list(a),float)-base)/200
for nm,c in zip(['intake+copy','toPoint+narrow+A','scan+combine','Z','wait walk','(barrier)'],cyc): print(f'   fe: {nm:16s} {c:9.0f}')
