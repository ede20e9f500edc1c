import sys, ctypes, numpy as np, time
sys.path.insert(0,'.')
import torch
import awesomeslam_amd.core as ac
ac._CORE = ac._CORE.replace('libaslam_core.so','libaslam_core_stamps.so')
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core
NAMES=['frontend','chol(P)','sigma poses + mean','D, Zsig -> HBM','P_pred','Zpred + DZ','GEMM Tc','GEMM S+ (tiles)','chol + solve rows','Sherman-Morrison','X, GEMM P -= Tc K^T']
L=64; T=264
tr=tg.make_traces(L,T,B=4,seed=1)
core=Core('ukf',tg.dim_cap(L),batch=4,max_obs=tr.max_obs,max_wait=256); core.set_trace(tr)
core.replay(0,64); torch.cuda.synchronize()
lib=ac.core_lib(); a=(ctypes.c_ulonglong*12)(); lib.aslam_debug_stamps(core._h,a); base=np.array(list(a),dtype=np.float64)
t=time.time(); core.replay(64,200); torch.cuda.synchronize(); el=time.time()-t
lib.aslam_debug_stamps(core._h,a); cyc=(np.array(list(a),dtype=np.float64)-base)/200
print(f'UKF L={L}: {el/200*1e6:.1f} us/step wall; cycles/step by phase (workgroup 0), total {cyc.sum():.0f}:')
for nm,c in zip(NAMES,cyc): print(f'   {nm:24s} {c:9.0f}  {100*c/cyc.sum():5.1f}%')
if hasattr(lib,'aslam_debug_wave_busy'):
    w=(ctypes.c_ulonglong*24)(); lib.aslam_debug_wave_busy(core._h,w); w=np.array(list(w),dtype=np.float64).reshape(12,2)/264
    print('   factor + solve rows, per-role cycles/step [busy in the factorisation loop / behind it (row blocks: W store + row dots; role 9 = the diagonal wave: forward substitution of Zdiff)]:')
    print('   '+' '.join(f'w{i}:{w[i,0]:.0f}/{w[i,1]:.0f}' for i in range(12)))
