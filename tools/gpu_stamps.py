import sys, ctypes, numpy as np, time
sys.path.insert(0,'.')
import torch
import awesomeslam_amd.core as ac
ac._CORE = ac._CORE.replace('libaslam_core.so','libaslam_core_stamps.so')
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core
NAMES=['assoc','predict+hcoef','fwd rows','fwd cols','S tiles','cholesky','trsm','X update','back rows','back cols']
def run(L,B,T=264):
    tr1=tg.make_traces(L,T,B=min(B,8),seed=1)
    reps=max(1,B//tr1.B)
    tr=tg.Trace(*(np.concatenate([getattr(tr1,f)]*reps) for f in ('odom','dt','obs_new','n_obs','obs','landmarks','truth')),tr1.warmup)
    core=Core('ekf',tg.dim_cap(L),batch=tr.B,max_obs=tr.max_obs,max_wait=256)
    core.set_trace(tr)
    core.replay(0,64); torch.cuda.synchronize()
    lib=ac.core_lib()
    a=(ctypes.c_ulonglong*12)(); lib.aslam_debug_stamps(core._h,a); base=np.array(list(a),dtype=np.float64)
    t=time.time(); core.replay(64,200); torch.cuda.synchronize(); el=time.time()-t
    lib.aslam_debug_stamps(core._h,a); cyc=(np.array(list(a),dtype=np.float64)-base)/200
    print(f'L={L} B={tr.B}: {el/200*1e6:.1f} us/step wall; cycles/step by phase (workgroup 0), total {cyc.sum():.0f}:')
    for nm,c in zip(NAMES,cyc): print(f'   {nm:14s} {c:9.0f}  {100*c/cyc.sum():5.1f}%')
    if hasattr(lib,'aslam_debug_wave_busy'):
        w=(ctypes.c_ulonglong*24)(); lib.aslam_debug_wave_busy(core._h,w); w=np.array(list(w),dtype=np.float64).reshape(12,2)/264  # (role index, not physical wave: role 9 = the diagonal wave, its second figure = update + factorisation of the next diagonal tile)
        print('   per-role cycles/step [row-block roles 0..8: busy in the factorisation loop / behind it; role 9 = the diagonal wave: WAITING at the loop barriers / update + factorisation of the next diagonal tiles; helper roles 10, 11: the S^-1 product section / what follows it]:')
        print('   '+' '.join(f'w{i}:{w[i,0]:.0f}/{w[i,1]:.0f}' for i in range(12)))
run(64,1)
