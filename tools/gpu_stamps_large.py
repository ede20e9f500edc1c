import sys, ctypes, numpy as np, time
sys.path.insert(0,'.')
import torch
import awesomeslam_amd.core as ac
ac._CORE = ac._CORE.replace('libaslam_core.so','libaslam_core_stamps.so')
import awesomeslam_amd.trace as tg
from awesomeslam_amd.core import Core
NAMES=['small_load','small_frontend','X, Hc, Y','P predict','small_store']
L,B=512,int(sys.argv[1]) if len(sys.argv)>1 else 8
tr=tg.make_traces(L,80,B=B,seed=1)
core=Core('ekf',tg.dim_cap(L),batch=B,max_obs=tr.max_obs,max_wait=2048,dtype=ac.F32)
core.set_trace(tr)
core.replay(0,60); torch.cuda.synchronize()
lib=ac.core_lib()
a=(ctypes.c_ulonglong*12)(); lib.aslam_debug_stamps(core._h,a); base=np.array(list(a),dtype=np.float64)
r=(ctypes.c_ulonglong*2)(); lib.aslam_debug_fe_realtime(core._h,r); rbase=np.array(list(r),dtype=np.float64)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
core.replay(60,20,None,None,torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
lib.aslam_debug_fe_realtime(core._h,r); rr=np.array(list(r),dtype=np.float64)-rbase
print('B=%d: 20 callbacks %.3f ms by events; front-end kernel, workgroup 0: %.1f us per launch by s_memrealtime (%d launches)'%(B,e0.elapsed_time(e1),rr[0]/100.0/max(rr[1],1),rr[1]))
lib.aslam_debug_stamps(core._h,a); cyc=(np.array(list(a),dtype=np.float64)-base)/20
print('large frontend cycles/callback (workgroup 0): total %.0f'%cyc.sum())
for nm,c in zip(NAMES,cyc): print(f'   {nm:16s} {c:9.0f}')
for nm,c in zip(['intake+copy','toPoint+narrow','scan','combine+Z','wait walk','A'],cyc[5:11]): print(f'      fe: {nm:16s} {c:9.0f}')


if hasattr(lib,'aslam_debug_fe_per_filter'):
    pf=(ctypes.c_ulonglong*1024)(); lib.aslam_debug_fe_per_filter(core._h,pf,B)
    us=np.array(list(pf)[:B],dtype=np.float64)/100.0
    print('front end, last launch, us inside the kernel per workgroup (filter): min %.0f  median %.0f  mean %.0f  p90 %.0f  max %.0f'%(us.min(),np.median(us),us.mean(),np.percentile(us,90),us.max()))
    st=np.array([core.state(b,with_P=False)[0].shape[0] for b in range(min(B,8))]); print('   dims of the first filters',st, ' wait-list sizes', [len(core.wait_list(b,4096)[0]) for b in range(min(B,8))])
