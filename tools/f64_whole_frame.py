"""configs[1] (1920x1080 x 256 spp, seed 1) rendered whole by the strict f32 oracle (= the GPU frame, bit for bit: tests/test_gpu_parity.py)
and by the oracle over double (liboracle_f64.so): the whole-frame figures bench.py's `f64_reference` (every third row) is a sample of.
CPU only, ~90 s on 8 cores:   python tools/f64_whole_frame.py > profiles/r6/f64_whole_frame.txt"""
import sys,time
sys.path.insert(0,'/root/repo/tests')
import conftest, numpy as np, oracle_lib
w,h,spp=1920,1080,256
o64=oracle_lib.Oracle("liboracle_f64.so"); ost=oracle_lib.Oracle("liboracle.so")
t=time.time(); a=o64.render(o64.scene_analytical(),w,h,spp,seed=1,threads=8); print("f64",time.time()-t,flush=True)
t=time.time(); b=ost.render(ost.scene_analytical(),w,h,spp,seed=1,threads=8); print("strict",time.time()-t,flush=True)
d=b[...,:3].astype(np.float64)-a[...,:3].astype(np.float64)
fin=np.isfinite(d).all(-1); d=np.where(np.isfinite(d),d,0)
l2=np.sqrt((d*d).sum(-1))
print("whole frame: rmse %.4e  median %.3e  max %.4e  pixels l2>1e-4: %d of %d (%.3f %%) nonfinite %d"%(np.sqrt((d*d).mean()),np.median(np.abs(d)),np.abs(d).max(),(l2>1e-4).sum(),w*h,100*(l2>1e-4).mean(),(~fin).sum()))
# per row-band rmse profile
for r0 in range(0,1080,120):
    dd=d[r0:r0+120]; print(r0, "%.3e"%np.sqrt((dd*dd).mean()), int((l2[r0:r0+120]>1e-4).sum()))
r=(d*d).mean(axis=(1,2))
print('every third row (bench.py f64_reference): rmse %.4e' % np.sqrt(r[1::3].mean()))
