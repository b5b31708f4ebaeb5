import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import codesign_kernels_amd as M
from oracle import oracle as O
from util import run_hip
M.set_variant(M.VARIANT_FAST)
for shape,dist in (((4096,32,28),1),((4097,32,28),1),((256,32,28),2),((256,32,28),3),((48,32,58),2)):
    inp=O.make_inputs(*shape,seed=100,dist=dist,dtype=np.float32)
    f,fl=run_hip(M,inp); fr,flr=O.advect(inp,nthreads=4)
    print(shape,dist,'max|df| %.3e'%np.abs(f.astype(np.float64)-fr).max(),'relL1 %.3e'%O.rel_l1(f,fr),'flux max|d| %.3e'%np.abs(fl.astype(np.float64)-flr).max(), 'flux relL1 %.3e'%O.rel_l1(fl[:,:-1],flr[:,:-1]))
