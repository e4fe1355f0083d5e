import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np, tempfile
from opticalraytrace_amd.sweeps import Sweep
from opticalraytrace_amd.system import OpticalSystem
from opticalraytrace_amd.params import resource_dir
from oracle.binding import Oracle
n=20000
sw=Sweep(nphotons=n, data_dir=tempfile.mkdtemp())
sw.lens_experiment(); sw.iris_experiment()
bad=0
for idx,(name,s,res) in enumerate(sw.results):
    osys=OpticalSystem.from_settings(s, resource_dir()); orc=Oracle(osys)
    img=np.zeros((2,401,401),np.int32); cnt=np.zeros(8,np.uint64)
    orc.trace(1,0,n,123456789,img,cnt); orc.trace(2,0,n,123456789,img,cnt)
    d=np.abs(res.counters.astype(np.int64)-cnt.astype(np.int64)).max()
    if d>2:
        bad+=1
        if bad<8: print(idx,name,s.bottle_file,s.L2_file,s.L3_file,s.iris,s.iris_size,res.counters, cnt)
print("bad",bad,"of",len(sw.results))
sw.close()
