import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import cpu
from fdtd_amd import Stencil, make_program
cpu.lib()
N=512
t=time.time(); a,b=cpu.init_mt(N,N,N); print("oracle init_mt %.2f"%(time.time()-t))
t=time.time(); ca,cb=a.copy(),b.copy(); print("copies %.2f"%(time.time()-t))
t=time.time(); st=Stencil(N,N,N,8); st.mallocCube("p0"); st.mallocCube("p1"); print("create+malloc %.2f"%(time.time()-t))
t=time.time(); st.transferCubeToGPU("p0",a); st.transferCubeToGPU("p1",b); st.sync(); print("upload x2 %.2f"%(time.time()-t))
t=time.time(); st.sweep("p0","p1",make_program("star",radius=8),2); st.sync(); print("sweep %.2f"%(time.time()-t))
ga,gb=np.empty_like(a),np.empty_like(b)
t=time.time(); st.transferCubeToCPU(ga,"p0"); st.transferCubeToCPU(gb,"p1"); print("download x2 %.2f"%(time.time()-t))
t=time.time(); cpu.run("star",ca,cb,2,M=8,order=cpu.ORDER_CHAIN); print("oracle 2 steps M=8 %.2f"%(time.time()-t))
t=time.time(); ok=np.array_equal(ga.view(np.uint32),ca.view(np.uint32)) and np.array_equal(gb.view(np.uint32),cb.view(np.uint32)); print("compare %.2f"%(time.time()-t), ok)
t=time.time(); st.initMt("p0","p1"); print("device initMt %.2f"%(time.time()-t))
t=time.time(); st.close(); print("close %.2f"%(time.time()-t))
