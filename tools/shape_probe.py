import torch, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
def timeit(f, xs, steps=100):
    n=len(xs); outs=[None]*n
    for i in range(10): outs[i%n]=f(xs[i%n])
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): outs[i%n]=f(xs[i%n])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/steps
for shape in ((4096,11008),(4096,11264),(4096,8192),(8192,8192)):
    x=torch.randn(*shape,device="cuda"); xs=[x,x.clone(),x.clone()]
    thr=[1.0+0.001*i for i in range(shape[0])]
    qa=Q.WeightsSymmetricInferableQuantizer(8,thr,True,0)
    ql=Q.WeightsLUTSymmetricInferableQuantizer(4,[-128.,-96.,-64.,-40.,-24.,-12.,-5.,0.,5.,12.,24.,40.,64.,96.,120.,127.],[4.5]*shape[0],True,0,2)
    b=x.numel()*8
    ta=timeit(qa,xs); 
    res=[f"affine {ta:7.2f}us {b/ta/1e3:6.0f}GB/s"]
    for pers in (0,):
        native.set_tuning("heavy_persistent",pers)
        for hu in (2,4,8):
            native.set_tuning("heavy_unroll",hu); tl=timeit(ql,xs); res.append(f"lut p{pers} U{hu} {tl:7.2f}us {b/tl/1e3:6.0f}GB/s")
    native.set_tuning("heavy_persistent",1)
    native.set_tuning("heavy_unroll",0)
    print(shape," | ".join(res),flush=True)
