import os, sys, torch
sys.path.insert(0, os.getcwd())
import mct_quantizers_amd as mq
Q = mq.pytorch_quantizers
def timeit(f, xs, steps):
    n=len(xs); outs=[None]*n
    for i in range(50): outs[i%n]=f(xs[i%n])
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): outs[i%n]=f(xs[i%n])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/steps
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for shape in ((512,512,3,3),(512,4608),(2048,4608)):
        q=Q.WeightsSymmetricInferableQuantizer(8,[1.0+(i%97)*0.01 for i in range(shape[0])],True,0)
        x=torch.randn(*shape,device="cuda").to(dt); xs=[x.clone() for _ in range(64)]
        for rep in range(2):
            print(dt, shape, "%.2f us"%timeit(q,xs,2000), flush=True)
