import time, torch
x = torch.randn(4096, 4096, device="cuda")
torch.cuda.synchronize()
def t(f, n=200):
    s = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - s) / n * 1e6
print("torch.cuda.synchronize() on an idle device: %.1f us" % t(torch.cuda.synchronize))
def ev():
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
print("event record + query spin on an idle stream: %.1f us" % t(ev))
def first_kernel():
    torch.cuda.synchronize()
    s = time.perf_counter()
    y = x.mul(2.0)
    e = torch.cuda.Event(); e.record()
    while not e.query(): pass
    return time.perf_counter() - s
ts = sorted(first_kernel() for _ in range(50))
print("launch of one 128 MiB-traffic kernel on an idle GPU until its completion is visible: median %.1f us (kernel itself ~21)" % (ts[25] * 1e6))
