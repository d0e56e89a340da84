"""HBM bandwidth probe with torch's own kernels: pure write (fill), pure read (sum), copy."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (57, 229, 916):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    x.normal_()
    tf = t(lambda: x.fill_(1.0)); ts = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
    print(f"{mb:4d} MiB  fill {mb*1.048576e6/tf/1e12:5.2f} TB/s ({tf*1e6:6.1f} us)   sum {mb*1.048576e6/ts/1e12:5.2f} TB/s   copy {2*mb*1.048576e6/tc/1e12:5.2f} TB/s (r+w)")
