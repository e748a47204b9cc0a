"""Plain HBM streaming rates on this box (torch kernels): fill (write only), copy (1:1), sum (read only)."""
import torch
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, device=dev, dtype=torch.bfloat16).normal_()
    y = torch.empty_like(x)
    tf = timeit(lambda: y.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    ts = timeit(lambda: x.sum())
    ta = timeit(lambda: torch.add(x, x, out=y))
    print(f"{mb:5d} MiB: fill {mb*1.048576/tf*1e3/1e3:5.2f} TB/s | copy {2*mb*1.048576/tc:5.2f} TB/s total | sum(read) {mb*1.048576/ts:5.2f} TB/s | add(x,x)->y {2*mb*1.048576/ta:5.2f} TB/s")
