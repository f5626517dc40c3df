"""HBM streaming rates on this GPU (torch ops, HIP events): pure write (fill), copy (read+write), pure read (sum)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (33, 268, 805, 2147):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    tm = t(lambda: torch.mul(a, 2.0, out=b))
    print(f"{mb:5d} MB: fill {mb/1e3/tf/1e3:5.2f} TB/s ({tf*1e6:6.1f} us) | copy {2*mb/1e3/tc/1e3:5.2f} TB/s ({tc*1e6:6.1f} us) | mul {2*mb/1e3/tm/1e3:5.2f} TB/s | sum(read) {mb/1e3/tr/1e3:5.2f} TB/s ({tr*1e6:6.1f} us)")
