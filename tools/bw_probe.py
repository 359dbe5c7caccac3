import sys, time
sys.path.insert(0, '/root/repo')
import torch
n = 1528270037
x = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
for name, f in (("sum int32", lambda: x.sum()), ("max int32", lambda: x.max()), ("copy", lambda: x.clone())):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    b = n * 4 * (2 if name == "copy" else 1)
    print(name, "%.3f ms" % (dt * 1e3), "%.2f TB/s" % (b / dt / 1e12))
