"""Probe: which host-side torch calls block on previously queued GPU work on ROCm?"""
import time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
def busy(n=20):
    for _ in range(n):
        torch.mm(a, a)
torch.cuda.synchronize()
t0 = time.perf_counter(); busy(); torch.cuda.synchronize(); print(f"busy work = {1e3*(time.perf_counter()-t0):.1f} ms")
for name, fn in [
    ("torch.tensor(list, device)", lambda: torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=dev)),
    ("cpu_tensor.to(dev)", lambda: torch.arange(1000).to(dev)),
    ("pinned.to(dev, non_blocking)", lambda: torch.arange(1000).pin_memory().to(dev, non_blocking=True)),
    ("torch.zeros(device)", lambda: torch.zeros(1000, device=dev)),
    ("torch.full", lambda: torch.full((1000,), 3, dtype=torch.int32, device=dev)),
]:
    torch.cuda.synchronize(); busy()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:34s}: host blocked {1e3*dt:.2f} ms")
# side stream tolist while main is busy
side = torch.cuda.Stream()
x = torch.arange(10, device=dev)
torch.cuda.synchronize(); busy()
t0 = time.perf_counter()
with torch.cuda.stream(side):
    y = (x + 1).tolist()
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"side-stream tolist while main busy : host blocked {1e3*dt:.2f} ms")
torch.cuda.synchronize(); busy()
t0 = time.perf_counter(); y = (x + 1).tolist(); dt = time.perf_counter() - t0
print(f"main-stream tolist                 : host blocked {1e3*dt:.2f} ms")
