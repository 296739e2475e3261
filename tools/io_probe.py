"""What the GPU box's host side can do for the config-5 pipeline: cores, file systems, torch.save / torch.load thread scaling, pinned D2H / H2D rates."""
import os, sys, time, threading, tempfile, subprocess, json
import torch
out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
out["df"] = subprocess.run("df -h /tmp /dev/shm / | cat; nproc; free -g | head -2", shell=True, capture_output=True, text=True).stdout
g = torch.zeros(128, 128, 128, 7)
g.view(-1)[::97] = 1.5
for root in ("/tmp", "/dev/shm"):
    d = tempfile.mkdtemp(dir=root)
    for n in (1, 2, 4, 8, 16, 32):
        def save(i):
            torch.save(g, f"{d}/g{i}.pt")
        t0 = time.time()
        th = [threading.Thread(target=save, args=(i,)) for i in range(n)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.time() - t0
        out[f"save_{root}_{n}thr_GBps"] = n * g.numel() * 4 / dt / 1e9
    # load: mmap + to(device)
    if torch.cuda.is_available():
        torch.cuda.init()
        p = torch.randn(12602992)
        torch.save({"model": {"mlp_base.params": p}, "x": torch.zeros(128 ** 3)}, f"{d}/m.pth")
        for n in (1, 2, 4, 8):
            def load(i):
                s = torch.load(f"{d}/m.pth", mmap=True, weights_only=False)
                s["model"]["mlp_base.params"].to("cuda")
            t0 = time.time()
            th = [threading.Thread(target=load, args=(i,)) for i in range(n)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize()
            out[f"load_{root}_{n}thr_ms_per_block"] = 1e3 * (time.time() - t0) / n
    subprocess.run(["rm", "-rf", d])
if torch.cuda.is_available():
    dg = torch.zeros(128, 128, 128, 7, device="cuda")
    hp = torch.empty(128, 128, 128, 7).pin_memory()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(8):
            hp.copy_(dg, non_blocking=True)
        torch.cuda.synchronize(); out["d2h_pinned_GBps"] = 8 * dg.numel() * 4 / (time.time() - t0) / 1e9
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(8):
            dg.copy_(hp, non_blocking=True)
        torch.cuda.synchronize(); out["h2d_pinned_GBps"] = 8 * dg.numel() * 4 / (time.time() - t0) / 1e9
    t0 = time.time(); x = torch.empty(128, 128, 128, 7).pin_memory(); out["pin_alloc_ms"] = 1e3 * (time.time() - t0)
print(json.dumps(out, indent=1))
