"""Threads vs processes for the grid-file writers: aggregate torch.save rate and what a launching main thread loses meanwhile (measurement for eval_pipeline)."""
import json, os, sys, time, threading, tempfile, shutil
import multiprocessing as mp
import torch

def proc_worker(q, done, shm_list):
    torch.set_num_threads(1)
    while True:
        item = q.get()
        if item is None:
            return
        i, path = item
        torch.save(shm_list[i], path)
        done.put(i)

def main():
    out = {}
    d = tempfile.mkdtemp(prefix="wp_")
    K = 8
    bufs = [torch.zeros(128, 128, 128, 7).share_memory_() for _ in range(K)]
    for b in bufs:
        b.view(-1)[::97] = 1.5
    x = torch.zeros(16, device="cuda")
    def main_loop(stop):
        n = 0
        while not stop.is_set():
            x.add_(1); n += 1
            if n % 64 == 0:
                torch.cuda.synchronize()
        return n
    def run(label, start_jobs, wait_jobs):
        stop = threading.Event(); cnt = [0]
        t = threading.Thread(target=lambda: cnt.__setitem__(0, main_loop(stop)))
        t0 = time.perf_counter(); t.start()
        start_jobs(); wait_jobs()
        el = time.perf_counter() - t0
        stop.set(); t.join()
        return el, cnt[0] / el
    # baseline launch rate
    stop = threading.Event(); tt = threading.Timer(0.5, stop.set); tt.start(); t0 = time.perf_counter(); n = main_loop(stop); out["launches_per_s_alone"] = n / (time.perf_counter() - t0)
    files = 64
    for nw in (4, 8, 12):
        # threads
        import concurrent.futures as cf
        ex = cf.ThreadPoolExecutor(nw); futs = []
        el, rate = run("thr", lambda: futs.extend(ex.submit(torch.save, bufs[i % K], f"{d}/t{i}.pt") for i in range(files)), lambda: [f.result() for f in futs])
        ex.shutdown()
        out[f"threads{nw}"] = {"GBps": files * bufs[0].numel() * 4 / el / 1e9, "launches_per_s": rate}
        # processes
        ctx = mp.get_context("fork")
        q, done = ctx.Queue(), ctx.Queue()
        ps = [ctx.Process(target=proc_worker, args=(q, done, bufs), daemon=True) for _ in range(nw)]
        [p.start() for p in ps]
        time.sleep(0.3)
        el, rate = run("proc", lambda: [q.put((i % K, f"{d}/p{i}.pt")) for i in range(files)], lambda: [done.get() for _ in range(files)])
        [q.put(None) for _ in ps]; [p.join() for p in ps]
        out[f"procs{nw}"] = {"GBps": files * bufs[0].numel() * 4 / el / 1e9, "launches_per_s": rate}
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
