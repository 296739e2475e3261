"""Where the pipelined chain's host time goes: variants of the pipeline (no file writes, one loader, fewer writers) + allocator statistics per pass."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dreg_nerf_amd.eval_pipeline import ExtractionPipeline
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
td, paths = bench.write_generated_blocks(8, 6, 11)
paths = [paths[i % 8] for i in range(32)]
out = {}
import shutil, tempfile
# every block its own directory (outputs go next to model.pth)
root = tempfile.mkdtemp(prefix="dreg_cp_")
ps = []
for i, p in enumerate(paths):
    d = os.path.join(root, f"b{i}"); os.makedirs(d); os.link(p, os.path.join(d, "model.pth")); ps.append(os.path.join(d, "model.pth"))

def stat():
    s = torch.cuda.memory_stats(dev)
    return {k: s.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "reserved_bytes.all.current", "allocation.all.allocated")}

for label, kw, skip in (("default", {}, ""), ("no_copies", {}, "copy"), ("no_writes", {}, "write"), ("no_files", {"write_files": False}, ""), ("default_again", {}, "")):
    os.environ["DREG_PIPE_SKIP"] = skip
    with ExtractionPipeline(dev, **kw) as pipe:
        for rep in range(3):
            for k in pipe.timings:
                pipe.timings[k] = {} if isinstance(pipe.timings[k], dict) else 0 if isinstance(pipe.timings[k], int) else 0.0
            s0 = stat()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for ex in pipe.run(ps):
                pass
            pipe.flush()
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            s1 = stat()
        out[label] = {"s_per_32_blocks": el, "enqueue_s": pipe.timings["enqueue_s"], "detail": pipe.timings["host_detail_s"], "load_wait_s": pipe.timings["load_wait_s"],
                      "slot_wait_s": pipe.timings["slot_wait_s"], "flush_s": pipe.timings["flush_s"],
                      "alloc_delta": {k: s1[k] - s0[k] for k in s0}, "reserved_GB": s1["reserved_bytes.all.current"] / 1e9}
shutil.rmtree(td, ignore_errors=True); shutil.rmtree(root, ignore_errors=True)
print(json.dumps(out, indent=1))
