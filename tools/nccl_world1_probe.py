"""Probe: what does a world-size-1 `nccl` (= RCCL) process group do on this box -- init with device_id, all_reduce, all_gather_into_tensor,
async work on RCCL's stream, and the same collectives inside a HIP graph capture (thread_local mode)?  Run through gpurun."""
import os, json, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
res = {}
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
t = time.time(); dist.init_process_group("nccl", device_id=dev); res["init_s"] = round(time.time() - t, 2)
x = torch.arange(1024, device=dev, dtype=torch.float32)
t = time.time(); dist.all_reduce(x); torch.cuda.synchronize(); res["first_all_reduce_s"] = round(time.time() - t, 2)
res["all_reduce_ok"] = bool((x == torch.arange(1024, device=dev)).all())
big = torch.randn(16 << 20, device=dev); ref = big.clone()
w = dist.all_reduce(big, async_op=True); w.wait(); torch.cuda.synchronize(); res["async_all_reduce_ok"] = bool((big == ref).all())
src = torch.randint(0, 255, (8, 1024, 1024, 3), device=dev, dtype=torch.uint8); out = torch.empty_like(src)
w = dist.all_gather_into_tensor(out, src, async_op=True); w.wait(); torch.cuda.synchronize(); res["all_gather_ok"] = bool((out == src).all())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dist.all_gather_into_tensor(out, src)
e1.record(); torch.cuda.synchronize(); res["all_gather_25MB_ms"] = round(e0.elapsed_time(e1) / 20, 4)
e0.record()
for _ in range(20):
    dist.all_reduce(big)
e1.record(); torch.cuda.synchronize(); res["all_reduce_64MB_ms"] = round(e0.elapsed_time(e1) / 20, 4)
# capture
for mode in ("thread_local", "global"):
    try:
        g = torch.cuda.CUDAGraph(); y = torch.ones(1 << 20, device=dev); s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                y.mul_(2.0); w = dist.all_reduce(y, async_op=True); w.wait(); y.add_(1.0)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        y.fill_(1.0); torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode=mode):
            y.mul_(2.0); w = dist.all_reduce(y, async_op=True); w.wait(); y.add_(1.0)
        y.fill_(1.0); g.replay(); g.replay(); torch.cuda.synchronize()
        res["capture_" + mode] = float(y[0].item())          # 1 -> 3 -> 7
    except Exception as e:
        res["capture_" + mode] = "ERR " + repr(e)[:300]
        try:
            torch.cuda.synchronize()
        except Exception as e2:
            res["capture_" + mode + "_sync"] = repr(e2)[:200]
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True); open("gpurun_out/nccl_world1_probe.json", "w").write(json.dumps(res))
dist.destroy_process_group()
