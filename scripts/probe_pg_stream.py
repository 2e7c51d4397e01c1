import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
side = torch.cuda.Stream()
x = torch.zeros(64, dtype=torch.float64, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
def timed(label, fn):
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)
    t=time.perf_counter(); out = fn(); dt=time.perf_counter()-t
    torch.cuda.synchronize(); print(f"{label:50s} {dt*1e3:8.2f} ms", flush=True); return out
with torch.cuda.stream(side):
    timed("H2D torch.tensor(list, device) on side", lambda: torch.tensor([1.0]*64, dtype=torch.float64, device="cuda"))
    pinned = torch.ones(64, dtype=torch.float64).pin_memory()
    timed("H2D from pinned non_blocking on side", lambda: pinned.to("cuda", non_blocking=True))
    y = torch.ones(64, dtype=torch.float64, device="cuda"); side.synchronize()
    timed("all_reduce on side", lambda: dist.all_reduce(y))
    timed("all_reduce async_op on side + wait", lambda: dist.all_reduce(y, async_op=True).wait())
    timed("tolist on side", lambda: y.tolist())
    land = torch.empty(64, dtype=torch.float64).pin_memory()
    def pinned_read():
        land.copy_(y, non_blocking=True); e = torch.cuda.Event(); e.record(side); e.synchronize(); return land.tolist()
    timed("D2H into pinned + event on side", pinned_read)
for label, stream in (("second pool stream", torch.cuda.Stream()), ("third pool stream", torch.cuda.Stream()), ("high-priority stream", torch.cuda.Stream(priority=-1))):
    with torch.cuda.stream(stream):
        z = torch.ones(64, dtype=torch.float64, device="cuda"); stream.synchronize()
        def read(stream=stream):
            land.copy_(z, non_blocking=True); e = torch.cuda.Event(); e.record(stream); e.synchronize(); return land.tolist()
        timed(f"D2H into pinned + event on {label}", read)
        def reduce_read(stream=stream):
            w = pinned.to("cuda", non_blocking=True); dist.all_reduce(w); land.copy_(w, non_blocking=True)
            e = torch.cuda.Event(); e.record(stream); e.synchronize(); return land.tolist()
        timed(f"pinned H2D + all_reduce + pinned D2H on {label}", reduce_read)
# ... and with the long kernel on a NON-default stream (what a trainer running on its own stream would park)
work = torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)
torch.cuda.synchronize()
with torch.cuda.stream(work):
    torch.cuda._sleep(200_000_000)
with torch.cuda.stream(hi):
    t = time.perf_counter(); w = pinned.to("cuda", non_blocking=True); dist.all_reduce(w); land.copy_(w, non_blocking=True)
    e = torch.cuda.Event(); e.record(hi); e.synchronize(); print(f"{'sleep on a pool stream, reduce on high-priority':50s} {(time.perf_counter() - t) * 1e3:8.2f} ms", flush=True)
torch.cuda.synchronize()
os._exit(0)
