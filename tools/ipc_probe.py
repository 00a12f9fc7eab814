"""development helper: per-step cost of the xGMI write+notify exchange, 2 processes sharing cuda:0
(launch with RANK/WORLD_SIZE/MASTER_* set, e.g. through tools/ipc_probe.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from __graft_entry__ import load_package
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
dims = tuple(int(x) for x in os.environ.get("DIMS", "32,32,32").split(","))
nd = int(os.environ.get("ND", "8"))
gp = pkg.gen_params(*dims, ndomains=nd)
part, st = mg.build_rank_partition(gp, nd, world, rank, via_files=False)
mg.exchange_requests(part, rank, world, dist)
solver = mg.RankSolver(part, rank, world, 0, dist, transport=os.environ.get("TRANSPORT", "ipc"))
def timeit(tag, fn, n):
    solver.synchronize(); dist.barrier(); t = time.perf_counter(); fn(n); t1 = time.perf_counter(); solver.synchronize(); dist.barrier(); t2 = time.perf_counter()
    if rank == 0: print("%-40s host %.1f us/step, total %.1f us/step" % (tag, (t1 - t) / n * 1e6, (t2 - t) / n * 1e6), flush=True)
if rank == 0: print("transport", solver.transport, "own", part.nown, "ghost", part.nall - part.nown, flush=True)
for rep in range(2):
    timeit("no exchange (graph)", lambda n: solver.run_steps(n, with_exchange=False), 502)
    timeit("exchange overlapped (graph)", lambda n: solver.run_steps(n, with_exchange=True, overlap=True), 502)
    timeit("exchange bulk (graph)", lambda n: solver.run_steps(n, with_exchange=True, overlap=False), 502)
    if solver.transport == "ipc":
        timeit("exchange overlapped (stream launches)", lambda n: solver.gpu.run_steps_ipc(n, True, True, True, 0, use_graph=False), 500)
print("rank", rank, "ipc error", solver.gpu.ipc_error() if solver.transport == "ipc" else None, flush=True)
solver.close()
dist.destroy_process_group()
