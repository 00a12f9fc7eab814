"""development helper: the small partitions of the strong-scaling configs (dualgrid.24/.48/.192 on 2/4/8 GPUs: 131 k /
65 k / 33 k points per GPU).  Rank 0's partition on one GPU, iterations without exchange (a rank's compute floor),
over tile sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
for name, world in (("dualgrid.12", 1), ("dualgrid.24", 2), ("dualgrid.48", 4), ("dualgrid.192", 8), ("dualgrid.384", 8)):
    cfg = mg.bench_config(name, world)
    gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
    if world == 1:
        part = mg.build_rank_partition(gp, cfg["ndomains"], 1, 0, via_files=False)[0]
    else:
        parts = [mg.build_rank_partition(gp, cfg["ndomains"], world, r, via_files=False)[0] for r in range(world)]
        reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
        mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
        part = parts[0]
    for tp in [int(x) for x in os.environ.get("TPS", "64,32").split(",")]:
        try:
            g = pkg.GpuPartition(part, tile_points=tp)
            g.set_fusion(True)
            g.run_iterations(200)
            ms = min(g.run_iterations(1000) for _ in range(3)) / 1000
            print(f"{name:13s} rank 0 of {world}: own {part.nown:7d} ghost {part.nall - part.nown:6d} tp {tp:3d} tiles {g.stats['ntiles']:5d} "
                  f"(boundary {g.stats['nbtiles']:4d})  iteration {ms * 1e3:6.2f} us  -> {1e3 / ms:8.0f} it/s per rank", flush=True)
            g.close()
        except Exception as e:
            print(name, tp, "FAILED", repr(e)[:200], flush=True)
