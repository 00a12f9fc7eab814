"""the exchange at size: an N^3 lattice stand-in cut into ND domains, G in-process ranks on ONE GPU (ND / G domains merged per
rank), fused iterations with the overlapped halo exchange (peer copies between the ranks of one process) -- own rows AND
delivered ghost rows of every domain, and the flux, against the C oracle run on the un-partitioned mesh (1e-10).
python tools/big_ranks.py [N] [ND] [G]      (256 16 4: 4.2 M points per rank; the suite's largest is 128^3 on 8 ranks)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 16
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pkg = load_package(); orc = load_oracle()
from cfd_proxy_amd import multigpu as mg
T0 = time.time()
out = {"n": n, "ndomains": nd, "ranks": G, "seconds": {}}


def stage(name, t):
    out["seconds"][name] = round(time.time() - t, 2)
    print(f"[{time.time() - T0:7.1f} s] {name}: {time.time() - t:.1f} s", flush=True)


t = time.time()
whole = pkg.gen_domain(pkg.gen_params(n, ndomains=1), 0)
pkg.fill_var(whole, None, pkg.VAR_HASH, n, n, n)
P = whole.nown
stage("whole mesh", t)
t = time.time()
ref = orc.CpuRef(whole.fpoint, whole.fnormal, whole.pvolume, whole.nown, nthreads=min(16, os.cpu_count() or 1))
truth = ref.gradients(whole.var); ftruth = ref.flux(truth, mode=0); ref.close()
stage("oracle on the whole mesh", t)
t = time.time()
fp = np.asarray(whole.fpoint); fn = np.abs(np.asarray(whole.fnormal)); var = np.asarray(whole.var)
s = np.zeros((P, 7, 3)); CH = 8_000_000
for a in range(0, len(fp), CH):
    p0, p1 = fp[a:a + CH, 0], fp[a:a + CH, 1]
    lo = int(min(p0.min(), p1.min())); hi = int(max(p0.max(), p1.max())) + 1
    for v in range(7):
        av = 0.5 * np.abs(var[p0, v] + var[p1, v])
        for k in range(3):
            c = av * fn[a:a + CH, k]
            s[lo:hi, v, k] += np.bincount(p0 - lo, weights=c, minlength=hi - lo) + np.bincount(p1 - lo, weights=c, minlength=hi - lo)
s /= np.asarray(whole.pvolume)[:P, None, None]
wscale = np.maximum(np.abs(truth), s); wscale[wscale == 0] = 1.0
del s, fp, fn
stage("cancellation scale", t)
fmax = float(np.abs(ftruth[:P]).max())
t = time.time()
gp = pkg.gen_params(n, ndomains=nd)
parts = [mg.build_rank_partition(gp, nd, G, r, via_files=False)[0] for r in range(G)]
pkg.merge_link_group(parts)
stage("rank partitions (generate, merge, link)", t)
out["points_per_rank"] = [int(p.nown) for p in parts]; out["ghost_rows_per_rank"] = [int(p.nall - p.nown) for p in parts]
t = time.time()
gparts = [pkg.GpuPartition(p) for p in parts]
for g in gparts:
    g.set_fusion(True)
stage("plans + uploads", t)
out["tiles_per_rank"] = [int(g.stats["ntiles"]) for g in gparts]; out["boundary_tiles_per_rank"] = [int(g.stats["nbtiles"]) for g in gparts]
t = time.time()
for _ in range(3):
    pkg.group_iteration(gparts, with_exchange=True, overlap=True, with_flux=True)
pkg.group_sync(gparts)
stage("3 fused iterations with exchange", t)
t = time.time()
for _ in range(20):
    pkg.group_iteration(gparts, with_exchange=True, overlap=True, with_flux=True)
pkg.group_sync(gparts)
out["ms_per_iteration_all_ranks_on_one_gpu"] = (time.time() - t) / 20 * 1e3
stage("20 more", t)
t = time.time()
worst = worst_f = 0.0
rows = 0
for r, (p, g) in enumerate(zip(parts, gparts)):
    g.pull_fields()
    for dl, d in enumerate(pkg.rank_domain_list(r, nd, G)):
        dom = pkg.gen_domain(gp, d)
        gid = pkg.gen_global_ids(gp, d, dom.nall)
        back = pkg.merge_scatter(p, dl, dom.nall, p.grad)
        worst = max(worst, float((np.abs(back - truth[gid]) / wscale[gid]).max()))  # ghost rows included
        fb = pkg.merge_scatter(p, dl, dom.nall, p.psd_flux)
        worst_f = max(worst_f, float(np.abs(fb[: dom.nown] - ftruth[gid[: dom.nown]]).max()) / fmax)
        rows += int(dom.nall)
        dom.free()
    g.close()
    print(f"   rank {r}: worst so far {worst:.2e} (gradient rows, ghosts included), {worst_f:.2e} (flux)", flush=True)
stage("compare", t)
out["parity"] = {"rows_compared_ghosts_included": rows, "worst_component_error_over_scale": worst, "flux_inf_norm_ratio": worst_f,
                 "tolerance": 1e-10, "ok": bool(worst <= 1e-10 and worst_f <= 1e-10)}
print(json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"big_ranks_{n}_{nd}_{G}.json"), "w"), indent=1)
sys.exit(0 if out["parity"]["ok"] else 1)
