"""the hot path at a size that needs the card: an N^3 lattice stand-in in ONE domain on one GPU (N = 256: 16.8 M points, 116 M
faces; N = 320: 32.8 M points, 228 M faces, a 9.5-GB tile blob) -- plan (device stages), upload, fused iterations from a
hipGraph, EVERY row against the C oracle with the tolerance of SURVEY 8c (cancellation scale computed in chunks), kernel
times.  Refuses when the host's available memory is below what the check needs.  python tools/big_mesh.py [N] [tile_points]; IRREGULAR=1: the
generator's irregular option (capacity classes in launches of their own, long lists in chunks) at that size"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P = n ** 3
avail = 0
for line in open("/proc/meminfo"):
    if line.startswith("MemAvailable"):
        avail = int(line.split()[1]) * 1024
need = int(1800 * P)
print(f"N = {n}: {P} points; host memory available {avail / 1e9:.0f} GB, this check needs about {need / 1e9:.0f} GB", flush=True)
if avail < need:
    raise SystemExit("not enough host memory for the full-row check at this size")
m = load_package(); orc = load_oracle()
out = {"n": n, "points": P}
T0 = time.time()


def stage(name, t):
    out.setdefault("seconds", {})[name] = round(time.time() - t, 2)
    print(f"[{time.time() - T0:7.1f} s] {name}: {time.time() - t:.1f} s", flush=True)


irr = os.environ.get("IRREGULAR", "0") != "0"  # the generator's irregular option (random tetrahedralisation + hubs, scrambled numbering)
out["connectivity"] = "irregular" if irr else "lattice stand-in"
t = time.time()
dom = m.gen_domain(m.gen_params(n, ndomains=1, connectivity=m.CONN_IRREGULAR if irr else 7, numbering=1 if irr else 0), 0)
m.fill_var(dom, None, m.VAR_HASH); stage("generate", t)
out["faces"] = int(dom.nfaces)
t = time.time(); g = m.GpuPartition(dom, tile_points=tp); stage("plan + upload", t)
st = g.stats
out["tiles"] = int(st["ntiles"]); out["blob_GB"] = round(st.get("blob_bytes", 0) / 1e9, 3); out["groups"] = st.get("groups")
g.set_fusion(True)
t = time.time(); g.run_iterations(3, True, 0, use_graph=True); g.sync(); stage("3 fused iterations (graph)", t)
t = time.time(); g.pull_fields(); stage("download", t)
t = time.time()
gr, fl = g.time_kernels(5); fu = g.time_fused(10)
bg = m.algo_bytes_grad(dom.nfaces, dom.nown, 0); bf = m.algo_bytes_flux(dom.nfaces, dom.nown, 0)
out["kernels"] = {"gradient_us": gr * 1e3, "gradient_frac": bg / gr / 1e6 / 8000, "flux_us": fl * 1e3, "flux_frac": bf / fl / 1e6 / 8000,
                  "fused_us": fu * 1e3, "fused_frac": (bg + bf) / fu / 1e6 / 8000, "iterations_per_s": 1e3 / fu}
stage("kernel times", t)
print(json.dumps(out["kernels"]), flush=True)
g.pull_fields()
got_g = dom.grad.copy(); got_f = dom.psd_flux.copy()
g.close()
t = time.time(); ref = orc.CpuRef(dom.fpoint, dom.fnormal, dom.pvolume, dom.nown, nthreads=min(16, os.cpu_count() or 1)); stage("oracle: colours", t)
t = time.time(); rg = ref.gradients(dom.var); stage("oracle: gradients", t)
t = time.time(); rf = ref.flux(rg, mode=0); stage("oracle: flux", t)
ref.close()
# the cancellation scale s_p = sum_f |n_f| 0.5 |var_p0 + var_p1| / V_p, faces in chunks
t = time.time()
fp = np.asarray(dom.fpoint); fn = np.abs(np.asarray(dom.fnormal)); var = np.asarray(dom.var)
s = np.zeros((P, 7, 3))
CH = 8_000_000
for a in range(0, len(fp), CH):
    p0, p1 = fp[a:a + CH, 0], fp[a:a + CH, 1]
    lo = int(min(p0.min(), p1.min())); hi = int(max(p0.max(), p1.max())) + 1  # (a chunk of a file-ordered face list touches a narrow range of points)
    q0, q1 = p0 - lo, p1 - lo
    for v in range(7):
        av = 0.5 * np.abs(var[p0, v] + var[p1, v])
        for k in range(3):
            c = av * fn[a:a + CH, k]
            s[lo:hi, v, k] += np.bincount(q0, weights=c, minlength=hi - lo) + np.bincount(q1, weights=c, minlength=hi - lo)
    if (a // CH) % 4 == 0:
        print(f"   scale: faces {a} of {len(fp)}", flush=True)
s /= np.asarray(dom.pvolume)[:P, None, None]
stage("cancellation scale", t)
t = time.time()
worst = 0.0; gmax = 0.0; dmax = 0.0
RC = 2_000_000
for a in range(0, P, RC):
    r = rg[a:a + RC]; d = np.abs(got_g[a:a + RC] - r)
    sc = np.maximum(np.abs(r), s[a:a + RC]); sc[sc == 0] = 1.0
    worst = max(worst, float((d / sc).max())); gmax = max(gmax, float(np.abs(r).max())); dmax = max(dmax, float(d.max()))
fe = float(np.abs(got_f[:P] - rf[:P]).max() / np.abs(rf[:P]).max())
stage("compare", t)
out["parity"] = {"gradient_rows_compared": P, "worst_component_error_over_scale": worst, "global_inf_norm_ratio": dmax / gmax,
                 "flux_inf_norm_ratio": fe, "tolerance": 1e-10, "ok": bool(worst <= 1e-10 and dmax / gmax <= 1e-10 and fe <= 1e-10)}
print(json.dumps(out), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"big_mesh_{n}{'_irregular' if irr else ''}.json"), "w"), indent=1)
sys.exit(0 if out["parity"]["ok"] else 1)
