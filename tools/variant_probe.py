"""development helper: fused-pass / gradient / flux kernel times of one mesh for a
list of environment variations, all in one process.  CONFIGS="label:K=V,K=V;label2:..." SIZES=64,128"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
out = open(os.environ.get("PROBE_OUT", "gpurun_out/variant_probe.log"), "a")
def log(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.write(s + "\n"); out.flush()
cfgs = []
for c in os.environ.get("CONFIGS", "base:").split(";"):
    label, _, kv = c.partition(":")
    cfgs.append((label, dict(x.split("=") for x in kv.split(",") if x)))
sizes = [int(x) for x in os.environ.get("SIZES", "64,128").split(",")]
reps = int(os.environ.get("REPS", "3"))
for n in sizes:
    gp = m.gen_params(n, ndomains=1); dom = m.gen_domain(gp, 0)
    m.fill_var(dom, None, m.VAR_HASH)
    bg = m.algo_bytes_grad(dom.nfaces, dom.nown, 0); bf = m.algo_bytes_flux(dom.nfaces, dom.nown, 0)
    uniq = 32 * dom.nfaces + 328 * dom.nown
    for label, env in cfgs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            t0 = time.time()
            part = m.GpuPartition(dom, tile_points=int(env.get("TP", "0")))
            st = part.stats
            iters = 100 if n <= 64 else 30
            part.set_fusion(True)
            fu = min(part.time_fused(iters) for _ in range(reps))
            part.set_fusion(False)
            g, f = None, None
            if os.environ.get("SEPARATE", "1") == "1":
                tk = [part.time_kernels(iters) for _ in range(reps)]
                g = min(t[0] for t in tk); f = min(t[1] for t in tk)
            log(f"n {n} {label:24s} tiles {st['ntiles']} dup {st['nfaces_dup']/max(st['nfaces_used'],1):.3f} halo/tile {st['nhalo']/st['ntiles']:.1f} blobMB {st['blob_bytes']/1e6:.1f} "
                f"fused {fu*1e3:7.2f} us  8d-frac {(bg+bf)/fu/1e6/8000:.3f}  uniq-frac {uniq/fu/1e6/8000:.3f}"
                + (f"  grad {g*1e3:7.2f} us frac {bg/g/1e6/8000:.3f}  flux {f*1e3:7.2f} us" if g else "")
                + f"  setup {time.time()-t0:.1f}s")
            part.close()
        except Exception as e:
            log("n", n, label, "FAILED", repr(e)[:300])
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    dom.free()
