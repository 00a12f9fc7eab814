"""development helper (CPU only): what the tiler makes of a mesh -- tiles, points per tile, rows a tile stages, blob
bytes, face duplication, and how many tiles fall into each capacity class of the fused pass -- on the lattice stand-in
and on irregular meshes.  MESH=lattice:64 | delaunay:262144 | irregular:64 (the generator's irregular option);
TPS=64,...  The Delaunay mesh (scipy, ~11 s for 262144 points) is cached under /tmp."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
m = load_package()


def mesh(spec):
    kind, _, n = spec.partition(":")
    n = int(n or 64)
    if kind == "lattice":
        gp = m.gen_params(n, ndomains=1)
        return m.gen_domain(gp, 0)
    if kind == "irregular":
        gp = m.gen_params(n, ndomains=1, connectivity=m.CONN_IRREGULAR)
        return m.gen_domain(gp, 0)
    if kind == "delaunay":
        path = f"/tmp/delaunay_{n}.npz"
        if not os.path.exists(path):
            from unstructured import delaunay_mesh
            xyz, fp, fn, vol, var = delaunay_mesh(n)
            np.savez(path, fp=fp, fn=fn, vol=vol, var=var)
        z = np.load(path)
        return m.domain_from_arrays(z["fp"], z["fn"], z["vol"], n, var=z["var"])
    raise SystemExit("MESH=lattice:N | delaunay:N | irregular:N")


def stats(dom, tp):
    t0 = time.time()
    pl = m.Plan(dom, tile_points=tp)
    dt = time.time() - t0
    nt = pl.ntiles
    td = pl.p.tiles
    npts = np.array([td[t].npts for t in range(nt)]); rows = np.array([td[t].npts + td[t].nhalo for t in range(nt)])
    blob = np.array([td[t].blob_qw * 16 for t in range(nt)])
    s = (blob <= 20480) & (rows <= 192)
    mm = ~s & (blob <= 24576) & (rows <= 256)
    deg = np.ctypeslib.as_array(pl.p.degree, shape=(pl.nown,))
    print(f"tp {tp}: {nt} tiles ({dt:.2f} s)  points/tile {npts.mean():.1f} (min {npts.min()})  rows/tile {rows.mean():.1f} (max {rows.max()})  "
          f"blob/tile {blob.mean():.0f} (max {blob.max()})  dup {pl.nfaces_dup / pl.nfaces_used:.3f}  blob total {pl.blob_bytes / 1e6:.1f} MB  "
          f"halo rows/own point {(rows.sum() - npts.sum()) / npts.sum():.2f}  classes S/M/other {s.sum()}/{mm.sum()}/{nt - s.sum() - mm.sum()}  "
          f"lanes busy {npts.sum() / (nt * tp):.3f}  degree mean {deg.mean():.2f} max {deg.max()}", flush=True)
    pl.free()


if __name__ == "__main__":
    for spec in os.environ.get("MESH", "lattice:64").split(","):
        dom = mesh(spec)
        print(f"{spec}: {dom.nown} points, {dom.nfaces} faces", flush=True)
        for tp in [int(x) for x in os.environ.get("TPS", "64").split(",")]:
            stats(dom, tp)
