"""Genuinely unstructured test meshes (test infrastructure): random points in a box, faces = the edges of their
Delaunay tetrahedralisation -- variable degree (about 15 faces per point, up to 35+), no lattice regularity --
with random normals and volumes; and a partition of such a mesh into domains with ghost points, in the schema
of the dualgrid files (reference src/solver_data.c:96-122, src/comm_data.c:79-112)."""
import numpy as np
from scipy.spatial import Delaunay


def delaunay_mesh(npts, seed=7):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0.0, 1.0, (npts, 3))
    tets = Delaunay(xyz).simplices
    e = np.concatenate([tets[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
    e = np.unique(np.sort(e, axis=1), axis=0).astype(np.int32)
    flip = rng.random(len(e)) < 0.5            # p0/p1 orientation is arbitrary in a face list
    e[flip] = e[flip, ::-1]
    e = e[rng.permutation(len(e))]             # ... and so is the face order
    h = npts ** (-1.0 / 3.0)
    fnormal = rng.normal(size=(len(e), 3)) * h * h
    pvolume = rng.uniform(0.5, 2.0, npts) * h ** 3
    var = 1.0 + 0.01 * ((7 * np.arange(npts)[:, None] + 13 * np.arange(7)[None, :]) % 101) + 0.1 * xyz[:, :1]
    return xyz, e, fnormal, pvolume, var


def partition(xyz, fpoint, fnormal, pvolume, var, ndomains):
    """recursive coordinate bisection into `ndomains` (a power of two) domains.  Returns per domain the arrays of
    a dualgrid file: local faces (every face with at least one owned end; a cut face is stored in both domains),
    owned points first, then ghosts; owner / owner-local id of every ghost; partner lists and counts; and the
    global id of every local point."""
    npts = len(xyz)
    owner = np.zeros(npts, np.int32)
    groups = [np.arange(npts)]
    while len(groups) < ndomains:
        nxt = []
        for g in groups:
            ax = np.argmax(xyz[g].max(0) - xyz[g].min(0))
            order = g[np.argsort(xyz[g, ax], kind="stable")]
            nxt += [order[: len(order) // 2], order[len(order) // 2:]]
        groups = nxt
    for d, g in enumerate(groups):
        owner[g] = d
    own_lists = [np.sort(g) for g in groups]
    local_of = np.full(npts, -1, np.int64)
    for g in own_lists:
        local_of[g] = np.arange(len(g))
    doms = []
    for d in range(ndomains):
        own = own_lists[d]
        touch = (owner[fpoint[:, 0]] == d) | (owner[fpoint[:, 1]] == d)
        f = fpoint[touch]
        ends = np.unique(f)
        ghosts = ends[owner[ends] != d]                       # file order of the ghosts: ascending global id
        gid = np.concatenate([own, ghosts])
        g2l = {int(g): i for i, g in enumerate(gid)}
        lf = np.array([[g2l[int(a)], g2l[int(b)]] for a, b in f], np.int32)
        partners = np.unique(owner[ghosts]).astype(np.int32)
        recvcount = np.zeros(ndomains, np.int32)
        for k in partners:
            recvcount[k] = int((owner[ghosts] == k).sum())
        doms.append(dict(fpoint=lf, fnormal=fnormal[touch], pvolume=pvolume[gid], var=var[gid], nown=len(own), gid=gid,
                         addpoint_owner=owner[ghosts].astype(np.int32), addpoint_idx=local_of[ghosts].astype(np.int32),
                         commpartner=partners, recvcount=recvcount))
    for d in range(ndomains):                                 # what I send to k = what k receives from me
        doms[d]["sendcount"] = np.array([doms[k]["recvcount"][d] for k in range(ndomains)], np.int32)
        # (symmetric neighbourhoods: a cut face gives each side a ghost on the other)
        assert set(doms[d]["commpartner"]) == {k for k in range(ndomains) if doms[d]["sendcount"][k] > 0}
    return doms
