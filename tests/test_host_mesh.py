"""Generator, domain merger and tiler (host C side, CPU only).  The tiler tests restate the
run-time invariants of the reference's eval.c / thread_comm.c asserts for GPU tiles
(reference src/eval.c:88-235, src/thread_comm.c:159-205,329-428)."""
import os

import numpy as np
import pytest

from conftest import golden_domain, load_golden


def global_truth(pkg, orc, gp, var_kind):
    """gradient of every lattice point from the un-partitioned mesh (numpy statement)"""
    g1 = pkg.gen_params(gp.nx, gp.ny, gp.nz, ndomains=1, connectivity=gp.connectivity, normals=gp.normals,
                        volumes=gp.volumes, seed=gp.seed)
    d = pkg.gen_domain(g1, 0)
    pkg.fill_var(d, None, var_kind, gp.nx, gp.ny, gp.nz)
    g = orc.np_gradients(d.fpoint, d.fnormal, d.pvolume, d.var, d.nown)
    out = (g, d.var.copy(), d.nfaces)
    d.free()
    return out


@pytest.mark.parametrize("dims,nd,gf", [((9, 8, 7), 2, 0), ((12, 10, 9), 4, 1), ((13, 7, 5), 5, 0), ((16, 12, 10), 12, 0)])
def test_generator_domains_are_consistent(pkg, orc, dims, nd, gf):
    gp = pkg.gen_params(*dims, ndomains=nd, ghost_faces=gf)
    truth, _, nf_global = global_truth(pkg, orc, gp, pkg.VAR_HASH)
    doms = [pkg.gen_domain(gp, d) for d in range(nd)]
    gids = [pkg.gen_global_ids(gp, d, doms[d].nall) for d in range(nd)]
    assert sum(d.nown for d in doms) == dims[0] * dims[1] * dims[2]
    own_gid = np.concatenate([gids[d][: doms[d].nown] for d in range(nd)])
    assert len(np.unique(own_gid)) == len(own_gid)  # every lattice point owned exactly once
    for d, dom in enumerate(doms):
        fp = dom.fpoint
        has_own = (fp[:, 0] < dom.nown) | (fp[:, 1] < dom.nown)
        assert gf or has_own.all()
        if gf:
            assert (~has_own).any()  # ghost-ghost faces present when asked for
        owner, idx = dom.addpoint_owner(), dom.addpoint_id()
        for j in range(dom.nall - dom.nown):  # ghost j really is point idx of its owner
            assert gids[owner[j]][idx[j]] == gids[d][dom.nown + j] and idx[j] < doms[owner[j]].nown
        for k in range(nd):
            assert dom.cd.sendcount[k] == doms[k].cd.recvcount[d]  # symmetric halo
            assert dom.cd.recvcount[k] == np.count_nonzero(owner == k)
        assert d not in dom.partners and sorted(dom.partners) == dom.partners
        # the domain alone reproduces the global gradient on its owned points
        pkg.fill_var(dom, gids[d], pkg.VAR_HASH, *dims)
        g = orc.np_gradients(dom.fpoint, dom.fnormal, dom.pvolume, dom.var, dom.nown)
        assert np.abs(g[: dom.nown] - truth[gids[d][: dom.nown]]).max() <= 1e-12 * np.abs(truth).max()
    for dom in doms:
        dom.free()


@pytest.mark.parametrize("dims,nd,G", [((12, 10, 9), 4, 1), ((12, 10, 9), 4, 2), ((16, 12, 10), 12, 3), ((16, 12, 10), 12, 4),
                                       ((13, 7, 5), 5, 2)])
def test_merge_domains(pkg, orc, dims, nd, G):
    gp = pkg.gen_params(*dims, ndomains=nd, ghost_faces=1)
    truth, _, nf_global = global_truth(pkg, orc, gp, pkg.VAR_HASH)
    doms = [pkg.gen_domain(gp, d) for d in range(nd)]
    gids = [pkg.gen_global_ids(gp, d, doms[d].nall) for d in range(nd)]
    for d in range(nd):
        pkg.fill_var(doms[d], gids[d], pkg.VAR_HASH, *dims)
    parts, mgids = [], []
    for r in range(G):
        f, c = pkg.rank_domains(r, nd, G)
        assert all(pkg.host_lib().cfdp_domain_rank(d, nd, G) == r for d in range(f, f + c))
        part = pkg.merge_domains(doms[f:f + c], list(range(f, f + c)), nd, G, r)
        mi = part.merge_info.contents
        gid = np.full(part.nall, -1, np.int64)
        for dl in range(c):
            l2m = np.ctypeslib.as_array(mi.local2merged[dl], shape=(doms[f + dl].nall,))
            assert np.all((gid[l2m] == -1) | (gid[l2m] == gids[f + dl]))  # one merged id per lattice point
            gid[l2m] = gids[f + dl]
            part.var[l2m] = doms[f + dl].var
            assert np.array_equal(part.pvolume[l2m], doms[f + dl].pvolume)
        assert (gid >= 0).all() and len(np.unique(gid)) == part.nall
        parts.append(part)
        mgids.append(gid)
        # every face with an owned end is present exactly once
        fp = part.fpoint
        key = np.sort(np.stack([gid[fp[:, 0]], gid[fp[:, 1]]], 1), axis=1)
        assert len(np.unique(key, axis=0)) == len(key)
        assert ((fp[:, 0] < part.nown) | (fp[:, 1] < part.nown)).all()
        g = orc.np_gradients(part.fpoint, part.fnormal, part.pvolume, part.var, part.nown)
        assert np.abs(g[: part.nown] - truth[gid[: part.nown]]).max() <= 1e-12 * np.abs(truth).max()
        if G == 1:
            assert part.nall == part.nown and part.nfaces == nf_global and part.partners == []
    pkg.merge_link_group(parts)
    for r, part in enumerate(parts):
        for s in part.partners:
            si, ri = part.sendindex(s), parts[s].recvindex(r)
            assert len(si) == len(ri) > 0
            assert np.array_equal(mgids[r][si], mgids[s][ri])  # element j is the same lattice point on both sides
            assert (si < part.nown).all() and (ri >= parts[s].nown).all()
            assert np.array_equal(ri, np.arange(ri[0], ri[0] + len(ri)))  # a message is one block of ghost rows
    # scatter back to file numbering reproduces owner values in ghost rows
    part = parts[0]
    f, c = pkg.rank_domains(0, nd, G)
    field = np.arange(part.nall * 21, dtype=float).reshape(part.nall, 7, 3)
    for dl in range(c):
        back = pkg.merge_scatter(part, dl, doms[f + dl].nall, field)
        l2m = np.ctypeslib.as_array(part.merge_info.contents.local2merged[dl], shape=(doms[f + dl].nall,))
        assert np.array_equal(back, field[l2m])
    for p in parts:
        p.free()
    for d in doms:
        d.free()


def interpret_plan(plan, var_new, vol_new):
    """what gg_gradient_kernel computes, tile by tile, in numpy (owned rows only)"""
    g = np.full((plan.nall, 7, 3), np.nan)
    for t in range(plan.ntiles):
        td = plan.tile(t)
        fn, inc, ioff, halo = plan.tile_arrays(t)
        ids = np.concatenate([np.arange(td.pstart, td.pstart + td.npts), halo]).astype(np.int64)
        for li in range(td.npts):
            ks, ke = int(ioff[li]), int(ioff[li + 1])
            if ke == ks:
                continue
            w = inc[ks:ke].astype(np.int64)
            nbr, f, neg = w & 0xFFFF, (w >> 16) & 0x7FFF, (w >> 31) & 1
            val = np.where(neg[:, None] == 1, -0.5, 0.5) * (var_new[td.pstart + li][None, :] + var_new[ids[nbr]])
            g[td.pstart + li] = (val[:, :, None] * fn[f][:, None, :]).sum(0) / vol_new[td.pstart + li]
    return g


@pytest.mark.parametrize("tile_points", [8, 32, 128, 1024])
def test_plan_single_partition(pkg, orc, tile_points):
    gp = pkg.gen_params(11, 9, 8, ndomains=1)
    dom = pkg.gen_domain(gp, 0)
    pkg.fill_var(dom, None, pkg.VAR_HASH)
    plan = pkg.Plan(dom, tile_points=tile_points)
    n2o, o2n = plan.new2old, plan.old2new
    assert np.array_equal(np.sort(n2o), np.arange(dom.nall)) and np.array_equal(o2n[n2o], np.arange(dom.nall))
    deg = np.bincount(dom.fpoint.ravel(), minlength=dom.nall)
    cover = np.zeros(dom.nown, int)
    faces_seen = 0
    for t in range(plan.ntiles):
        td = plan.tile(t)
        assert 0 < td.npts <= tile_points and td.npts + td.nhalo <= 65535 and td.nfaces <= 32767
        cover[td.pstart: td.pstart + td.npts] += 1
        fn, inc, ioff, halo = plan.tile_arrays(t)
        assert ioff[0] == 0 and ioff[-1] == td.ninc == len(inc)
        assert np.array_equal(np.diff(ioff.astype(np.int64)), deg[n2o[td.pstart: td.pstart + td.npts]])
        assert ((inc & 0xFFFF) < td.npts + td.nhalo).all() and (((inc >> 16) & 0x7FFF) < td.nfaces).all()
        assert len(np.unique(halo)) == len(halo)
        assert ((halo < td.pstart) | (halo >= td.pstart + td.npts)).all()
        # every tile-local face is referenced once (cross-tile) or twice (internal)
        refs = np.bincount(((inc >> 16) & 0x7FFF).astype(np.int64), minlength=td.nfaces)
        assert refs.min() >= 1 and refs.max() <= 2
        faces_seen += td.nfaces
    assert (cover == 1).all()  # every owned point in exactly one tile (eval.c:126-199 analogue)
    assert faces_seen == plan.nfaces_dup >= plan.nfaces_used == dom.nfaces
    g = interpret_plan(plan, dom.var[n2o], dom.pvolume[n2o])
    back = np.empty_like(g)
    back[n2o] = g
    ref = orc.np_gradients(dom.fpoint, dom.fnormal, dom.pvolume, dom.var, dom.nown)
    assert np.abs(back - ref).max() <= 1e-12 * np.abs(ref).max()
    assert plan.lds_grad == max((plan.tile(t).blob_qw * 16 + (plan.tile(t).npts + plan.tile(t).nhalo) * 64)
                                for t in range(plan.ntiles))
    plan.free()
    dom.free()


def test_plan_with_partners_tiles_send_points_first(pkg, orc):
    fx = load_golden("g4_12x10x9")
    doms = [golden_domain(pkg, fx, d) for d in range(4)]
    pkg.link_raw_group(doms)
    for d, dom in enumerate(doms):
        plan = pkg.Plan(dom, tile_points=32)
        n2o, o2n = plan.new2old, plan.old2new
        send = dom.send_points()
        assert len(send) > 0 and plan.nbtiles > 0
        bnd_end = plan.tile(plan.nbtiles - 1).pstart + plan.tile(plan.nbtiles - 1).npts
        assert (o2n[send] < bnd_end).all()                       # every sent point is in a boundary tile
        assert bnd_end == len(send)                              # ... and boundary tiles hold nothing else
        # ghosts: grouped by partner, message order (zero-copy unpack)
        pos = dom.nown
        for s in range(plan.npartners):
            k = plan.partner[s]
            ri = dom.recvindex(k)
            assert np.array_equal(o2n[ri], np.arange(pos, pos + len(ri)))
            assert plan.recv_off[s] == pos - dom.nown
            si = dom.sendindex(k)
            assert np.array_equal(np.array(plan.send_idx[plan.send_off[s]: plan.send_off[s + 1]]), o2n[si])
            pos += len(ri)
        assert pos == dom.nall
        # boundary tiles of one partner sit next to each other: sorted by the first partner slot they send to (the tile
        # that completes a partner's rows raises its flag: per-partner notification), each tile sends to somebody, and
        # a tile reads ghost rows only of partners it sends to (what the per-partner wait masks rely on)
        slot_of_send = np.full(dom.nown, plan.npartners, np.int64)  # new numbering -> first slot
        sends_to = [set() for _ in range(dom.nown)]
        for s in range(plan.npartners):
            for p_new in plan.send_idx[plan.send_off[s]: plan.send_off[s + 1]]:
                slot_of_send[p_new] = min(slot_of_send[p_new], s)
                sends_to[p_new].add(s)
        keys = []
        for t in range(plan.nbtiles):
            td = plan.tile(t)
            keys.append(int(slot_of_send[td.pstart: td.pstart + td.npts].min()))
            s_t = set().union(*[sends_to[i] for i in range(td.pstart, td.pstart + td.npts)])
            halo = plan.tile_arrays(t)[3]
            ghost = halo[halo >= dom.nown] - dom.nown
            r_t = {int(np.searchsorted(np.array(plan.recv_off[1: plan.npartners + 1]), gi, side="right")) for gi in ghost}
            assert r_t <= s_t, (d, t, r_t, s_t)
        assert keys == sorted(keys) and keys[-1] < plan.npartners, keys
        g = interpret_plan(plan, dom.var[n2o], dom.pvolume[n2o])
        back = np.empty_like(g)
        back[n2o] = g
        gold = fx[f"grad_mpi_bulk_sync_t1_d{d}"]
        assert np.abs(back[: dom.nown] - gold[: dom.nown]).max() <= 1e-12 * np.abs(gold).max()
        assert np.isnan(back[dom.nown:]).all()  # ghost rows are never written by the kernel
        plan.free()
    for dom in doms:
        dom.free()


def test_plan_handles_isolated_points_and_ghost_ghost_faces(pkg, orc):
    """a point without faces is in no tile list's incidences (the reference leaves it alone);
    faces between two ghosts are ignored (rangelist.c:513-523)"""
    fp = np.array([[0, 1], [1, 2], [2, 4], [4, 5], [5, 4]], np.int32)  # point 3 isolated; 4,5 ghosts
    fn = np.arange(15, dtype=float).reshape(5, 3) + 1
    vol = np.arange(6, dtype=float) + 1
    var = np.arange(42, dtype=float).reshape(6, 7) * 0.5 + 1
    dom = pkg.domain_from_arrays(fp, fn, vol, 4, var=var)
    plan = pkg.Plan(dom, tile_points=8)
    assert plan.nfaces_used == 3
    n2o = plan.new2old
    g = interpret_plan(plan, var[n2o], vol[n2o])
    back = np.empty_like(g)
    back[n2o] = g
    ref = orc.np_gradients(fp, fn, vol, var, 4)
    assert np.isnan(back[3]).all()
    assert np.allclose(back[[0, 1, 2]], ref[[0, 1, 2]], rtol=1e-14)
    plan.free()
    dom.free()


def test_domain_clustering_from_commpartner_graph(pkg, tmp_path):
    """real dualgrid files need not number their domains coherently: ranks are then formed along
    the commpartner graph of the files.  On a relabelled graph the clustered map cuts
    far fewer halo points than blocks of ids, and about as few as blocks of the coherent ids."""
    nd, G = 24, 4
    gp = pkg.gen_params(24, 20, 16, ndomains=nd)
    prefix = str(tmp_path / "dualgrid")
    pkg.write_mesh(gp, prefix, 2)
    xadj, adj, wgt = pkg.domain_graph(prefix, 2, nd)
    assert xadj[0] == 0 and xadj[-1] == len(adj) == len(wgt) and (wgt > 0).all()
    for d in range(nd):                                           # symmetric graph, symmetric weights
        for e in range(xadj[d], xadj[d + 1]):
            k = adj[e]
            back = [i for i in range(xadj[k], xadj[k + 1]) if adj[i] == d]
            assert len(back) == 1

    def cut_of(mapping):
        return sum(int(wgt[e]) for d in range(nd) for e in range(xadj[d], xadj[d + 1]) if mapping[adj[e]] != mapping[d]) // 2

    blocks = np.array([pkg.host_lib().cfdp_domain_rank(d, nd, G) for d in range(nd)])
    m, cut = pkg.cluster_domains(xadj, adj, wgt, G)
    assert cut == cut_of(m) and sorted(np.bincount(m, minlength=G)) == [nd // G] * G
    assert cut <= 1.3 * cut_of(blocks)
    # relabel the domains at random: blocks of ids become scattered, the clustering does not care
    rng = np.random.default_rng(7)
    perm = rng.permutation(nd)                                    # new id of old domain d
    inv = np.argsort(perm)
    pxadj, padj, pwgt = [0], [], []
    for new in range(nd):
        old = inv[new]
        for e in range(xadj[old], xadj[old + 1]):
            padj.append(perm[adj[e]])
            pwgt.append(wgt[e])
        pxadj.append(len(padj))
    pm, pcut = pkg.cluster_domains(pxadj, padj, pwgt, G)
    scattered = sum(int(pwgt[e]) for d in range(nd) for e in range(pxadj[d], pxadj[d + 1]) if blocks[padj[e]] != blocks[d]) // 2
    assert pcut <= 1.3 * cut_of(blocks) and pcut < 0.7 * scattered


def test_merge_under_an_explicit_domain_map(pkg, orc):
    """any domain -> rank map works for the merger (here the worst one: round robin)"""
    dims, nd, G = (14, 12, 10), 8, 3
    gp = pkg.gen_params(*dims, ndomains=nd, ghost_faces=1)
    truth, _, _ = global_truth(pkg, orc, gp, pkg.VAR_HASH)
    doms = [pkg.gen_domain(gp, d) for d in range(nd)]
    gids = [pkg.gen_global_ids(gp, d, doms[d].nall) for d in range(nd)]
    for d in range(nd):
        pkg.fill_var(doms[d], gids[d], pkg.VAR_HASH, *dims)
    pkg.set_domain_map([d % G for d in range(nd)], G)
    try:
        parts, mgids = [], []
        for r in range(G):
            ids = pkg.rank_domain_list(r, nd, G)
            assert ids == [d for d in range(nd) if d % G == r]
            part = pkg.merge_domains([doms[d] for d in ids], ids, nd, G, r)
            mi = part.merge_info.contents
            gid = np.full(part.nall, -1, np.int64)
            for dl, d in enumerate(ids):
                l2m = np.ctypeslib.as_array(mi.local2merged[dl], shape=(doms[d].nall,))
                gid[l2m] = gids[d]
                part.var[l2m] = doms[d].var
            assert (gid >= 0).all() and len(np.unique(gid)) == part.nall
            g = orc.np_gradients(part.fpoint, part.fnormal, part.pvolume, part.var, part.nown)
            assert np.abs(g[: part.nown] - truth[gid[: part.nown]]).max() <= 1e-12 * np.abs(truth).max()
            parts.append(part)
            mgids.append(gid)
        pkg.merge_link_group(parts)
        for r, part in enumerate(parts):
            for s in part.partners:
                si, ri = part.sendindex(s), parts[s].recvindex(r)
                assert len(si) == len(ri) > 0 and np.array_equal(mgids[r][si], mgids[s][ri])
    finally:
        pkg.set_domain_map(None)
    for p in parts:
        p.free()
    for d in doms:
        d.free()


_PLAN_HASH = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from __graft_entry__ import load_package
pkg = load_package()
gp = pkg.gen_params(24, 20, 18, ndomains=6)
doms = [pkg.gen_domain(gp, d) for d in range(6)]
part = pkg.merge_domains(doms[:3], [0, 1, 2], 6, 2, 0)
other = pkg.merge_domains(doms[3:], [3, 4, 5], 6, 2, 1)
pkg.merge_link_group([part, other])
plan = pkg.Plan(part, tile_points=32)
h = hashlib.sha256()
h.update(np.ascontiguousarray(plan.new2old).tobytes())
h.update(np.ctypeslib.as_array(plan.p.blob, shape=(plan.p.blob_bytes,)).tobytes())
h.update(np.ctypeslib.as_array(plan.p.halo_idx, shape=(max(plan.p.nhalo_total, 1),)).tobytes())
for t in range(plan.ntiles):
    td = plan.tile(t)
    h.update(np.array([td.pstart, td.npts, td.nhalo, td.nfaces, td.ninc, td.blob_qw, td.halo_off], np.int64).tobytes())
h.update(np.ascontiguousarray(part.fpoint).tobytes())
print("PLAN", plan.ntiles, plan.nbtiles, h.hexdigest())
"""


def test_host_stages_do_not_depend_on_the_thread_count(pkg):
    """the library's own OpenMP regions (generator, merger, point->face CSR, tile graph, tile blobs) produce the same
    merged partition and the same plan, bit for bit, on 1, 3 and 8 threads -- the check that stands in for a thread
    sanitizer on these regions (neither OpenMP runtime of this image is instrumented: cfd-proxy_amd/Makefile, tsan)"""
    import subprocess
    import sys
    from conftest import ROOT
    seen = set()
    for nthreads in ("1", "3", "8"):
        r = subprocess.run([sys.executable, "-c", _PLAN_HASH, ROOT], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, CFDP_HOST_THREADS=nthreads, OMP_NUM_THREADS=nthreads))
        assert r.returncode == 0, r.stdout + r.stderr
        seen.add([l for l in r.stdout.splitlines() if l.startswith("PLAN")][-1])
    assert len(seen) == 1, seen


def test_stored_form_of_gradient_rows(pkg):
    """the two maps between a gradient row as the reference has it and as the device keeps and sends it (csrc/gg_kernels.h:
    gg_a_encode / gg_a_decode; numpy statements in the package): the first six stored doubles are the diagonal and the
    three symmetric sums of the velocity-gradient block -- exactly the operands of the stress formula, src/flux.c:139-173 --
    and a row comes back with 18 of its 21 doubles untouched and the three upper off-diagonals within one rounding of the
    pair's sum"""
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((1000, 7, 3)) * 10.0 ** rng.integers(-6, 7, size=(1000, 1, 1))
    e = pkg.stored_rows(rows)
    g = rows.reshape(-1, 21)
    assert np.array_equal(e[:, :3], g[:, [0, 4, 8]])                      # dvx_dx, dvy_dy, dvz_dz
    assert np.array_equal(e[:, 3:6], g[:, [1, 2, 5]] + g[:, [3, 6, 7]])  # dvx_dy + dvy_dx, dvx_dz + dvz_dx, dvy_dz + dvz_dy
    assert np.array_equal(e[:, 6:10], g[:, [3, 6, 7, 9]]) and np.array_equal(e[:, 10:], g[:, 10:])
    back = pkg.handed_out_rows(e)
    same = [c for c in range(21) if c not in (1, 2, 5)]
    assert np.array_equal(back[:, same], g[:, same])
    ulp = np.spacing(np.abs(e[:, 3:6]))
    assert np.all(np.abs(back[:, [1, 2, 5]] - g[:, [1, 2, 5]]) <= ulp)    # one rounding of the sum, nothing more
    # ... and what is handed out is a fixed point of the two maps as far as the flux is concerned: re-stored, the six
    # numbers the stress reads are within one more rounding
    again = pkg.stored_rows(back)
    assert np.array_equal(again[:, :3], e[:, :3]) and np.all(np.abs(again[:, 3:6] - e[:, 3:6]) <= ulp)
    # the contract under cancellation: the error of an upper off-diagonal is one rounding of the
    # PAIR'S SUM -- absolute -- so a component far smaller than its partner loses its relative precision: 1e-20 beside 1
    # comes back as 0; the partner, the diagonal and everything the flux reads are exact
    tiny = np.zeros((3, 7, 3))
    tiny[:, 0, 1], tiny[:, 1, 0] = [1e-20, -3e-17, 0.75], [1.0, 1.0, 1e-20]   # g1 beside g3
    tiny[:, 0, 2], tiny[:, 2, 0] = [5e-18, 1e-30, 2.0], [-1.0, 4.0, -2.0]     # g2 beside g6
    out = pkg.handed_out_rows(pkg.stored_rows(tiny)).reshape(3, 7, 3)
    assert np.array_equal(out[:, 1, 0], tiny[:, 1, 0]) and np.array_equal(out[:, 2, 0], tiny[:, 2, 0])  # the partners: exact
    assert out[0, 0, 1] == 0.0 and out[1, 0, 1] == 0.0 and out[2, 0, 1] == 0.75   # 1e-20, -3e-17 beside 1: gone; 0.75 beside 1e-20: exact
    assert out[0, 0, 2] == 0.0 and out[1, 0, 2] == 0.0 and out[2, 0, 2] == 2.0
    bound = np.spacing(np.abs(tiny[:, 0, 1] + tiny[:, 1, 0]))
    assert np.all(np.abs(out[:, 0, 1] - tiny[:, 0, 1]) <= bound)


# ------------------------------------------------------------------- irregular meshes (round 6)
@pytest.mark.parametrize("dims,nd", [((12, 10, 9), 4), ((16, 12, 10), 6)])
def test_irregular_generator_domains_are_consistent(pkg, orc, dims, nd):
    """the generator's irregular option (the edge graph of a random tetrahedralisation of the lattice's cubes + hubs,
    scrambled file numbering): every domain file holds exactly the faces of the whole mesh that touch its own points, with
    the same normals; ghosts name their owner's point; halos are symmetric; degrees really vary"""
    gp = pkg.gen_params(*dims, ndomains=nd, connectivity=pkg.CONN_IRREGULAR, numbering=1)
    whole = pkg.gen_domain(pkg.gen_params(*dims, ndomains=1, connectivity=pkg.CONN_IRREGULAR, numbering=0), 0)
    wf = {(int(a), int(b)): tuple(n) for (a, b), n in zip(whole.fpoint, whole.fnormal)}
    deg = np.bincount(whole.fpoint.ravel(), minlength=whole.nown)
    assert deg.max() - deg.min() >= 12 and 5.5 < whole.nfaces / whole.nown < 7.5  # (small meshes: the surface counts)
    doms = [pkg.gen_domain(gp, d) for d in range(nd)]
    gids = [pkg.gen_global_ids(gp, d, doms[d].nall) for d in range(nd)]
    own_gid = np.concatenate([gids[d][: doms[d].nown] for d in range(nd)])
    assert len(np.unique(own_gid)) == len(own_gid) == dims[0] * dims[1] * dims[2]
    scrambled = False
    for d, dom in enumerate(doms):
        g = gids[d]
        scrambled = scrambled or not np.array_equal(g[: dom.nown], np.sort(g[: dom.nown]))
        mine = {(int(g[a]), int(g[b])): tuple(n) for (a, b), n in zip(dom.fpoint, dom.fnormal)}
        own = set(int(x) for x in g[: dom.nown])
        assert mine == {e: n for e, n in wf.items() if e[0] in own or e[1] in own}
        owner, idx = dom.addpoint_owner(), dom.addpoint_id()
        for j in range(dom.nall - dom.nown):
            assert gids[owner[j]][idx[j]] == g[dom.nown + j] and idx[j] < doms[owner[j]].nown
        for k in range(nd):
            assert dom.cd.sendcount[k] == doms[k].cd.recvcount[d]
        assert np.array_equal(dom.pvolume, whole.pvolume[g])
    assert scrambled
    for dom in doms:
        dom.free()
    whole.free()


def test_tiles_of_irregular_meshes_fill_up_and_hub_tiles_go_last(pkg):
    """two-level tile budgets and launch groups (host/tiling.c 3e): on an irregular graph the small image alone would close
    tiles at ~2/3 of their points; a tile it closes below 7/8 full goes on under the large image.  A tile no fixed
    capacity holds (a hub of hundreds of faces) is moved to the end, into a group of its own; the lattice stand-in stays
    one group of small tiles"""
    def classes(plan, tp=64):
        return [pkg.host_lib().cfdp_tile_class_of(tp, plan.tile(t).npts + plan.tile(t).nhalo, plan.tile(t).blob_qw * 16)
                for t in range(plan.ntiles)]
    lat = pkg.gen_domain(pkg.gen_params(24, 20, 18, ndomains=1), 0)
    plan = pkg.Plan(lat)
    assert plan.ngroups == 1 and plan.group_begin[0] == 0 and plan.group_begin[1] == plan.ntiles and plan.group_class[0] == 0
    assert set(classes(plan)) == {0}
    plan.free()
    lat.free()
    irr = pkg.gen_domain(pkg.gen_params(32, 32, 32, ndomains=1, connectivity=pkg.CONN_IRREGULAR, numbering=1), 0)
    plan = pkg.Plan(irr)
    cls = classes(plan)
    assert plan.ngroups == 1 and plan.group_class[0] == 1 and 2 not in cls and cls.count(1) > plan.ntiles // 2
    assert irr.nown / plan.ntiles > 56  # ... and the tiles are full
    os.environ["CFDP_TILE_BUDGET"] = "1"  # the small image only
    try:
        small = pkg.Plan(irr)
    finally:
        del os.environ["CFDP_TILE_BUDGET"]
    assert irr.nown / small.ntiles < 0.9 * irr.nown / plan.ntiles and set(classes(small)) == {0}
    small.free()
    plan.free()
    irr.free()
    # a star of 300 leaves + a chain through them: the hub's tile fits no fixed capacity
    n = 301
    fp = np.concatenate([np.stack([np.zeros(300, np.int32), np.arange(1, n, dtype=np.int32)], 1),
                         np.stack([np.arange(1, n - 1, dtype=np.int32), np.arange(2, n, dtype=np.int32)], 1)]).astype(np.int32)
    rng = np.random.default_rng(5)
    dom = pkg.domain_from_arrays(fp, rng.standard_normal((len(fp), 3)), rng.uniform(0.5, 2.0, n), n)
    plan = pkg.Plan(dom)
    cls = classes(plan)
    assert plan.ngroups == 2 and plan.group_class[1] == 2 and plan.group_class[0] < 2
    assert plan.group_begin[0] == 0 and plan.group_begin[2] == plan.ntiles
    assert all(c < 2 for c in cls[: plan.group_begin[1]]) and all(c == 2 for c in cls[plan.group_begin[1]:])
    cover = np.zeros(n, int)
    for t in range(plan.ntiles):
        cover[plan.tile(t).pstart: plan.tile(t).pstart + plan.tile(t).npts] += 1
    assert (cover == 1).all()
    plan.free()
    dom.free()


def test_points_of_a_tile_are_ordered_by_degree(pkg):
    """inside a tile, points of like degree sit next to each other (host/tiling.c 3d): a wave is busy for as long as its
    longest incidence list takes"""
    irr = pkg.gen_domain(pkg.gen_params(20, 18, 16, ndomains=1, connectivity=pkg.CONN_IRREGULAR), 0)
    plan = pkg.Plan(irr)
    deg = np.ctypeslib.as_array(plan.p.degree, shape=(plan.nown,))
    for t in range(plan.ntiles):
        td = plan.tile(t)
        d = deg[td.pstart: td.pstart + td.npts]
        assert (np.diff(d) <= 0).all(), t
    assert np.array_equal(deg, np.bincount(irr.fpoint.ravel(), minlength=irr.nown)[plan.new2old[: irr.nown]])
    plan.free()
    irr.free()


def test_long_incidence_lists_are_cut_into_chunks_for_helper_lane_groups(pkg):
    """cfdproxy_host.h, long incidence lists: a list of more than 32 entries is cut into chunks of <= 28, the point's own lane group
    takes the first, helper lane groups of the same tile (slots behind its points, which own no row) the others.  In the blob:
    the chunk count in the top byte of the point's offsets word, a helper table + scratch behind the offsets of a tile that has
    helpers and nothing behind the offsets of one that has none; points + helpers fit the tile's lane groups; the chunks of a
    list cover it exactly once"""
    irr = pkg.gen_domain(pkg.gen_params(24, 20, 18, ndomains=1, connectivity=pkg.CONN_IRREGULAR, numbering=1), 0)
    deg_file = np.bincount(irr.fpoint.ravel(), minlength=irr.nown)
    assert (deg_file > 32).sum() >= 4
    plan = pkg.Plan(irr)
    blob = np.ctypeslib.as_array(plan.p.blob, shape=(plan.blob_bytes,))
    pad16 = lambda n: (n + 15) & ~15
    cut = 0
    for t in range(plan.ntiles):
        td = plan.tile(t)
        plane = pad16(td.nfaces * 8)
        base = td.blob_off * 16 + 3 * plane + pad16(td.ninc * 4)
        raw = blob[base: base + (td.npts + 1) * 4].view(np.uint32)
        start, nch = raw & 0xFFFFFF, (raw >> 24) + 1
        deg = np.diff(start.astype(np.int64))
        assert np.array_equal(deg, deg_file[plan.new2old[td.pstart: td.pstart + td.npts]])
        behind = td.blob_qw * 16 - (3 * plane + pad16(td.ninc * 4) + pad16((td.npts + 1) * 4))
        want = [pkg.host_lib().cfdp_list_chunks_of(int(d), 64) for d in deg]
        nh = sum(want) - td.npts
        if nh == 0:
            assert behind == 0 and (nch[:-1] == 1).all()
            continue
        assert td.npts + nh <= 64 and list(nch[:-1]) == want and behind == pad16(4 * (1 + nh)) + 192 * nh
        tab = blob[base + pad16((td.npts + 1) * 4):][: 4 * (1 + nh)].view(np.uint32)
        assert tab[0] == nh
        seen = {}
        for w in tab[1:]:
            li, c = int(w & 0xFFFF), int(w >> 16)
            assert 0 <= li < td.npts and 1 <= c < want[li]
            seen.setdefault(li, []).append(c)
        for li, cs in seen.items():
            assert cs == list(range(1, want[li])) and want[li] <= 16 and -(-int(deg[li]) // want[li]) <= 28
        assert not blob[base + pad16((td.npts + 1) * 4) + pad16(4 * (1 + nh)): td.blob_off * 16 + td.blob_qw * 16].any()  # the scratch: zeros
        cut += len(seen)
    assert cut == (deg_file > 32).sum()
    os.environ["CFDP_SPLIT_LISTS"] = "0"
    try:
        whole = pkg.Plan(irr)
    finally:
        del os.environ["CFDP_SPLIT_LISTS"]
    assert whole.blob_bytes < plan.blob_bytes
    whole.free()
    plan.free()
    irr.free()


@pytest.mark.parametrize("tile_points", [8, 16, 32, 64, 128])
def test_plans_of_irregular_graphs_at_every_tile_size(pkg, orc, tile_points):
    """the tiler on graphs that are not lattices -- the generator's irregular mesh (with hubs) and a random multigraph with
    isolated points, parallel faces and a few points of 40-90 faces -- at every tile size: every owned point in exactly one
    tile, points + helper lane groups within the tile's lane groups, launch groups contiguous and classed as their tiles
    are, the chunks of a long list covering it exactly once, and the plan INTERPRETED in numpy (tile by tile, incidence word
    by incidence word) equal to the gradient of the mesh"""
    rng = np.random.default_rng(17 + tile_points)
    n = 1500
    deg_target = rng.integers(0, 12, n)
    deg_target[rng.choice(n, 12, replace=False)] = rng.integers(40, 90, 12)
    ends = np.repeat(np.arange(n), deg_target)
    rng.shuffle(ends)
    fp = np.stack([ends, rng.integers(0, n, len(ends))], 1).astype(np.int32)
    fp = fp[fp[:, 0] != fp[:, 1]]
    rnd = pkg.domain_from_arrays(fp, rng.standard_normal((len(fp), 3)), rng.uniform(0.5, 2.0, n), n, var=rng.standard_normal((n, 7)) + 2.0)
    irr = pkg.gen_domain(pkg.gen_params(14, 12, 10, ndomains=1, connectivity=pkg.CONN_IRREGULAR, numbering=1), 0)
    pkg.fill_var(irr, None, pkg.VAR_HASH)
    pad16 = lambda x: (x + 15) & ~15
    for dom in (rnd, irr):
        plan = pkg.Plan(dom, tile_points=tile_points)
        n2o = plan.new2old
        deg = np.bincount(dom.fpoint.ravel(), minlength=dom.nall)
        blob = np.ctypeslib.as_array(plan.p.blob, shape=(plan.blob_bytes,))
        cover = np.zeros(dom.nown, int)
        cls = []
        for t in range(plan.ntiles):
            td = plan.tile(t)
            cover[td.pstart: td.pstart + td.npts] += 1
            plane = pad16(td.nfaces * 8)
            base = td.blob_off * 16 + 3 * plane + pad16(td.ninc * 4)
            raw = blob[base: base + (td.npts + 1) * 4].view(np.uint32)
            nch = (raw[:-1] >> 24).astype(int) + 1
            d = np.diff((raw & 0xFFFFFF).astype(np.int64))
            assert np.array_equal(d, deg[n2o[td.pstart: td.pstart + td.npts]])
            nh = int(nch.sum() - td.npts)
            assert 0 < td.npts and td.npts + nh <= tile_points
            behind = td.blob_qw * 16 - (3 * plane + pad16(td.ninc * 4) + pad16((td.npts + 1) * 4))
            assert behind == (pad16(4 * (1 + nh)) + 192 * nh if nh else 0)
            for li in np.nonzero(nch > 1)[0]:
                L = -(-int(d[li]) // int(nch[li]))  # chunk length: the chunks [c L, min((c + 1) L, deg)) cover the list once
                assert d[li] > 32 and (nch[li] - 1) * L < d[li] <= nch[li] * L
            cls.append(pkg.host_lib().cfdp_tile_class_of(tile_points, td.npts + td.nhalo, td.blob_qw * 16))
        assert (cover == 1).all()
        gb = list(plan.group_begin)[: plan.ngroups + 1]
        assert gb[0] == plan.nbtiles and gb[-1] == plan.ntiles and gb == sorted(gb)
        for k in range(plan.ngroups):
            assert max(cls[gb[k]: gb[k + 1]], default=0) == plan.group_class[k]
        if plan.ngroups > 1:  # tiles no fixed capacity holds go last
            assert plan.group_class[plan.ngroups - 1] == 2 and all(c < 2 for c in cls[gb[0]: gb[plan.ngroups - 1]])
        g = interpret_plan(plan, dom.var[n2o], dom.pvolume[n2o])
        back = np.empty_like(g)
        back[n2o] = g
        ref = orc.np_gradients(dom.fpoint, dom.fnormal, dom.pvolume, dom.var, dom.nown)
        has = deg[: dom.nown] > 0
        assert np.abs(back[: dom.nown][has] - ref[: dom.nown][has]).max() <= 1e-12 * np.abs(ref[: dom.nown][has]).max()
        plan.free()
    rnd.free()
    irr.free()
