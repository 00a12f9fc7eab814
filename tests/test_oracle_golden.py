"""Pin the oracle: cpu_ref.c (and the independent numpy statement) against the golden
vectors written by the COMPILED reference (tests/golden/make_golden.py)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, TOL, golden_domain, load_golden, rel_err

SINGLE = ["g1_7x6x5", "g1_cart_6x6x6", "g1_one_9x9x9"]


def _tags(fx, prefix):
    return sorted(k for k in fx.files if k.startswith(prefix))


@pytest.mark.parametrize("name", SINGLE)
@pytest.mark.parametrize("nthreads", [1, 2, 3, 5])
def test_oracle_matches_reference_single_domain(orc, name, nthreads):
    fx = load_golden(name)
    fp, fn, vol, var, nown = fx["d0_fpoint"], fx["d0_fnormal"], fx["d0_pvolume"], fx["d0_var"], int(fx["d0_nown"])
    ref = orc.CpuRef(fp, fn, vol, nown, nthreads=nthreads)
    assert ref.check_invariants() == 0
    g = ref.gradients(var)
    ref.close()
    for key in _tags(fx, "grad_comm_free"):
        assert rel_err(orc, g, fx[key], fp, fn, vol, var, nown) <= 1e-12, key
    g_np = orc.np_gradients(fp, fn, vol, var, nown)
    assert rel_err(orc, g_np, fx[_tags(fx, "grad_comm_free")[0]], fp, fn, vol, var, nown) <= 1e-12


@pytest.mark.parametrize("name", SINGLE)
def test_oracle_flux_reference_mode(orc, name):
    """1-thread psd_flux of the reference (class-numbering quirk) is reproduced by
    mode 1 of the oracle and by the numpy statement."""
    fx = load_golden(name)
    fp, fn, vol, var, nown = fx["d0_fpoint"], fx["d0_fnormal"], fx["d0_pvolume"], fx["d0_var"], int(fx["d0_nown"])
    ref = orc.CpuRef(fp, fn, vol, nown, nthreads=1)
    g = ref.gradients(var)
    f = ref.flux(g, mode=1)
    ref.close()
    gold = fx["flux_comm_free_t1_d0"]
    scale = max(np.abs(gold[:nown]).max(), 1e-300)
    assert np.abs(f - gold)[:nown].max() / scale <= 1e-12
    f_np = orc.np_flux(fp, fn, fx["grad_comm_free_t1_d0"], nown, mode=1)
    assert np.abs(f_np - gold)[:nown].max() / scale <= 1e-12


def test_flux_consistent_mode_agrees_between_oracle_and_numpy(orc):
    fx = load_golden("g1_7x6x5")
    fp, fn, vol, var, nown = fx["d0_fpoint"], fx["d0_fnormal"], fx["d0_pvolume"], fx["d0_var"], int(fx["d0_nown"])
    for nt in (1, 2, 4):
        ref = orc.CpuRef(fp, fn, vol, nown, nthreads=nt)
        g = ref.gradients(var)
        f = ref.flux(g, mode=0)
        ref.close()
        f_np = orc.np_flux(fp, fn, g, nown, mode=0)
        assert np.abs(f - f_np)[:nown].max() / np.abs(f_np[:nown]).max() <= 1e-12


@pytest.mark.parametrize("name,nd", [("g2_10x8x6", 2), ("g4_12x10x9", 4)])
def test_oracle_multi_domain_with_pack_unpack(orc, name, nd):
    """gradients per domain + pack/copy/unpack reproduce the reference's mpi_bulk_sync result,
    ghost rows included (message element j from k = j-th ghost owned by k, comm_data.c:167-173)."""
    fx = load_golden(name)
    grads, send_pts = [], []
    # sendindex: what k's ghosts reference in me (reference builds it via MPI, comm_data.c:203-249)
    sendindex = [[None] * nd for _ in range(nd)]
    recvindex = [[None] * nd for _ in range(nd)]
    for d in range(nd):
        owner, idx, nown = fx[f"d{d}_addpoint_owner"], fx[f"d{d}_addpoint_idx"], int(fx[f"d{d}_nown"])
        for k in range(nd):
            sel = np.nonzero(owner == k)[0]
            recvindex[d][k] = nown + sel
            sendindex[k][d] = idx[sel]
            assert len(sel) == fx[f"d{d}_recvcount"][k]
    for d in range(nd):
        fp, fn, vol, var, nown = (fx[f"d{d}_fpoint"], fx[f"d{d}_fnormal"], fx[f"d{d}_pvolume"], fx[f"d{d}_var"],
                                  int(fx[f"d{d}_nown"]))
        sp = np.unique(np.concatenate([sendindex[d][k] for k in range(nd) if k != d])).astype(np.int32)
        for k in range(nd):
            if k != d:
                assert len(sendindex[d][k]) == fx[f"d{d}_sendcount"][k]
        ref = orc.CpuRef(fp, fn, vol, nown, nthreads=2, sendpoints=sp)
        assert ref.check_invariants() == 0
        grads.append(ref.gradients(var))
        ref.close()
    for d in range(nd):  # exchange
        for k in range(nd):
            if k != d and len(sendindex[d][k]):
                orc.unpack(recvindex[k][d], grads[k], orc.pack(sendindex[d][k], grads[d]))
    for d in range(nd):
        fp, fn, vol, var, nown = (fx[f"d{d}_fpoint"], fx[f"d{d}_fnormal"], fx[f"d{d}_pvolume"], fx[f"d{d}_var"],
                                  int(fx[f"d{d}_nown"]))
        for t in (1, 3):
            gold = fx[f"grad_mpi_bulk_sync_t{t}_d{d}"]
            assert rel_err(orc, grads[d], gold, fp, fn, vol, var, nown) <= 1e-12
            # ghost rows are copies of the owner's rows
            scale = np.abs(gold).max()
            assert np.abs(grads[d][nown:] - gold[nown:]).max() / scale <= 1e-12


def test_single_domain_of_a_partitioned_mesh_leaves_ghost_rows_alone(orc):
    fx = load_golden("g4_dom0_alone")
    fp, fn, vol, var, nown = fx["d0_fpoint"], fx["d0_fnormal"], fx["d0_pvolume"], fx["d0_var"], int(fx["d0_nown"])
    gold = fx["grad_alone_0_t1_d0"]
    assert np.all(gold[nown:] == 1.0)  # the reference never writes a ghost row (init value 1.0)
    ref = orc.CpuRef(fp, fn, vol, nown, nthreads=3)
    g = ref.gradients(var)
    ref.close()
    assert np.all(g[nown:] == 1.0)
    assert rel_err(orc, g, gold, fp, fn, vol, var, nown) <= 1e-12


def test_known_answer_linear_field_cartesian(orc):
    """Green-Gauss with 0.5*(f0+f1) is exact for a linear field in the interior of the
    Cartesian lattice (n = h^2 e_axis, V = h^3): grad = the field's slope."""
    fx = load_golden("g1_cart_6x6x6")
    g = fx["grad_comm_free_t1_d0"]
    gid = fx["d0_gid"]
    nx = ny = nz = 6
    x, y, z = gid % nx, (gid // nx) % ny, gid // (nx * ny)
    interior = (x > 0) & (x < nx - 1) & (y > 0) & (y < ny - 1) & (z > 0) & (z < nz - 1)
    for eq in range(7):
        slope = np.array([eq + 1.0, 2.0 * eq - 3.0, 0.5 * eq + 1.0])
        assert np.abs(g[interior, eq, :] - slope).max() <= 1e-11


REF_RAW = os.path.join(ROOT, "oracle", "_ref", "ref_dump_raw")
needs_ref = pytest.mark.skipif(not (os.path.exists(REF_RAW) and os.path.exists("/root/reference/src/gradients.c")),
                               reason="compiled reference only exists in the build container")


def test_reference_binary_links_no_product_code():
    """the pin: oracle/_ref/ref_dump_raw = reference translation units + oracle/ref_dump_raw.c; the
    recipe names nothing under cfd-proxy_amd/ and the binary holds no NetCDF reader of any kind"""
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    start = mk.index("_ref/ref_dump_raw: ")
    recipe = mk[start:mk.index("\n\n", start)]  # the target line and its recipe lines
    assert "-o $@" in recipe and "$(REF)/src/$$f.c" in recipe
    assert "cfd-proxy_amd" not in recipe and "nc_classic" not in recipe and "dropin" not in recipe
    if os.path.exists(REF_RAW):
        r = subprocess.run(["nm", REF_RAW], capture_output=True, text=True)
        assert r.returncode == 0
        names = [ln.split()[-1] for ln in r.stdout.splitlines() if ln.strip()]
        bad = [n for n in names if n.startswith(("nc_", "cfdp_", "get_nc_")) or n in ("read_solver_data", "read_communication_data")]
        assert not bad, bad


@needs_ref
def test_live_reference_agrees_with_fixture(orc):
    """where the compiled reference is available, re-run it on the fixture's stored input arrays
    (raw files: no dualgrid file, no loader of ours) and compare with the committed outputs"""
    fx = load_golden("g1_7x6x5")
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "raw")
        orc.write_raw_domain(raw, 0, fx["d0_fpoint"], fx["d0_fnormal"], fx["d0_pvolume"], int(fx["d0_nown"]), var=fx["d0_var"])
        env = dict(os.environ, OMP_NUM_THREADS="2")
        r = subprocess.run([REF_RAW, "dump", raw, "comm_free", os.path.join(tmp, "out")],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        g = np.fromfile(os.path.join(tmp, "out_grad_0.bin")).reshape(-1, 7, 3)
    assert np.array_equal(g, fx["grad_comm_free_t2_d0"])


@needs_ref
@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/mpiexec"), reason="no mpiexec")
def test_live_reference_two_domains_agrees_with_fixture(orc):
    """the same for the exchanged case: 2 MPI ranks, mpi_bulk_sync, ghost rows included"""
    fx = load_golden("g2_10x8x6")
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "raw")
        for d in range(2):
            orc.write_raw_domain(raw, d, fx[f"d{d}_fpoint"], fx[f"d{d}_fnormal"], fx[f"d{d}_pvolume"],
                                 int(fx[f"d{d}_nown"]), var=fx[f"d{d}_var"], ndomains=2,
                                 commpartner=fx[f"d{d}_commpartner"], sendcount=fx[f"d{d}_sendcount"],
                                 recvcount=fx[f"d{d}_recvcount"], addpoint_owner=fx[f"d{d}_addpoint_owner"],
                                 addpoint_id=fx[f"d{d}_addpoint_idx"])
        env = dict(os.environ, OMP_NUM_THREADS="3")
        r = subprocess.run(["/opt/conda/bin/mpiexec", "-n", "2", REF_RAW, "dump", raw, "mpi_bulk_sync",
                            os.path.join(tmp, "out")], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        for d in range(2):
            g = np.fromfile(os.path.join(tmp, f"out_grad_{d}.bin")).reshape(-1, 7, 3)
            assert np.array_equal(g, fx[f"grad_mpi_bulk_sync_t3_d{d}"])
