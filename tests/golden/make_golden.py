#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the COMPILED reference (oracle/_ref/ref_dump_raw).

Runs only where /root/reference exists (the build container).  Inputs are meshes from our
own deterministic generator (the reference ships no meshes and no tests); the expected
outputs are what the reference's own compute_gradients_gg_* / compute_psd_flux write into
sd->grad / sd->psd_flux, dumped by oracle/ref_dump_raw.c -- a binary whose link line holds the
reference's translation units, that driver, MPICH and libm, and nothing of the product: the
input arrays reach it as raw files written below from the numpy arrays the fixture stores.
Each fixture stores inputs AND outputs, so tests never need the reference or the generator to
agree with today's code.

    python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import build, load_oracle, load_package  # noqa: E402

MPIEXEC = "/opt/conda/bin/mpiexec"
REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump_raw")
OUT = os.environ.get("GOLDEN_OUT", os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # name, dims, ndomains, ghost_faces, connectivity, var kind, [(variant, ranks, threads)]
    ("g1_7x6x5", (7, 6, 5), 1, 0, 7, "hash", [("comm_free", 1, 1), ("comm_free", 1, 2), ("comm_free", 1, 4)]),
    ("g1_cart_6x6x6", (6, 6, 6), 1, 0, 3, "linear", [("comm_free", 1, 1), ("comm_free", 1, 3)]),
    ("g2_10x8x6", (10, 8, 6), 2, 1, 7, "hash", [("mpi_bulk_sync", 2, 1), ("mpi_bulk_sync", 2, 3)]),
    ("g4_12x10x9", (12, 10, 9), 4, 0, 7, "hash", [("mpi_bulk_sync", 4, 1), ("mpi_bulk_sync", 4, 3)]),
    ("g1_one_9x9x9", (9, 9, 9), 1, 0, 7, "one", [("comm_free", 1, 1), ("comm_free", 1, 2)]),
    # a single domain of a 4-domain mesh run alone: ghost rows keep their initial 1.0
    ("g4_dom0_alone", (12, 10, 9), 4, 0, 7, "hash", [("alone:0", 1, 1), ("alone:0", 1, 2)]),
]


def main():
    build()
    pkg = load_package()
    orc = load_oracle()
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/ref_dump_raw missing (needs /root/reference)")
    for name, dims, nd, gf, conn, vk, runs in CASES:
        gp = pkg.gen_params(*dims, ndomains=nd, ghost_faces=gf, connectivity=conn,
                            normals=0 if conn == 3 else 1, volumes=0 if conn == 3 else 1)
        kind = {"hash": pkg.VAR_HASH, "one": pkg.VAR_ONE, "linear": pkg.VAR_LINEAR}[vk]
        with tempfile.TemporaryDirectory() as tmp:
            prefix = os.path.join(tmp, "dualgrid")
            pkg.write_mesh(gp, prefix, 2)
            fx = {"dims": np.array(dims), "ndomains": nd, "connectivity": conn, "var_kind": vk}
            doms = []
            for d in range(nd):
                dom = pkg.load_domain(prefix, d, 2)
                gid = pkg.gen_global_ids(gp, d, dom.nall)
                pkg.fill_var(dom, gid, kind, *dims)
                fx[f"d{d}_fpoint"] = dom.fpoint.copy()
                fx[f"d{d}_fnormal"] = dom.fnormal.copy()
                fx[f"d{d}_pvolume"] = dom.pvolume.copy()
                fx[f"d{d}_var"] = dom.var.copy()
                fx[f"d{d}_gid"] = gid
                fx[f"d{d}_nown"] = dom.nown
                if nd > 1:
                    fx[f"d{d}_addpoint_owner"] = dom.addpoint_owner().copy()
                    fx[f"d{d}_addpoint_idx"] = dom.addpoint_id().copy()
                    fx[f"d{d}_commpartner"] = np.array(dom.partners, np.int32)
                    fx[f"d{d}_sendcount"] = np.array([dom.cd.sendcount[k] for k in range(nd)], np.int32)
                    fx[f"d{d}_recvcount"] = np.array([dom.cd.recvcount[k] for k in range(nd)], np.int32)
                doms.append(dom)
            # the reference binary sees these arrays and nothing else (no dualgrid file, no loader of ours)
            for d in range(nd):
                comm = {} if nd == 1 else dict(
                    commpartner=fx[f"d{d}_commpartner"], sendcount=fx[f"d{d}_sendcount"],
                    recvcount=fx[f"d{d}_recvcount"], addpoint_owner=fx[f"d{d}_addpoint_owner"],
                    addpoint_id=fx[f"d{d}_addpoint_idx"])
                orc.write_raw_domain(os.path.join(tmp, "raw"), d, fx[f"d{d}_fpoint"], fx[f"d{d}_fnormal"],
                                     fx[f"d{d}_pvolume"], fx[f"d{d}_nown"],
                                     var=None if vk == "one" else fx[f"d{d}_var"], ndomains=nd, **comm)
            raw = os.path.join(tmp, "raw")
            for variant, ranks, threads in runs:
                env = dict(os.environ, OMP_NUM_THREADS=str(threads))
                outp = os.path.join(tmp, f"out_{variant.replace(':', '_')}_{threads}")
                if variant.startswith("alone:"):
                    d0 = int(variant.split(":")[1])
                    env["REF_DUMP_DOMAIN"] = str(d0)
                    cmd = [REF, "dump", raw, "comm_free", outp]
                    which = [d0]
                elif ranks == 1:
                    cmd = [REF, "dump", raw, variant, outp]
                    which = [0]
                else:
                    cmd = [MPIEXEC, "-n", str(ranks), REF, "dump", raw, variant, outp]
                    which = list(range(ranks))
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                if r.returncode:
                    raise SystemExit(f"{name} {variant} failed:\n{r.stdout}\n{r.stderr}")
                tag = f"{variant.replace(':', '_')}_t{threads}"
                for d in which:
                    fx[f"grad_{tag}_d{d}"] = np.fromfile(f"{outp}_grad_{d}.bin").reshape(-1, 7, 3)
                    if threads == 1:  # psd_flux is thread-count dependent in the reference
                        fx[f"flux_{tag}_d{d}"] = np.fromfile(f"{outp}_flux_{d}.bin").reshape(-1, 3)
            for dom in doms:
                dom.free()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, keys: {len(fx)}")


if __name__ == "__main__":
    main()
