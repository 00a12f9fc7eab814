"""N>1 path: world_size-2/3 process groups (gloo).  CPU: host logic with the oracle as the
device stand-in.  GPU: the real RankSolver (staged transport), ranks sharing cuda:0."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, extra):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py")] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in out, out[-3000:]


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--files"]), (2, ["--dims", "12,12,12", "--ndomains", "8"])])
def test_multirank_host_logic_gloo(pkg, orc, world, extra):
    _launch(world, extra)


@pytest.mark.gpu
def test_multirank_ranksolver_staged_on_one_gpu(gpu):
    _launch(2, ["--gpu"])
    _launch(3, ["--gpu", "--files"])
