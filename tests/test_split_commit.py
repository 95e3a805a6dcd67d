"""One large commitment split across ranks (vpin_amd.dist.SplitCommit; BASELINE configs[4]).

CPU (gloo, world 2 and 4): the orchestration -- broadcast of (rx, ry), block order, all-gather -- with the oracle's
Pippenger as each rank's commitment engine (test-side only); the gathered rows must equal the unsplit commitment.
GPU (2 processes sharing the one MI355X, gloo for the exchange): the wired path -- rank 0 proves a 64-op point-mult
instance through vpin_snark_prove_dev with the split hooks installed, rank 1 serves its block with
vpin_spark_derefs_commit_rows -- and the SNARK bytes must equal the single-rank proof's.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(rank, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS=str(max(1, (os.cpu_count() or 2) // world)))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


class OracleEngine:
    """each rank's block through the oracle: Z is a deterministic function of (rx, ry), like the derefs polynomial"""

    def __init__(self, Ls, Rs):
        import oracle_lib as O
        self.O, self.Ls, self.Rs = O, Ls, Rs
        _, self.og = O.gens_stream_xyzt(Rs + 2, b"gens_r1cs_eval")

    def poly(self, rx, ry):
        import pymodel as M
        a = sum(M.table_to_ints(rx)) % M.Q
        b = sum(M.table_to_ints(ry)) % M.Q
        return M.ints_to_table([(a * (i + 1) + b * (i * i + 3)) % M.Q if i % 5 else 0 for i in range(self.Ls * self.Rs)])

    def rows(self, Z, row0, nrows):
        zero = np.zeros((nrows, 4), dtype=np.uint64)
        return self.O.hyrax_commit(Z[row0 * self.Rs:(row0 + nrows) * self.Rs], nrows, zero, self.og, self.Rs + 1, threads=1)

    def commit_rows(self, z_handle, L, row0, nrows):
        return self.rows(z_handle, row0, nrows)

    def derefs_commit_rows(self, rx, ry, row0, nrows):
        return self.rows(self.poly(rx, ry), row0, nrows)


def _cpu_worker(rank, world, port, q, Ls, Rs):
    _env(rank, world, port)
    import pymodel as M
    from vpin_amd.dist import Group, SplitCommit
    grp = Group(backend="gloo")
    eng = OracleEngine(Ls, Rs)
    sc = SplitCommit(grp, eng, owner=0)
    ok = True
    if rank == 0:
        rx, ry = M.ints_to_table([3, 5, 7]), M.ints_to_table([11, 13])
        Z = eng.poly(rx, ry)
        for _ in range(2):  # two proofs in a row reuse the channel
            sc.begin(rx, ry)
            got = sc.commit(Z, Ls, Rs)
            ok = ok and np.array_equal(got, eng.rows(Z, 0, Ls))
        sc.stop()
    else:
        served = 0
        while sc.serve_one(Ls):
            served += 1
        ok = served == 2
    oks = grp.gather_objects(bool(ok))
    if rank == 0:
        q.put(all(oks))
    grp.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_split_commit_protocol_gloo_cpu(world):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_cpu_worker, args=(r, world, port, q, 8, 16)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_block_partition():
    from vpin_amd.dist import SplitCommit
    assert [SplitCommit.block(16, r, 4) for r in range(4)] == [(0, 4), (4, 4), (8, 4), (12, 4)]


def _gpu_worker(rank, world, port, q, n_ops):
    _env(rank, world, port)
    import vpin_amd
    from vpin_amd import gadgets as G
    from vpin_amd.dist import Group, SplitCommit, SplitEngine
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    grp = Group(backend="gloo")  # the ranks share one GPU here: the exchange goes over gloo, the arithmetic is the library's
    ok, info = True, ""
    with vpin_amd.Context(0) as ctx:
        g = ctx.gadget_point_mult_dev(*G.synthetic_mult_inputs("A", n_ops))
        dec, comm = g.spark_encode()
        N = 1
        while N < 5260 * n_ops:
            N *= 2
        ell = (N.bit_length() - 1) + 3
        L = 1 << (ell // 2)
        sc = SplitCommit(grp, SplitEngine(ctx, decomm=dec, derefs_ell=ell), owner=0)
        if rank == 0:
            ref = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, seed_c, seed_p)
            ctx.set_split(sc, min_len=1 << 10)
            got = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, seed_c, seed_p)
            ctx.set_split(None)
            sc.stop()
            ok = got["proof"] == ref["proof"] and len(ref["proof"]) > 0 and ctx._split_error is None
            info = f"{len(ref['proof'])} bytes"
        else:
            served = 0
            while sc.serve_one(L):
                served += 1
            ok = served == 1
        dec.free()
        g.free()
    oks = grp.gather_objects((bool(ok), info))
    if rank == 0:
        q.put(oks)
    grp.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_split_derefs_commitment_processes_sharing_one_gpu(world):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_gpu_worker, args=(r, world, port, q, 64)) for r in range(world)]
    for p in procs:
        p.start()
    oks = q.get(timeout=500)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[0] for o in oks), oks
