"""One sum-check split across ranks (vpin_amd.dist.sharded_cubic_sumcheck): strided
shards, per-round all-gather of the 96-byte partials, modular sum.  The CPU test runs world_size 2 and 4 over gloo
with the oracle's round function as the local engine (test-side only) and compares every round's (e0, e2, e3) and
the final values with the unsharded oracle; the GPU test runs two processes on the one MI355X with the HIP round
kernels as the local engine and compares with the same kernels on the whole tables."""
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


def _tables(seed, n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pymodel as M
    rng = np.random.default_rng(seed)
    vals = [[int(rng.integers(0, 2**62)) ** 4 % M.Q for _ in range(n)] for _ in range(4)]
    ch = [int(rng.integers(0, 2**62)) ** 4 % M.Q for _ in range(n.bit_length() - 1)]
    return [M.ints_to_table(v) for v in vals], M.ints_to_table(ch)


class OracleOps:
    """local engine on numpy tables, through the CPU oracle (tests only)"""

    def __init__(self):
        import oracle_lib as O
        self.O = O

    def length(self, t):
        return t[0].shape[0]

    def round(self, tabs):
        return self.O.sc_cubic_round(*[t[0] for t in tabs])

    def bind(self, tabs, r):
        for t in tabs:
            t[0] = self.O.bound_top(t[0], r)

    def bind_round(self, tabs, r):
        self.bind(tabs, r)
        return self.round(tabs)

    def read(self, t):
        return t[0]


class GpuOps:
    """local engine on device tables: the kernels a single GPU runs"""

    def __init__(self, ctx):
        self.ctx = ctx

    def length(self, t):
        return len(t)

    def round(self, tabs):
        return self.ctx.sc_cubic_round(*tabs)

    def bind(self, tabs, r):
        self.ctx.sc_bind(list(tabs), r)

    def bind_round(self, tabs, r):
        return self.ctx.sc_cubic_bind_round(*tabs, r)

    def read(self, t):
        return t.read()


def _unsharded(ops, tabs, ch):
    """the same rounds on whole tables, one rank"""
    class One:
        world = 1

        def gather_objects(self, o):
            return [o]
    from vpin_amd.dist import sharded_cubic_sumcheck
    return sharded_cubic_sumcheck(One(), ops, tabs, ch)


def _worker(rank, world, port, q, n, use_gpu):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from vpin_amd.dist import Group, sharded_cubic_sumcheck, strided_shard
    grp = Group(backend="gloo")
    full, ch = _tables(77, n)
    if use_gpu:
        import vpin_amd
        ctx = vpin_amd.Context(0)
        ops = GpuOps(ctx)
        shards = [ctx.upload(strided_shard(t, rank, world)) for t in full]
    else:
        ops = OracleOps()
        shards = [[strided_shard(t, rank, world)] for t in full]
    evals, final = sharded_cubic_sumcheck(grp, ops, shards, ch)
    if rank == 0:
        if use_gpu:
            exp_e, exp_f = _unsharded(ops, [ctx.upload(t) for t in full], ch)
        else:
            exp_e, exp_f = _unsharded(ops, [[t.copy()] for t in full], ch)
        ok = len(evals) == len(exp_e) and all(np.array_equal(a, b) for a, b in zip(evals, exp_e)) and np.array_equal(final, exp_f)
        q.put((ok, len(evals)))
    grp.barrier()
    grp.close()


def _commit_worker(rank, world, port, q, Ls, Rs, use_gpu):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    import pymodel as M
    from vpin_amd.dist import Group, sharded_hyrax_commit
    grp = Group(backend="gloo")
    rng = np.random.default_rng(5)
    Z = M.ints_to_table([int(rng.integers(0, 2**62)) ** 4 % M.Q if rng.random() > 0.3 else 0 for _ in range(Ls * Rs)])
    blinds = M.ints_to_table([int(rng.integers(0, 2**62)) ** 4 % M.Q for _ in range(Ls)])
    xyzt, og = O.gens_stream_xyzt(Rs + 2)
    per = Ls // world
    zl, bl = Z[rank * per * Rs:(rank + 1) * per * Rs], blinds[rank * per:(rank + 1) * per]
    if use_gpu:
        import vpin_amd
        ctx = vpin_amd.Context(0)
        g = ctx.gens_create(xyzt)
        fn = lambda z, b: ctx.hyrax_commit(g, ctx.upload(z), b, Rs + 1)
    else:
        fn = lambda z, b: O.hyrax_commit(z, per, b, og, Rs + 1, threads=1)
    got = sharded_hyrax_commit(grp, fn, zl, bl)
    if rank == 0:
        q.put((bool(np.array_equal(got, O.hyrax_commit(Z, Ls, blinds, og, Rs + 1, threads=2))), Ls))
    grp.barrier()
    grp.close()


def _run_commit(world, Ls, Rs, use_gpu):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_commit_worker, args=(r, world, port, q, Ls, Rs, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    ok, _ = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


@pytest.mark.timeout(300)
def test_row_sharded_commitment_gloo_cpu():
    _run_commit(2, 8, 16, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_row_sharded_commitment_two_processes_one_gpu():
    _run_commit(2, 64, 64, use_gpu=True)


def _run(world, n, use_gpu):
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q, n, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    ok, rounds = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and rounds == n.bit_length() - 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (2, 2), (4, 4)])
def test_sharded_sumcheck_gloo_cpu(world, n):
    _run(world, n, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n", [(2, 4096), (4, 1 << 14)])
def test_sharded_sumcheck_two_processes_one_gpu(world, n):
    _run(world, n, use_gpu=True)
