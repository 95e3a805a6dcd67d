"""One proof over several GPUs (vpin_comm, include/vpin_hip.h): every rank runs the same
vpin_snark_prove_resident on the same instance and seeds; row commitments, product circuits, dot-product halves, slice
evaluations and polynomial bounds are sharded and their small results all-gathered.  The bytes every rank returns must be
the single-GPU proof's (= the oracle's, tests/golden/config_digests.json).  Two splits are covered: by residue class (a
power-of-two world: every circuit, both sum-checks, the derefs gather and the slices shrink with the world) and by whole circuits
(any world up to 12).  Here the ranks share the one MI355X: as threads
of this process (local transport, world 2..8, also through the serialized rehearsal) and as separate processes through
POSIX shared memory (world 2 and 4; an instance of 2^20 constraints)."""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))


def _build(ctx, label, kind, n):
    from vpin_amd import gadgets as G
    if kind == "mult":
        return ctx.gadget_point_mult_dev(*G.synthetic_mult_inputs(label, n))
    return ctx.gadget_point_add_dev(*G.synthetic_add_inputs(label, n))


def _prove_threads(world, label, kind, n, serialize=False):
    """-> (single-GPU proof, [proof of rank r], comm stats of rank 0)"""
    import vpin_amd
    from vpin_amd import Comm
    ctxs = [vpin_amd.Context(0) for _ in range(world)]
    g = _build(ctxs[0], label, kind, n)
    dec, _comm = g.spark_encode()
    single = ctxs[0].snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
    ctxs[0].pool_trim()   # the single-GPU proof's cached temporaries (64 GB for the 2^25 instance) are not the ranks' to carry
    comms = Comm.local(world)
    out, errs = [None] * world, []

    def body(r):
        try:
            ctxs[r].set_comm(comms[r])
            if serialize:
                comms[r].set_serialize(True)
            comms[r].stats(reset=True)
            out[r] = ctxs[r].snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
            if serialize:
                comms[r].set_serialize(False)
            ctxs[r].set_comm(None)
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    stats = comms[0].stats()
    stats["tags"] = comms[0].tag_stats()
    dec.free()
    g.free()
    for cm in comms:
        cm.destroy()
    for c in ctxs:
        c.close()
    assert not errs, errs
    return single, out, stats


def _same(single, out):
    for r, res in enumerate(out):
        assert res is not None, f"rank {r} returned nothing"
        assert res["proof"] == single["proof"], f"rank {r}: proof differs from the single-GPU proof"
        assert np.array_equal(res["comm_para"], single["comm_para"]) and np.array_equal(res["comm_input"], single["comm_input"])


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_threads_point_add_64(world):
    """N = 2^10: host-proved, tail-only and launched layers; L = 16 rows over up to 8 ranks"""
    single, out, st = _prove_threads(world, "3_32", "add", 64)
    _same(single, out)
    assert st["collectives"] > 50


@pytest.mark.parametrize("world", [2, 5, 8])
def test_threads_point_mult_18(world):
    """conv f=3's point-mult instance (2^16 constraints, N = 2^17): the oracle's bytes (config digest)"""
    single, out, st = _prove_threads(world, "3_32", "mult", None)
    _same(single, out)
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f)["cases"]["3_32-mult"]
    assert hashlib.sha256(out[-1]["proof"]).hexdigest() == want["snark_sha256"]
    # a power-of-two world splits every circuit and both sum-checks by residue class; any other world deals the circuits out whole
    by_residue = "ops_gather_tables" in st["tags"]
    assert by_residue == (world in (2, 8)) and ("sat_gather_tables" in st["tags"]) == (world in (2, 8))


def test_threads_tiny_instance_more_ranks_than_rows():
    """4 point additions: 2^6 constraints, L = 8 rows of the witness, N = 2^6 -- ranks with empty row blocks"""
    single, out, _ = _prove_threads(8, "3_32", "add", 4)
    _same(single, out)


def test_serialized_rehearsal_gives_the_same_bytes_and_a_critical_path():
    single, out, st = _prove_threads(4, "3_32", "mult", None, serialize=True)
    _same(single, out)
    assert 0.0 < st["crit_s"] < 60.0 and st["busy_s"] <= st["crit_s"] + 1e-9


# ---- separate processes sharing the GPU, POSIX shared memory between them ------------------------------------------------

def _proc_worker(name, rank, world, label, kind, n, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import vpin_amd
        from vpin_amd import Comm
        with vpin_amd.Context(0) as ctx:
            g = _build(ctx, label, kind, n)
            dec, comm = g.spark_encode()
            cm = Comm.shm(name, rank, world)
            ctx.set_comm(cm)
            res = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
            st = cm.stats()
            ctx.set_comm(None)
            cm.destroy()
            dec.free()
            g.free()
        q.put((rank, hashlib.sha256(res["proof"]).hexdigest(), hashlib.sha256(comm).hexdigest(), st["collectives"]))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, "error: " + repr(e), "", 0))


@pytest.mark.parametrize("world,label", [(2, "A"), (4, "A")])
def test_processes_share_one_gpu_2pow20_instance(world, label):
    """CNN A's point-mult instance: 178 operations, 2^20 constraints, N = 2^20 -- hot-column commitment, streaming rounds"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/vpin-dist-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    os.environ.setdefault("VPIN_GENS_BUDGET_GB", "8")        # every process builds its own window tables: keep them small
    os.environ.setdefault("VPIN_SPARK_GENS_BUDGET_GB", "8")
    ps = [ctx.Process(target=_proc_worker, args=(name, r, world, label, "mult", None, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in range(world)]
    [p.join(120) for p in ps]
    assert all(not r[1].startswith("error") for r in res), res
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f)["cases"][f"{label}-mult"]
    for r in res:
        assert r[1] == want["snark_sha256"], (r, want["snark_sha256"])
        assert r[2] == want["comm_sha256"]
        assert r[3] > 100


@pytest.mark.parametrize("world", [2, 4])
def test_processes_share_one_gpu_cnn_e_config(world):
    """BASELINE configs[3] (CNN E: 658 point-mults, 2,279,312 constraints, 2^22 padded) proven by 2 and by 4 PROCESSES together
    through POSIX shared memory -- config size inside the driver-run suite; bytes = the oracle's digest"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/vpin-dist-E-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    os.environ.setdefault("VPIN_GENS_BUDGET_GB", "8")
    os.environ.setdefault("VPIN_SPARK_GENS_BUDGET_GB", "8")
    ps = [ctx.Process(target=_proc_worker, args=(name, r, world, "E", "mult", None, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=800) for _ in range(world)]
    [p.join(120) for p in ps]
    assert all(not r[1].startswith("error") for r in res), res
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f)["cases"]["E-mult"]
    for r in res:
        assert r[1] == want["snark_sha256"], (r, want["snark_sha256"])
        assert r[2] == want["comm_sha256"]
        assert r[3] > 100


def test_threads_l5_mult_by_two_ranks():
    """BASELINE configs[4]'s large instance (LeNet L5: 6000 point-mults, 20,784,000 constraints, 2^25 padded) proven by two
    ranks together (threads of this process, one decommitment and one set of window tables between them): every rank returns
    the single-GPU proof = the oracle's digest"""
    single, out, st = _prove_threads(2, "L5", "mult", None)
    _same(single, out)
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f)["cases"]["L5-mult"]
    assert hashlib.sha256(out[-1]["proof"]).hexdigest() == want["snark_sha256"]
    assert "ops_gather_tables" in st["tags"] and st["collectives"] > 100


def test_rccl_world_one_device_allgather():
    """RCCL refuses two ranks on one GPU, so on this box only the world-1 communicator can run: dlopen, ncclGetUniqueId,
    ncclCommInitRank and ncclAllGather on the context's stream are exercised; world > 1 on one GPU takes the staged path"""
    import ctypes as C
    import vpin_amd
    from vpin_amd import Comm
    with vpin_amd.Context(0) as ctx:
        cm = Comm.local(1)[0]
        cm.enable_rccl(ctx)
        src = ctx.upload(np.arange(4 * 64, dtype=np.uint64).reshape(64, 4))
        dst = ctx.alloc(64)
        cm.allgather_dev(ctx, C.c_void_p(src.device_ptr), C.c_void_p(dst.device_ptr), 64 * 32)
        ctx.sync()
        assert np.array_equal(dst.read(), src.read())
        src.free()
        dst.free()
        cm.destroy()


def test_staged_device_allgather_threads():
    import ctypes as C
    import vpin_amd
    from vpin_amd import Comm
    world = 4
    comms = Comm.local(world)
    ctxs = [vpin_amd.Context(0) for _ in range(world)]
    outs, errs = [None] * world, []

    def body(r):
        try:
            src = ctxs[r].upload(np.full((128, 4), r + 1, dtype=np.uint64))
            dst = ctxs[r].alloc(128 * world)
            comms[r].allgather_dev(ctxs[r], C.c_void_p(src.device_ptr), C.c_void_p(dst.device_ptr), 128 * 32)
            outs[r] = dst.read()
            src.free()
            dst.free()
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs
    want = np.concatenate([np.full((128, 4), r + 1, dtype=np.uint64) for r in range(world)])
    for o in outs:
        assert np.array_equal(o, want)
    for cm in comms:
        cm.destroy()
    for c in ctxs:
        c.close()


def test_peer_failure_in_the_middle_of_a_proof_fails_the_others_fast_and_leaves_them_usable():
    """Rank 1 of 2 never joins the proof and takes the group down a few milliseconds into rank 0's proof (vpin_comm_abort:
    what the library does itself when a rank leaves a collective proof with VPIN_ENOMEM / VPIN_EHIP).  Rank 0 is then inside
    a round exchange, with its persistent round kernel resident: it must come back with VPIN_ECOMM within seconds -- not
    after the 120 s timeout --, its resident kernel must drain, and the same context must prove the instance alone afterwards."""
    import vpin_amd
    from vpin_amd import Comm
    ctx = vpin_amd.Context(0)
    g = _build(ctx, "3_32", "mult", None)
    dec, _ = g.spark_encode()
    want = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)["proof"]
    for delay in (0.0, 0.004, 0.012):
        comms = Comm.local(2)
        ctx.set_comm(comms[0])
        killer = threading.Timer(delay, comms[1].abort)
        t0 = time.time()
        killer.start()
        with pytest.raises(vpin_amd.VpinError) as ei:
            ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
        dt = time.time() - t0
        killer.join()
        assert ei.value.code == -7, ei.value
        assert dt < 20.0, dt
        ctx.set_comm(None)
        for cm in comms:
            cm.destroy()
        again = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)["proof"]
        assert again == want
    dec.free()
    g.free()
    ctx.close()


def _dying_worker(name, rank, world, q):
    """rank 0 proves; rank 1 attaches to the group and then dies without a word, a few milliseconds into rank 0's proof"""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import vpin_amd
        from vpin_amd import Comm
        with vpin_amd.Context(0) as ctx:
            g = _build(ctx, "3_32", "mult", None)
            dec, _ = g.spark_encode()
            cm = Comm.shm(name, rank, world)
            if rank == 1:
                time.sleep(0.05)
                os._exit(17)  # no abort, no destroy: a killed process
            ctx.set_comm(cm)
            t0 = time.time()
            try:
                ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
                q.put((rank, "proved", 0.0))
            except vpin_amd.VpinError as e:
                dt = time.time() - t0
                ctx.set_comm(None)
                # the context survives the dead group: the same instance alone
                alone = ctx.snark_prove_resident(g.r1cs, dec, g.vars_para, g.vars_input, g.vars, g.inputs, SEED_C, SEED_P)
                q.put((rank, e.code, dt, hashlib.sha256(alone["proof"]).hexdigest()))
            cm.destroy()
            dec.free()
            g.free()
    except BaseException as e:  # noqa: BLE001
        q.put((rank, "error: " + repr(e), 0.0))


def test_a_killed_process_is_noticed_within_the_timeout():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/vpin-dying-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    os.environ["VPIN_COMM_TIMEOUT_S"] = "3"
    os.environ.setdefault("VPIN_GENS_BUDGET_GB", "8")
    os.environ.setdefault("VPIN_SPARK_GENS_BUDGET_GB", "8")
    try:
        ps = [ctx.Process(target=_dying_worker, args=(name, r, 2, q)) for r in range(2)]
        [p.start() for p in ps]
        res = q.get(timeout=300)
        [p.join(60) for p in ps]
    finally:
        del os.environ["VPIN_COMM_TIMEOUT_S"]
    assert res[0] == 0 and res[1] == -7, res          # VPIN_ECOMM on the survivor
    assert 2.0 < res[2] < 30.0, res                   # after the timeout, not after the default 120 s
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f)["cases"]["3_32-mult"]["snark_sha256"]
    assert res[3] == want
    assert ps[1].exitcode == 17
