"""The exchange layer of one proof over several GPUs (vpin_comm, include/vpin_hip.h) on the CPU: the three host transports
(threads of a process, processes through POSIX shared memory, the caller's own all-gather = gloo here), chunking of large
messages, the compute token of the serialized rehearsal, bounded waits when a peer never shows up, and the circuit plan."""
import multiprocessing as mp
import os
import threading
import time

import numpy as np
import pytest

import vpin_amd
from vpin_amd import Comm


def _payload(rank, it, n):
    return bytes(((rank * 131 + it * 17 + k) & 0xFF) for k in range(n))


def _run_rounds(cm, world, sizes):
    for it, n in enumerate(sizes):
        got = cm.allgather(_payload(cm.rank, it, n))
        assert len(got) == n * world
        for r in range(world):
            assert got[r * n:(r + 1) * n] == _payload(r, it, n), (cm.rank, it, r)


SIZES = [96, 0, 1, 1728, 32 * 4096, 5000, 96, 96, 3 * 4096 + 5]


def test_local_threads_allgather_and_chunking():
    world = 4
    comms = Comm.local(world, slot_bytes=4096)  # messages above 4 KiB travel in pieces
    errs = []

    def body(cm):
        try:
            _run_rounds(cm, world, SIZES * 3)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=body, args=(cm,)) for cm in comms]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errs, errs
    st = comms[0].stats()
    assert st["collectives"] == len(SIZES) * 3 and st["bytes"] == 3 * sum(SIZES)
    for cm in comms:
        cm.destroy()


def _shm_worker(name, rank, world, q):
    try:
        cm = Comm.shm(name, rank, world, slot_bytes=8192)
        _run_rounds(cm, world, SIZES * 2)
        st = cm.stats()
        cm.destroy()
        q.put((rank, "ok", st["collectives"]))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e), 0))


def test_shm_processes_allgather():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/vpin-test-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    ps = [ctx.Process(target=_shm_worker, args=(name, r, world, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(30) for p in ps]
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] == "ok" and r[2] == 2 * len(SIZES) for r in res), res
    assert not os.path.exists("/dev/shm" + name)  # unlinked as soon as everyone was attached


def test_serialized_sections_never_overlap_and_critical_path():
    """vpin_comm_set_serialize: one rank computes at a time; crit_s sums the slowest rank's section per collective"""
    world = 3
    comms = Comm.local(world)
    spans, errs = [], []
    lock = threading.Lock()
    work = {0: 0.03, 1: 0.01, 2: 0.02}

    def body(cm):
        try:
            cm.set_serialize(True)
            cm.stats(reset=True)
            for it in range(4):
                t0 = time.perf_counter()
                time.sleep(work[cm.rank])
                t1 = time.perf_counter()
                with lock:
                    spans.append((t0, t1, cm.rank))
                cm.allgather(b"x" * 8)
            cm.set_serialize(False)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=body, args=(cm,)) for cm in comms]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert not errs, errs
    spans.sort()
    for (a0, a1, _), (b0, b1, _) in zip(spans, spans[1:]):
        assert b0 >= a1 - 1e-4, "two ranks computed at the same time"
    st = comms[1].stats()
    assert st["collectives"] == 4
    assert 4 * 0.03 * 0.9 < st["crit_s"] < 4 * 0.03 * 2.5   # max over ranks per section, not the sum (0.06 each)
    assert st["busy_s"] < st["crit_s"]                        # rank 1 is never the slowest
    for cm in comms:
        cm.destroy()


def _late_worker(name, q):
    os.environ["VPIN_COMM_TIMEOUT_S"] = "1.5"
    try:
        Comm.shm(name, 1, 2)
        q.put("attached")
    except vpin_amd.VpinError as e:
        q.put(e.code)


def test_missing_peer_times_out_instead_of_hanging():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/vpin-test-late-{os.getpid()}"
    p = ctx.Process(target=_late_worker, args=(name, q))  # rank 1 of 2; rank 0 never comes
    t0 = time.time()
    p.start()
    assert q.get(timeout=60) == -7  # VPIN_ECOMM
    p.join(30)
    assert time.time() - t0 < 40


def test_peer_that_stops_makes_the_collective_fail():
    os.environ["VPIN_COMM_TIMEOUT_S"] = "1.0"
    try:
        comms = Comm.local(2)
    finally:
        del os.environ["VPIN_COMM_TIMEOUT_S"]
    with pytest.raises(vpin_amd.VpinError) as ei:
        comms[0].allgather(b"abc")  # rank 1 never calls
    assert ei.value.code == -7
    with pytest.raises(vpin_amd.VpinError):  # the abort word is sticky: the other rank fails at once
        comms[1].allgather(b"abc")
    for cm in comms:
        cm.destroy()


def test_abort_fails_a_waiting_peer_at_once():
    comms = Comm.local(3)  # default timeout: 120 s
    got = {}

    def waiter(r):
        t0 = time.time()
        try:
            comms[r].allgather(b"x" * 96)
            got[r] = ("returned", time.time() - t0)
        except vpin_amd.VpinError as e:
            got[r] = (e.code, time.time() - t0)

    ts = [threading.Thread(target=waiter, args=(r,)) for r in (0, 1)]
    [t.start() for t in ts]
    time.sleep(0.3)            # ranks 0 and 1 are inside the collective, rank 2 "fails outside the library"
    comms[2].abort()
    [t.join(30) for t in ts]
    assert got[0][0] == -7 and got[1][0] == -7, got
    assert got[0][1] < 10 and got[1][1] < 10, got
    with pytest.raises(vpin_amd.VpinError):
        comms[2].allgather(b"x" * 96)  # sticky for everyone
    for cm in comms:
        cm.destroy()


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allgather(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return b"".join(bytes(o.numpy().tobytes()) for o in outs)

    try:
        cm = Comm.callbacks(rank, world, allgather)
        _run_rounds(cm, world, [96, 1, 4096, 96])
        cm.destroy()
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    dist.destroy_process_group()


def test_callbacks_transport_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    ps = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(60) for p in ps]
    assert all(r[1] == "ok" for r in res), res


def test_plan_is_balanced_and_complete():
    for world in range(1, 13):
        ops, dotp, mem = vpin_amd.dist_plan(world)
        assert len(ops) == 12 and len(dotp) == 6 and len(mem) == 4
        assert all(0 <= r < world for r in ops + dotp + mem)
        load = [4 * ops.count(r) + 3 * dotp.count(r) for r in range(world)]
        assert min(ops.count(r) for r in range(world)) >= 1           # every rank runs rounds of the ops forest
        assert max(load) - min(load) <= 4                              # one circuit of imbalance at most
        assert max(mem.count(r) for r in range(world)) == -(-4 // world)
    ops, dotp, mem = vpin_amd.dist_plan(8)
    assert max(4 * ops.count(r) + 3 * dotp.count(r) for r in range(8)) == 10  # of 66: the ops phase at 1/6.6 per rank


def test_ranks_that_disagree_on_a_collective_fail_promptly():
    """every rank publishes what it believes the collective IS (size, call site) next to its sequence number;
    a rank that took another protocol branch is refused at the first collective the ranks disagree on, on both sides,
    instead of its slot being read at the wrong size."""
    comms = Comm.local(2)
    out = {}

    def body(cm, n):
        try:
            cm.allgather(b"s" * 8)          # in step
            cm.allgather(b"x" * n)          # rank 0: 96 bytes, rank 1: 64 bytes
            out[cm.rank] = "no error"
        except vpin_amd.VpinError as e:
            out[cm.rank] = e.code
            msgs.append(str(e))

    msgs = []
    t0 = time.time()
    ts = [threading.Thread(target=body, args=(cm, n)) for cm, n in zip(comms, (96, 64))]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert out == {0: -7, 1: -7}, out     # VPIN_ECOMM on both ranks
    assert time.time() - t0 < 10          # not the 120 s timeout
    assert any("disagree" in m for m in msgs), msgs   # vpin_last_error() is per thread: the rank that saw the mismatch
    for cm in comms:
        cm.destroy()


def _stale_then_fresh_worker(name, rank, world, delay, q):
    os.environ["VPIN_COMM_TIMEOUT_S"] = "20"
    time.sleep(delay)
    try:
        cm = Comm.shm(name, rank, world, slot_bytes=4096)
        got = cm.allgather(bytes([rank]) * 4)
        cm.destroy()
        q.put((rank, got))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_shm_leftover_of_a_dead_job_is_not_attached_to():
    """a segment left under the same name by a job that died (here: a file of the right size whose header never
    gets initialised, opened by rank 1 BEFORE rank 0 replaces it) must not swallow the late rank: it notices that the name
    moved on, re-opens, and the group forms."""
    name = f"/vpin-test-stale-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    with open("/dev/shm" + name, "wb") as f:
        f.truncate(1 << 20)   # larger than the real segment: passes the size check
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_stale_then_fresh_worker, args=(name, 1, 2, 0.0, q)),   # maps the leftover first
          ctx.Process(target=_stale_then_fresh_worker, args=(name, 0, 2, 1.0, q))]   # replaces it a second later
    [p.start() for p in ps]
    res = dict(q.get(timeout=90) for _ in range(2))
    [p.join(30) for p in ps]
    assert res == {0: bytes([0] * 4 + [1] * 4), 1: bytes([0] * 4 + [1] * 4)}, res
    assert not os.path.exists("/dev/shm" + name)


def _die_as_rank0(name):
    os.environ["VPIN_COMM_TIMEOUT_S"] = "60"
    Comm.shm(name, 0, 2, slot_bytes=4096)  # initialises the segment, then waits for a rank 1 that never comes: killed by the test


def test_shm_segment_of_a_job_that_just_died_is_not_attached_to():
    """the leftover is a FULLY initialised segment (magic, world, a fresh time stamp) whose creator was killed a
    moment ago -- it passes the size / magic / age / still-named checks.  The late rank of the next job must recognise it by
    its dead creator, wait for rank 0 to replace it, and the group must form."""
    name = f"/vpin-test-dead-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}"
    ctx = mp.get_context("spawn")
    dead = ctx.Process(target=_die_as_rank0, args=(name,))
    dead.start()
    t0 = time.time()
    while not os.path.exists("/dev/shm" + name) and time.time() - t0 < 60:
        time.sleep(0.05)
    time.sleep(1.0)   # let it finish seg_init
    dead.kill()
    dead.join(10)
    assert os.path.exists("/dev/shm" + name)   # the dead job's segment is still there, under the name the next job uses
    q = ctx.Queue()
    ps = [ctx.Process(target=_stale_then_fresh_worker, args=(name, 1, 2, 0.0, q)),   # meets the dead job's segment first
          ctx.Process(target=_stale_then_fresh_worker, args=(name, 0, 2, 1.5, q))]   # replaces it later
    [p.start() for p in ps]
    res = dict(q.get(timeout=90) for _ in range(2))
    [p.join(30) for p in ps]
    assert res == {0: bytes([0] * 4 + [1] * 4), 1: bytes([0] * 4 + [1] * 4)}, res
    assert not os.path.exists("/dev/shm" + name)


def test_callback_fabric_hands_a_size_mismatch_back_as_ecomm():
    """the callback transport (kind 2) carries size and call site in front of the payload; a fabric that survives
    ranks sending different sizes must make BOTH ranks fail with VPIN_ECOMM ('disagree'), not mix the slots up."""
    world = 2
    box, bar = [None] * world, threading.Barrier(world)

    def fabric(rank):
        def allgather(send):
            box[rank] = send
            bar.wait(30)
            n = len(send)
            out = b"".join((box[r] + b"\0" * n)[:n] for r in range(world))  # every rank's piece cut / padded to MY size
            bar.wait(30)
            return out
        return allgather

    comms = [Comm.callbacks(r, world, fabric(r)) for r in range(world)]
    out, msgs = {}, []

    def body(cm, n):
        try:
            assert cm.allgather(b"s" * 8) == b"s" * 16          # in step
            cm.allgather(b"x" * n)                               # rank 0: 96 bytes, rank 1: 64 bytes
            out[cm.rank] = "no error"
        except vpin_amd.VpinError as e:
            out[cm.rank] = e.code
            msgs.append(str(e))

    ts = [threading.Thread(target=body, args=(cm, n)) for cm, n in zip(comms, (96, 64))]
    [t.start() for t in ts]
    [t.join(60) for t in ts]
    assert out == {0: -7, 1: -7}, out
    assert any("disagree" in m for m in msgs), msgs
    for cm in comms:
        cm.destroy()
