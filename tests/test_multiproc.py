"""N>1 control path on CPU: two gloo processes shard a list of gadget instances, each proves its
shard (with the CPU oracle -- test-side only), and the job-level quantities bench.py reports
(barrier, max-over-ranks time, whole-job constraint count) come out right."""
import hashlib
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gadgets_model as GM
    import oracle_lib as O
    from vpin_amd.dist import Group, plan_shards

    grp = Group(backend="gloo")
    # a LeNet-like mix of instance sizes (point-add instances of different op counts)
    ops = [2, 9, 3, 5, 1]
    costs = [10 * n for n in ops]
    shards = plan_shards(costs, world)
    mine = shards[rank]
    grp.barrier()
    t0 = time.perf_counter()
    digests = {}
    for i in mine:
        inst = GM.instance_new(GM.build_point_add(GM.synthetic_add_ops(100 + i, ops[i])))
        res = O.sat_prove(inst, bytes(range(64)), bytes(64), threads=1)
        assert O.sat_verify(inst, res) == 1
        digests[i] = hashlib.sha256(res["proof"]).hexdigest()
    if rank == 1:
        time.sleep(0.2)  # make the ranks uneven: the job time must be the slowest rank's
    elapsed = time.perf_counter() - t0
    grp.barrier()
    job_time = grp.max_over_ranks(elapsed)
    total = grp.sum_over_ranks(sum(costs[i] for i in mine))
    allg = grp.gather_objects((rank, mine, digests, elapsed))
    if rank == 0:
        q.put(dict(job_time=job_time, total=total, gathered=allg, shards=shards, costs=costs))
    grp.close()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted(i for _, mine, _, _ in out["gathered"] for i in mine)
    assert covered == list(range(len(out["costs"])))  # every instance proven exactly once
    assert out["total"] == sum(out["costs"])
    times = [e for _, _, _, e in out["gathered"]]
    assert abs(out["job_time"] - max(times)) < 1e-6 and out["job_time"] >= 0.2
    loads = [sum(out["costs"][i] for i in s) for s in out["shards"]]
    assert max(loads) - min(loads) <= max(out["costs"])  # LPT balance bound
    # proofs are a function of (instance, seeds) only, not of the rank that produced them
    all_d = {}
    for _, _, d, _ in out["gathered"]:
        all_d.update(d)
    assert len(all_d) == len(out["costs"]) and len(set(all_d.values())) == len(all_d)


def test_plan_shards_properties():
    sys.path.insert(0, ROOT)
    from vpin_amd.dist import plan_shards
    costs = [3464 * n for n in (300, 800, 6000, 240, 168)] + [10 * n for n in (288, 7056, 768, 2400, 5760, 406, 186)]
    for world in (1, 2, 4, 8):
        sh = plan_shards(costs, world)
        assert sorted(i for s in sh for i in s) == list(range(len(costs)))
        assert sh == plan_shards(costs, world)  # deterministic
    # LeNet: L5-mult dominates, so instance sharding alone cannot scale past ~1.25x
    sh8 = plan_shards(costs, 8)
    assert max(sum(costs[i] for i in s) for s in sh8) == 3464 * 6000


def test_plan_trace_groups_and_replay():
    """the static schedule of bench.py --scaling strong: the largest instance by all ranks, the mid-size one by half of them,
    every other instance exactly once on one rank; the same on every rank; replay with given times"""
    sys.path.insert(0, ROOT)
    from vpin_amd.dist import plan_trace, replay_trace
    cons = [3464 * n for n in (6000, 800, 300, 240, 168)] + [10 * n for n in (7056, 5760, 2400, 768, 406, 288, 186)]
    coop_min, sub_min = 0.5 * 2 ** 24, 0.5 * 2 ** 22
    for world in (1, 2, 3, 4, 8):
        coop, small, loads = plan_trace(cons, world, coop_min, sub_min)
        assert (coop, small, loads) == plan_trace(cons, world, coop_min, sub_min)
        seen = [i for i, _ in coop] + [i for sh in small for i in sh]
        assert sorted(seen) == list(range(len(cons)))
        assert len(small) == world and len(loads) == world
        for i, g in coop:
            assert 2 <= g <= world
        if world == 1:
            assert coop == []
        else:
            assert coop[0] == (0, world)  # L5-mult by everybody, first
        if world >= 4:
            assert (1, world // 2) in coop  # L3-mult by the first half of the ranks
            # .. during which the other half starts on the small instances
            assert all(len(small[r]) >= 1 for r in range(world // 2, world))
        elif world > 1:
            assert len(coop) == 1
    coop, small, _ = plan_trace(cons, 8, coop_min, sub_min)
    single = [1.0 + c / 1e5 for c in cons]
    times = {(i, g): single[i] / g + 0.5 for i, g in coop}
    loads = replay_trace(coop, small, 8, times, single)
    t_all = times[(0, 8)]
    assert all(x >= t_all for x in loads)
    assert abs(loads[0] - (t_all + times[(1, 4)] + sum(single[i] for i in small[0]))) < 1e-9
    assert abs(loads[7] - (t_all + sum(single[i] for i in small[7]))) < 1e-9
