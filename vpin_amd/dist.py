"""Multi-GPU plumbing of bench.py: one process per GPU.

vPIN's proofs are independent per gadget instance (2 per trace, 12 for LeNet: SURVEY.md F1), so the default
N-GPU path is a work partition plus a barrier and a max-reduce of the wall time over torch.distributed (backend
"nccl" = RCCL on a GPU node, "gloo" in the CPU tests) -- no data-path collective.  ONE proof over several GPUs does not
live here: its exchange layer is native (vpin_comm in include/vpin_hip.h, vpin_amd/csrc/comm.cpp), so that a Rust host
can drive it without Python.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def plan_shards(costs, world):
    """Greedy longest-processing-time partition of instances (cost = constraints) over `world`
    ranks.  Returns a list of index lists, one per rank; deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def est_single_ms(cons):
    """Rough single-GPU time of one whole SNARK, ms (a fit of the LeNet instances on MI355X: a latency floor plus a rate);
    only the ORDER and the rough proportions matter -- the plan below must be the same on every rank without measuring."""
    return 12.0 + cons / 40e3


def est_group_fraction(g):
    """share of the single-GPU time a cooperative proof over g ranks is expected to take (rehearsals: 0.61 / 0.43 / 0.30)"""
    return 0.25 + 0.75 / g


def plan_trace(cons, world, coop_min, sub_min):
    """Static schedule of one trace over `world` ranks (bench.py --scaling strong).  cons: unpadded constraints per instance.
    Instances of at least coop_min constraints are proven by ALL ranks together; instances of at least sub_min by the first
    world/2 ranks (a smaller group wastes less of a mid-size proof's latency-bound rounds) while the others start on the
    small ones; the rest go, longest first, to whichever rank is free first.  Returns (coop, small, loads):
    coop = [(index, group size)] in proving order (a group is always ranks [0, size)), small = index lists per rank,
    loads = the estimated finish time of every rank, ms.  Deterministic: every rank computes the same plan."""
    order = sorted(range(len(cons)), key=lambda i: (-cons[i], i))
    loads = [0.0] * world
    coop, rest = [], []
    for i in order:
        g = 1
        if world > 1 and cons[i] >= coop_min:
            g = world
        elif world >= 4 and cons[i] >= sub_min:
            g = world // 2
        if g > 1:
            end = max(loads[:g]) + est_single_ms(cons[i]) * est_group_fraction(g)
            for r in range(g):
                loads[r] = end
            coop.append((i, g))
        else:
            rest.append(i)
    small = [[] for _ in range(world)]
    for i in rest:
        r = min(range(world), key=lambda k: (loads[k], k))
        small[r].append(i)
        loads[r] += est_single_ms(cons[i])
    return coop, small, loads


def replay_trace(coop, small, world, coop_ms, single_ms, overlap=False):
    """finish time of every rank when the plan runs with the given times: coop_ms[(index, size)], single_ms[index].
    overlap: a rank's small instances run on a second stream UNDER its cooperative proofs (bench_strong since round 5) and are
    taken to hide completely behind them -- a LOWER bound; without it they follow the cooperative proofs -- an upper bound."""
    loads = [0.0] * world
    for i, g in coop:
        end = max(loads[:g]) + coop_ms[(i, g)]
        for r in range(g):
            loads[r] = end
    for r in range(world):
        own = sum(single_ms[i] for i in small[r])
        loads[r] = max(loads[r], own) if overlap else loads[r] + own
    return loads


class Group:
    """Thin wrapper so bench.py and the gloo tests share the same control path."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend=backend or "gloo", **kwargs)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
