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
