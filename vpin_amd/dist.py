"""Multi-GPU plumbing: one process per GPU.

vPIN's proofs are independent per gadget instance (2 per trace, 12 for LeNet: SURVEY.md F1), so the default
N-GPU path is a work partition plus a barrier and a max-reduce of the wall time -- no data-path collective.
The one exchange step that exists is the row split of a single large commitment (SplitCommit below): a broadcast
of two challenge vectors and an all-gather of 32-byte rows.  torch.distributed carries both (backend "nccl" =
RCCL on a GPU node, "gloo" in the CPU tests); no field or group arithmetic lives in this module.
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


# ---- one large commitment split across ranks (BASELINE configs[4] "split MSM") ---------------
# The L row commitments of a Hyrax commitment are independent MSMs over shared generators, so the single largest MSM of a
# SNARK -- the derefs commitment of R1CSEvalProof::prove, 6N full-width scalars -- splits by rows: rank g commits the
# contiguous block [g L/P, (g+1) L/P) and the 32-byte compressed rows are all-gathered; no point crosses a link and
# nothing is reduced (group addition is not a collective reduce op anyway).  The proving rank drives this through the
# two hooks of vpin_ctx_set_split_hooks (include/vpin_hip.h): `begin` broadcasts (rx, ry) the moment the sat proof has
# produced them, so the helpers rebuild the derefs polynomial from their own copy of the computation decommitment while
# the prover builds its own; `commit` commits the prover's block and gathers the others.  Collectives: torch.distributed
# on the group's backend -- "nccl" (RCCL over xGMI) on a GPU node, "gloo" in the CPU tests; payloads are 32 bytes per
# row and two challenge vectors, so the links carry kilobytes.  All arithmetic stays in the library.

_HDR = 16  # int64 words of the broadcast header: [nx, ny, L, R, stop]


class SplitCommit:
    """Row-split of the derefs commitment over the ranks of `grp`.

    engine: the per-rank calls into the library (on a GPU box: SplitEngine below):
      commit_rows(Z_handle, L, row0, nrows) -> (nrows, 32) uint8      prover's block from its own polynomial
      derefs_commit_rows(rx, ry, row0, nrows) -> (nrows, 32) uint8    helper's block, derefs rebuilt from (rx, ry)
    """

    def __init__(self, grp, engine, owner=0):
        self.grp, self.engine, self.owner = grp, engine, owner

    # ---- tensors on the backend's device ----
    def _tensor(self, arr):
        import torch
        t = torch.from_numpy(arr)
        return t.to(self.grp.device) if self.grp.device is not None else t

    def _bcast(self, arr):
        """arr: numpy array (filled on the owner, any content elsewhere) -> the owner's content on every rank"""
        t = self._tensor(arr)
        self.grp.dist.broadcast(t, src=self.owner)
        return t.cpu().numpy()

    def _allgather_rows(self, mine):
        import numpy as np
        import torch
        t = self._tensor(np.ascontiguousarray(mine))
        outs = [torch.empty_like(t) for _ in range(self.grp.world)]
        self.grp.dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy() for o in outs])

    @staticmethod
    def block(L, rank, world):
        per = L // world
        return rank * per, per

    # ---- prover side (called from inside vpin_snark_prove_* through the hooks) ----
    def begin(self, rx, ry):
        import numpy as np
        rx = np.ascontiguousarray(rx, dtype=np.uint64).reshape(-1, 4)
        ry = np.ascontiguousarray(ry, dtype=np.uint64).reshape(-1, 4)
        hdr = np.zeros(_HDR + 4 * 128, dtype=np.int64)
        hdr[0], hdr[1] = rx.shape[0], ry.shape[0]
        hdr[_HDR:_HDR + rx.size] = rx.reshape(-1).view(np.int64)
        hdr[_HDR + 256:_HDR + 256 + ry.size] = ry.reshape(-1).view(np.int64)
        if self.grp.dist is not None:
            self._bcast(hdr)

    def commit(self, z_handle, L, R):
        import numpy as np
        world = self.grp.world
        if L % world:
            raise ValueError("rows do not divide over the ranks")
        row0, nrows = self.block(L, self.grp.rank, world)
        mine = self.engine.commit_rows(z_handle, L, row0, nrows)
        if self.grp.dist is None:
            return mine
        return self._allgather_rows(np.ascontiguousarray(mine, dtype=np.uint8))

    def stop(self):
        """prover: tell the helpers no further request follows"""
        import numpy as np
        if self.grp.dist is not None:
            hdr = np.zeros(_HDR + 4 * 128, dtype=np.int64)
            hdr[4] = 1
            self._bcast(hdr)

    # ---- helper side ----
    def serve_one(self, L):
        """wait for the prover's (rx, ry), commit this rank's block of the derefs polynomial, join the all-gather.
        Returns False when the prover said stop."""
        import numpy as np
        hdr = self._bcast(np.zeros(_HDR + 4 * 128, dtype=np.int64))
        if hdr[4]:
            return False
        nx, ny = int(hdr[0]), int(hdr[1])
        rx = hdr[_HDR:_HDR + 4 * nx].view(np.uint64).reshape(nx, 4).copy()
        ry = hdr[_HDR + 256:_HDR + 256 + 4 * ny].view(np.uint64).reshape(ny, 4).copy()
        row0, nrows = self.block(L, self.grp.rank, self.grp.world)
        mine = self.engine.derefs_commit_rows(rx, ry, row0, nrows)
        self._allgather_rows(np.ascontiguousarray(mine, dtype=np.uint8))
        return True


class SplitEngine:
    """The library calls behind SplitCommit on a GPU rank (vpin_hyrax_commit_rows / vpin_spark_derefs_commit_rows)."""

    def __init__(self, ctx, decomm=None, derefs_ell=None):
        self.ctx, self.decomm, self.derefs_ell = ctx, decomm, derefs_ell

    def commit_rows(self, z_handle, L, row0, nrows):
        return self.ctx.spark_commit_rows(z_handle, self.derefs_ell, L, row0, nrows)

    def derefs_commit_rows(self, rx, ry, row0, nrows):
        return self.ctx.spark_derefs_commit_rows(self.decomm, rx, ry, row0, nrows)
