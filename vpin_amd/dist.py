"""Multi-GPU plumbing: one process per GPU, instances sharded across ranks, no data-path collective.

vPIN's proofs are independent per gadget instance (2 per trace, 12 for LeNet: SURVEY.md F1), so the
N-GPU path is a work partition plus a barrier and a max-reduce of the wall time.  torch.distributed
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests) is used for exactly those two things.
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


# ---- one large sum-check split across ranks ------------------------------------------
# The folds pair entry i with i + len/2, so a STRIDED partition (rank g owns the indices i = g mod P) keeps both
# members of every pair on one rank for as long as len/2 >= P: tables never move.  In local indexing the shard is
# again a table folded top variable first, so the single-GPU round kernels run unchanged on it; per round every rank
# contributes its partial (e0, e2, e3) -- 96 bytes -- and the sums are taken mod q.  Field addition is not a
# collective reduce op, so the exchange is an all-gather of the partials plus a local modular sum (the transcript
# needs them on the host anyway).  When the shards are down to one entry (len = P) the remaining log2 P rounds run
# on the gathered P entries per table, on every rank alike.

Q = 2**252 + 27742317777372353535851937790883648493
_R = 1 << 256
_RINV = pow(_R, -1, Q)


def _limbs_to_int(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def _int_to_limbs(v):
    import numpy as np
    return np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def strided_shard(table, rank, world):
    """rank's shard of a table of Montgomery scalars ((n,4) uint64): the entries with index = rank mod world"""
    return table[rank::world].copy()


def _gather_sum(grp, partial):
    """partial: (k,4) uint64 Montgomery scalars of this rank -> their sums mod q over all ranks (Montgomery form is
    linear, so the images add)"""
    import numpy as np
    parts = grp.gather_objects(np.ascontiguousarray(partial, dtype=np.uint64).tobytes())
    out = np.zeros_like(np.asarray(partial, dtype=np.uint64))
    for k in range(out.shape[0]):
        s = 0
        for p in parts:
            s += _limbs_to_int(np.frombuffer(p, dtype=np.uint64).reshape(-1, 4)[k])
        out[k] = _int_to_limbs(s % Q)
    return out


def _host_cubic_round(tabs):
    """(e0, e2, e3) of sum_i A(B C - D) on host integers; tabs: 4 lists of canonical ints of equal even length
    (sumcheck.rs:624-652, comb r1csproof.rs:104-108)"""
    half = len(tabs[0]) // 2
    e = [0, 0, 0]
    for i in range(half):
        lo = [t[i] for t in tabs]
        d = [(t[half + i] - t[i]) % Q for t in tabs]
        for k, x in enumerate((0, 2, 3)):
            a, b, c, dd = [(lo[j] + x * d[j]) % Q for j in range(4)]
            e[k] = (e[k] + a * (b * c - dd)) % Q
    return e


def sharded_cubic_sumcheck(grp, ops, shards, challenges):
    """Cubic sum-check rounds over four tables split across grp's ranks by `strided_shard`.

    ops: the local round engine on this rank's shard handles -- `length(tab)`, `round(tabs) -> (3,4)`,
         `bind_round(tabs, r) -> (3,4)` (fold with r, then evaluate), `bind(tabs, r)`, `read(tab) -> (n,4)` -- all in
         Montgomery limbs; on the GPU these are
         Context.sc_cubic_round / sc_cubic_bind_round / Table.read, i.e. the same kernels a single GPU runs.
    shards: the four local table handles (tau, Az, Bz, Cz), local length a power of two.
    challenges: one Montgomery scalar per round ((rounds,4) uint64); in a proof they come from the transcript on
         rank 0 and are broadcast, 32 bytes per round.
    Returns (list of per-round (3,4) evaluations -- the global e0, e2, e3 --, final (4,4) values of the four tables
    at the challenge point)."""
    import numpy as np
    world = grp.world
    local_len = ops.length(shards[0])
    total = local_len * world
    rounds = total.bit_length() - 1
    assert len(challenges) == rounds
    evals = []
    r_prev = None
    j = 0
    while local_len >= 2:  # both members of every pair are local
        if r_prev is None:
            part = ops.round(shards)
        elif local_len >= 4:
            part = ops.bind_round(shards, r_prev)  # fused fold + evaluation of the next round
            local_len //= 2
        else:
            ops.bind(shards, r_prev)  # the last local fold: the shards are single entries afterwards
            local_len //= 2
            break
        evals.append(_gather_sum(grp, part))
        r_prev = challenges[j]
        j += 1
    # finish on the gathered world-sized tables (rank g holds entry g of each)
    vals = [ops.read(t)[:1] for t in shards]
    gathered = grp.gather_objects(np.concatenate(vals).tobytes())
    tabs = [[0] * world for _ in range(4)]
    for g, blob in enumerate(gathered):
        a = np.frombuffer(blob, dtype=np.uint64).reshape(4, 4)
        for t in range(4):
            tabs[t][g] = _limbs_to_int(a[t]) * _RINV % Q  # index = g mod world: rank g holds entry g
    while len(tabs[0]) >= 2:
        e = _host_cubic_round(tabs)
        evals.append(np.stack([_int_to_limbs(x * _R % Q) for x in e]))
        r = _limbs_to_int(challenges[j]) * _RINV % Q
        j += 1
        half = len(tabs[0]) // 2
        tabs = [[(t[i] + r * (t[half + i] - t[i])) % Q for i in range(half)] for t in tabs]
    final = np.stack([_int_to_limbs(t[0] * _R % Q) for t in tabs])
    return evals, final


def sharded_hyrax_commit(grp, commit_rows, z_rows_local, blinds_local):
    """DensePolynomial::commit_inner split across ranks: the L row commitments are
    independent MSMs over the same generators, so rank g commits the contiguous rows [g L/P, (g+1) L/P) it holds and
    the 32-byte results are all-gathered -- no reduction, no point ever crosses a link.

    commit_rows(z_rows_local, blinds_local) -> (L/P, 32) uint8 compressed points (on the GPU: Context.hyrax_commit on
    the uploaded row block); returns the (L, 32) commitment in row order on every rank."""
    import numpy as np
    mine = np.ascontiguousarray(commit_rows(z_rows_local, blinds_local), dtype=np.uint8)
    parts = grp.gather_objects(mine.tobytes())
    return np.concatenate([np.frombuffer(p, dtype=np.uint8).reshape(-1, 32) for p in parts])
