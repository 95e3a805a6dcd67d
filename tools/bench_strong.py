"""bench.py --scaling strong [--rehearse W]: ONE trace over all ranks; also the `strong` sub-record of the
default N > 1 line (_strong_core).  Split out of bench.py in round 5; bench.py imports it on demand."""
import json
import os
import sys
import threading
import time

from bench_common import ROOT, SEED_C, SEED_P, _build_resident, _golden_digests, _prove_res, _strong_work, host_cpu_throttle, thread_cpu_seconds  # noqa: F401

T0 = time.perf_counter()
THR0 = host_cpu_throttle()


def main_strong(args):
    """--scaling strong: ONE trace over all ranks.  Instances of at least 2^--coop-log2 constraints are
    proven by ALL ranks together (vpin_comm: row commitments by interleaved rows, sum-check tables and product circuits by
    residue class over a power-of-two world, by circuit index otherwise, see include/vpin_hip.h), mid-size ones
    (--sub-coop-log2) by the first half of the ranks, one after another; the small, latency-bound instances go to the rank
    that is free first and are proven without any exchange (vpin_amd/dist.py plan_trace: a static plan every rank computes).
    value = the trace's constraints x steps / slowest rank's time.
    Ranks = processes (torch.distributed.run; the exchange goes through POSIX shared memory, device buffers through RCCL
    when the backend is nccl)."""
    if args.rehearse:
        return strong_rehearse(args)
    import hashlib
    import torch
    import vpin_amd
    from vpin_amd import Comm
    from vpin_amd.dist import Group, env_rank, plan_trace

    rank, local_rank, world = env_rank()
    ndev = max(1, torch.cuda.device_count())
    dev = local_rank % ndev
    use_nccl = args.backend == "nccl" and world > 1
    if use_nccl:
        torch.cuda.set_device(dev)
    grp = Group(backend=args.backend, device=torch.device("cuda", dev) if use_nccl else None)
    rec = _strong_core(args, grp, rank, world, dev, use_nccl, ndev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(rec))
    grp.close()


def _strong_core(args, grp, rank, world, dev, use_nccl, ndev, steps, warmup):
    """one trace over all ranks of `grp`: the JSON record on rank 0, None on the others (collective)"""
    import hashlib
    import torch
    import vpin_amd
    from vpin_amd import Comm
    from vpin_amd.dist import plan_trace

    trace, work = _strong_work(args)
    total_cons = sum(w[3] for w in work)
    coop_ix, small_ix, _ = plan_trace([w[3] for w in work], world, 0.5 * 2 ** args.coop_log2, 0.5 * 2 ** args.sub_coop_log2)
    coop = [(work[i], g) for i, g in coop_ix]       # in proving order; the group of an entry is ranks [0, g)
    mine = [work[i] for i in small_ix[rank]]

    ctx = vpin_amd.Context(dev)
    comms, rccl_failed, rccl_world = {}, [], {}
    if world > 1:
        name = grp.gather_objects(f"/vpin-{os.getpid()}-{int(time.time() * 1e3) & 0xffffff}")[0]  # rank 0's choice
        for g in sorted({g for _, g in coop}, reverse=True):
            if rank < g:
                comms[g] = Comm.shm(f"{name}-g{g}", rank, g)
                if use_nccl and ndev >= world:
                    try:
                        comms[g].enable_rccl(ctx)  # collective: every rank of the group gets the same verdict
                        rccl_world[g] = g
                    except vpin_amd.VpinError as e:
                        rccl_failed.append(g)     # device vectors are then staged through the shared-memory transport
                        if rank == 0:
                            print(f"bench: RCCL not enabled for the group of {g} ({e}); device buffers staged through the host", file=sys.stderr)
    built = {w[0]: _build_resident(ctx, w) for w in [w for w, g in coop if rank < g] + mine}
    proof_sha = {}

    # Round 5: a rank's own small instances run on a SECOND context (stream + host thread) UNDER the cooperative
    # proofs instead of after them -- two thirds of a cooperative proof are latency-bound rounds and exchanges that leave the
    # rank's GPU idle (dist.py plan_trace still places the small instances as if they came afterwards: an upper bound).
    # VPIN_STRONG_SEQUENTIAL=1 restores the old order (A/B).
    my_coop = [(w, g) for w, g in coop if rank < g]
    overlap = bool(my_coop) and bool(mine) and not os.environ.get("VPIN_STRONG_SEQUENTIAL")
    ctx_small = vpin_amd.Context(dev, priority=-1) if overlap else None
    if overlap:
        ctx.set_shared_device(True)
        ctx_small.set_shared_device(True)

    def prove_small(cx):
        for w in mine:
            proof_sha[w[0]] = hashlib.sha256(_prove_res(cx, *built[w[0]])["proof"]).hexdigest()

    def step():
        th = None
        if overlap:
            th = threading.Thread(target=prove_small, args=(ctx_small,))
            th.start()
        for w, g in my_coop:
            ctx.set_comm(comms[g])
            proof_sha[w[0]] = hashlib.sha256(_prove_res(ctx, *built[w[0]])["proof"]).hexdigest()
        ctx.set_comm(None)
        if th:
            th.join()
        else:
            prove_small(ctx)

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    for cm in comms.values():
        cm.stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t0)
    gold = _golden_digests()
    mine_ok = {k: (gold.get(k, {}).get("snark_sha256") == v) for k, v in proof_sha.items()}  # every rank checks what IT returned
    per_rank = grp.gather_objects({"rank": rank, "bytes_equal_oracle_digest": mine_ok, "rccl_world": rccl_world})
    st = {f"group_of_{g}": cm.stats() for g, cm in comms.items()} if comms else None
    rec = None
    if rank == 0:
        all_ok = {}
        for d in per_rank:
            for k, v in d["bytes_equal_oracle_digest"].items():
                all_ok[k] = all_ok.get(k, True) and v
        rccl = bool(comms) and use_nccl and ndev >= world and not rccl_failed
        rec = {
            "metric": "R1CS constraints/sec, whole Spartan SNARK (sat proof + SPARK evaluation proof; vPIN point-mult + point-add instances)",
            "value": total_cons * steps / elapsed, "unit": "constraints/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u256 (mod q = 2^252+..., mod p = 2^255-19; 32-bit limbs)", "data": "synthetic",
            "config": {"workload": f"ONE vPIN trace '{trace}' over {world} rank(s): {len(work)} SNARKs per step",
                       "constraints_unpadded_per_step": total_cons,
                       "parallelism": f"cooperative proofs {[(w[0], g) for w, g in coop]} (instance, ranks [0, g) together: vpin_comm over "
                                      f"shared memory{', device buffers over RCCL' if rccl else ''}: row commitments by interleaved rows; "
                                      "sum-check tables, product circuits and slices by residue class when the group is a power of two, "
                                      "by circuit index otherwise); the other instances go to the rank that is free first, no exchange",
                       "small_instances_per_rank": [[work[i][0] for i in sh] for sh in small_ix],
                       "small_instances_run": "on a second context of their rank, under the cooperative proofs" if overlap else
                                              "after the cooperative proofs"},
            "bytes_equal_oracle_digest": all_ok,
            "per_rank": per_rank,
            "rccl": {"enabled": rccl, "world_by_group": rccl_world, "failed_groups": rccl_failed,
                     "note": "ncclAllGather carries the device-resident partial vectors of the evaluation proofs; when it cannot be "
                             "enabled (fewer GPUs than ranks, init failure) they are staged through the shared-memory transport"},
            "comm": st}
    for cm in comms.values():
        cm.destroy()
    for g, dec in built.values():
        dec.free()
        g.free()
    if ctx_small:
        ctx_small.close()
    ctx.close()
    return rec


def strong_rehearse(args):
    """--scaling strong --rehearse W: the critical path of a W-GPU run, measured on ONE GPU.  W ranks run as threads of this
    process and prove each cooperative instance together with vpin_comm_set_serialize on: one rank computes at a time, so the
    time a rank spends between two collectives is its own work and nothing else, and the W-GPU time of the proof is the sum over
    the collectives of the slowest rank's section (crit_s) plus the exchanges themselves (collectives x the measured latency of
    an all-gather among W threads).  The trace's time is the static plan of main_strong (plan_trace: which instances are proven
    by all ranks, by half of them, by one) replayed with these modelled and measured times.  A MODEL of the multi-GPU run from
    measured sections -- no multi-GPU hardware was used."""
    import hashlib
    import threading
    import vpin_amd
    from vpin_amd import Comm
    from vpin_amd.dist import plan_trace, replay_trace

    W = args.rehearse
    trace, work = _strong_work(args)
    total_cons = sum(w[3] for w in work)
    cons = [w[3] for w in work]
    if args.coop_all:  # every point-mult instance by all ranks (how each size scales)
        coop_ix = [(i, W) for i, w in enumerate(work) if w[1] == "mult"]
        _, small_ix, _ = plan_trace([0 if w[1] == "mult" else w[3] for w in work], W, float("inf"), float("inf"))
        small_ix = [[i for i in sh if work[i][1] != "mult"] for sh in small_ix]
    else:
        coop_ix, small_ix, _ = plan_trace(cons, W, 0.5 * 2 ** args.coop_log2, 0.5 * 2 ** args.sub_coop_log2)
    group = {work[i][0]: g for i, g in coop_ix}
    gold = _golden_digests()
    ctx0 = vpin_amd.Context(0)

    # exchange latency among W threads (unserialised, 1728-byte pieces = 18 instances x 3 scalars)
    comms = Comm.local(W)
    lat = [0.0] * W

    def pingpong(r):
        lat[r] = comms[r].latency(1728, 5000)

    ts = [threading.Thread(target=pingpong, args=(r,)) for r in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for cm in comms:
        cm.destroy()
    t_ag = max(lat)

    per = {}
    single_ms = {}
    for w in work:
        g, dec = _build_resident(ctx0, w)
        # the single-GPU proof runs on a context of its own, closed afterwards: its pooled temporaries (tens of GB for the
        # largest instance) must not sit in HBM next to the W ranks' own
        ctx1 = vpin_amd.Context(0)
        _prove_res(ctx1, g, dec)  # warm: generator views, pools
        best = 1e9
        for _ in range(max(1, args.steps)):
            t0 = time.perf_counter()
            ref = _prove_res(ctx1, g, dec)
            best = min(best, time.perf_counter() - t0)
        ctx1.close()
        single_ms[w[0]] = best * 1e3
        sha = hashlib.sha256(ref["proof"]).hexdigest()
        rec = {"single_gpu_ms": round(best * 1e3, 3), "bytes_equal_oracle_digest": gold.get(w[0], {}).get("snark_sha256") == sha}
        if w[0] in group:
            Wg = group[w[0]]
            rec["ranks"] = Wg
            ctxs = [ctx0] + [vpin_amd.Context(0) for _ in range(Wg - 1)]
            comms = Comm.local(Wg)
            out, errs, stats, tags = [None] * Wg, [], [None] * Wg, [None] * Wg
            passes = [[] for _ in range(Wg)]

            def body(r):
                try:
                    ctxs[r].set_comm(comms[r])
                    for it in range(1 + args.rehearse_passes):  # first pass warms every rank's pools and generator views
                        comms[r].set_serialize(True)
                        comms[r].stats(reset=True)
                        out[r] = _prove_res(ctxs[r], g, dec)
                        ctxs[r].sync()
                        comms[r].allgather(b"")  # closes the section after the proof's last collective
                        st_r, tg_r = comms[r].stats(), comms[r].tag_stats()
                        comms[r].set_serialize(False)
                        # keep the quietest pass (allocation stalls and host scheduling only ever add time); every rank sees
                        # the same crit_s, so every rank keeps the same pass
                        if it >= 1:
                            passes[r].append((st_r, tg_r))
                            if stats[r] is None or st_r["crit_s"] < stats[r]["crit_s"]:
                                stats[r], tags[r] = st_r, tg_r
                    ctxs[r].set_comm(None)
                except BaseException as e:  # noqa: BLE001
                    errs.append((r, repr(e)))

            ts = [threading.Thread(target=body, args=(r,)) for r in range(Wg)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            for cm in comms:
                cm.destroy()
            for cx in ctxs[1:]:
                cx.close()
            if errs:
                rec["rehearsal_error"] = errs
            else:
                st = stats[0]
                rec.update({
                    "all_ranks_bytes_equal_single_gpu": all(o["proof"] == ref["proof"] for o in out),
                    "collectives": st["collectives"],
                    "crit_ms": round(st["crit_s"] * 1e3, 3),
                    "exchange_ms": round(st["collectives"] * t_ag * 1e3, 3),
                    "model_ms": round((st["crit_s"] + st["collectives"] * t_ag) * 1e3, 3),
                    "busy_ms_per_rank": [round(s["busy_s"] * 1e3, 3) for s in stats],
                    "crit_ms_by_step": {k: round(v["crit_s"] * 1e3, 3) for k, v in sorted(tags[0].items(), key=lambda kv: -kv[1]["crit_s"])},
                    "busy_ms_by_step_per_rank": {k: [round(tags[r].get(k, {"busy_s": 0.0})["busy_s"] * 1e3, 3) for r in range(Wg)]
                                                 for k in sorted(tags[0], key=lambda kk: -tags[0][kk]["crit_s"])[:10]},
                })
                # Per step of the protocol (tag): the library's crit_s is sum over the collectives of the slowest rank's section.
                # On one GPU shared by W ranks two artefacts inflate it: a stall in one rank's section in one pass (allocation
                # when the W ranks' temporaries nearly fill the 288 GB; host scheduling) -- so every step takes its quietest
                # pass -- and, for REPLICATED steps (identical work on every rank), a rank that is slow in every pass for the
                # same reason -- so those take the fastest rank's time.
                replicated = {"sat_replicated", "sat_phase1_rest", "sat_phase2_rest", "derefs_gather", "network_alloc",
                              "hash_eq_tables", "hash_bullet"}
                tagq = {}
                for k in tags[0]:
                    v = min(tg[k]["crit_s"] for _st, tg in passes[0] if k in tg)
                    ncoll = tags[0][k]["collectives"]
                    quiet_r = [min(tg[k]["busy_s"] for _st, tg in passes[r] if k in tg) for r in range(Wg)]
                    if k in replicated:
                        v = min(v, sorted(quiet_r)[len(quiet_r) // 2])  # the MEDIAN rank (the fastest rank is a floor, not an estimate)
                    else:
                        # a sharded step: the slowest rank, each rank at its quietest pass (for the round steps -- hundreds of
                        # collectives with the same work on every owner -- this drops only the per-round jitter)
                        v = min(v, max(quiet_r))
                    tagq[k] = v
                tagged_best = sum(v["crit_s"] for v in tags[0].values())
                untagged = max(0.0, st["crit_s"] - tagged_best)
                quiet = sum(tagq.values()) + untagged
                rec["crit_ms_quietest_pass_per_step"] = round(quiet * 1e3, 3)
                rec["model_ms_quietest_pass_per_step"] = round((quiet + st["collectives"] * t_ag) * 1e3, 3)
                rec["crit_ms_by_step_quietest"] = {k: round(v * 1e3, 3) for k, v in sorted(tagq.items(), key=lambda kv: -kv[1])}
                # every pass of the three longest steps, per rank (how stable the rehearsal is)
                rec["busy_ms_per_rank_every_pass"] = {
                    k: [[round(passes[r][i][1].get(k, {"busy_s": 0.0})["busy_s"] * 1e3, 3) for r in range(Wg)] for i in range(len(passes[0]))]
                    for k in sorted(tagq, key=lambda kk: -tagq[kk])[:3]}
                rec["fraction_of_single_gpu"] = round(rec["model_ms"] / rec["single_gpu_ms"], 4)
                rec["fraction_of_single_gpu_quietest"] = round(rec["model_ms_quietest_pass_per_step"] / rec["single_gpu_ms"], 4)
        per[w[0]] = rec
        dec.free()
        g.free()
    ok = all("model_ms_quietest_pass_per_step" in per[work[i][0]] for i, _ in coop_ix)
    single = [single_ms[w[0]] for w in work]
    # HEADLINE = the unfiltered model (every step at the slowest rank of the measured pass, as a real W-GPU run pays it); the
    # quietest-pass figure is a LOWER BOUND beside it
    loads = replay_trace(coop_ix, small_ix, W, {(i, g): per[work[i][0]]["model_ms"] for i, g in coop_ix}, single) if ok else None
    loads_q = replay_trace(coop_ix, small_ix, W, {(i, g): per[work[i][0]]["model_ms_quietest_pass_per_step"] for i, g in coop_ix}, single) if ok else None
    loads_ov = replay_trace(coop_ix, small_ix, W, {(i, g): per[work[i][0]]["model_ms"] for i, g in coop_ix}, single, overlap=True) if ok else None
    model_ms = max(loads) if ok else None
    model_q = max(loads_q) if ok else None
    serial_ms = sum(single_ms.values())
    step_ms = args.n1_step_ms  # the measured four-lane N = 1 step of the same trace (bench.py default), when given
    print(json.dumps({
        "metric": "critical-path MODEL of one vPIN trace proven by W GPUs (sections measured on one GPU, ranks serialised)",
        "unmeasured_on_multi_gpu_hardware": True, "world": W, "trace": trace, "constraints_unpadded_per_step": total_cons,
        "single_gpu_serial_ms": round(serial_ms, 3), "model_ms": None if model_ms is None else round(model_ms, 3),
        "model_ms_lower_bound_quietest_pass": None if model_q is None else round(model_q, 3),
        "model_ms_small_instances_hidden_under_the_cooperative_proofs": None if loads_ov is None else round(max(loads_ov), 3),
        "model_note": "model_ms places a rank's small instances AFTER its cooperative proofs (an upper bound since round 5: the real "
                      "schedule runs them on a second stream under those proofs); ..._hidden_... assumes they overlap completely (a lower bound)",
        "model_speedup_vs_single_gpu_serial": None if model_ms is None else round(serial_ms / model_ms, 3),
        "model_speedup_vs_n1_four_lane_step": None if (model_ms is None or not step_ms) else round(step_ms / model_ms, 3),
        "n1_four_lane_step_ms": step_ms,
        "speedup_note": "quote the speed-up against the four-lane N = 1 step (what one GPU delivers on the trace), not against the serial sum "
                        "of the instances; model_ms is an estimate from one measured pass, model_ms_lower_bound_quietest_pass an optimistic bound",
        "model_constraints_per_s": None if model_ms is None else total_cons / model_ms * 1e3,
        "allgather_latency_us_among_threads": round(t_ag * 1e6, 2),
        "cooperative": [[work[i][0], g] for i, g in coop_ix],
        "small_instances_per_rank": [[work[i][0] for i in sh] for sh in small_ix],
        "finish_ms_per_rank": None if loads is None else [round(x, 3) for x in loads], "instances": per,
        "plan": dict(zip(("owner_ops", "owner_dotp", "owner_mem"), vpin_amd.dist_plan(W))),
        "threads_cpu_s": thread_cpu_seconds(24), "wall_s": round(time.perf_counter() - T0, 2),
        "host_cpu": dict(host_cpu_throttle(), nr_throttled_in_this_run=host_cpu_throttle().get("nr_throttled", 0) - THR0.get("nr_throttled", 0), env={k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_INTERRUPT", "VPIN_HOST_THREADS")}),
    }))
    ctx0.close()
