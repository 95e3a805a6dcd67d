#!/usr/bin/env python3
"""bench.py -- Spartan proof throughput of the MI355X path on vPIN's LeNet trace.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json quotes its metric on the LeNet trace, and the whole trace fits one GPU): a
"step" is one pass over one LeNet inference trace = the 12 proofs vPIN produces for it
(vPIN_proof_generation/src/main.rs:14-46 per layer: 7 point-addition instances, 5
point-multiplication instances; L5's has 6000 point-mults = 20,784,000 constraints, 2^25 padded).
26.2 M unpadded constraints per step.  The witness is synthetic (seeded SplitMix64 points on the
curve E2, gadget-generated tables -- SURVEY.md 8(d)).  `--trace A` (BASELINE.json configs[1]; or
3_32, 7_256, E, L1..L7) times a single-network trace instead (2 instances).

What is proven per instance (default): the WHOLE SNARK the reference's my_lib_prove produces
(commit_test.rs:59-133): R1CSProof (two commitments, both ZK sum-checks, evaluation proof) +
inst_evals + R1CSEvalProof (SPARK: derefs commitment, 16 product circuits with batched cubic
sum-checks, hash-layer evaluation proofs).  SNARK::encode (the computation commitment, a function
of the circuit only) runs once per instance before the timed region; its time is reported as
encode_ms.  `--sat-only` times the R1CS satisfiability proof alone.
value = unpadded R1CS constraints proven per second, whole job (all ranks).  Multi-GPU: every
rank proves its own trace -- independent instances, no data-path collective -> weak scaling.

Schedule per rank: four host threads / HIP streams -- the largest instance; the other
point-mult instances on two; the point-add instances.  All but the first start when the largest
instance's phase-1 sum-check is done, so its kernels (the roofline sample) are timed undisturbed (--sat-only: two
lanes, the add lane starts after the largest instance).  `--serial`: one instance at a time.

The ONE line rank 0 prints is a flat, strict-JSON record of ~2 KB (tools/bench_common.py compact_line / dumps_line: at most
4096 bytes, scalars only inside `roofline` and `cpu_baseline`); everything else -- per-instance spans, kernel classes, the
reference span per instance and phase, power series, the strong sub-record in full -- goes to a side file whose path the line
carries in `detail` (--detail-out, default gpurun_out/bench_detail_n<N>.json).  The line carries
  roofline     : the fused phase-1 sum-check round kernel (sc_cubic3_kernel<true, true>), algorithmic bytes / HIP-event time
                 over the timed region, against the 8 TB/s HBM3E peak; `traffic` and `limiter_frac` (share of SIMD-cycles
                 with a VALU instruction in flight) are measured by this run itself (three `rocprofv3 --pmc` child passes
                 after everything else, N = 1 only); msm_* / prod_round_*: the kernels that own the step;
  cpu_baseline : the CPU oracle (a C restatement of the reference prover, oracle/) timed on this box's host threads on a
                 bounded sample (64 point-mults + 512 point-adds drawn like LeNet layer 1, ~6 s); --cpu-full-label A adds
                 CNN A's whole trace at full size, --cpu-single-thread the sample on one thread;
  value_reference_span, reference_span_*: the reference's own timed span per instance (witness inputs -> gadget -> is_sat
                 -> encode -> prove -> bytes), serially and on the lanes; with the work the reference does inside it and never
                 uses (third commitment measured; zlib digest measured with --digest, otherwise replayed and labelled so);
                 span_warning when a span exceeds 1.3 x the newest committed one;
  host_quota_cpus / host_throttled_ms: the CFS quota of the box and the time its cgroup was throttled inside the timed region;
  strong_*     : N > 1 only -- the same trace ONCE over all ranks (one LeNet trace, its 2^25 instance proven by all ranks
                 together; RCCL for device vectors), beside the weak headline; a watchdog and a signal handler
                 (vpin_crash_line_set) keep the weak line should that first outing of RCCL hang or take the process down;
  errors       : sections after the timed region that failed (the line still comes out; a parity failure exits 1).
Other modes: `--scaling strong [--rehearse W]`, `--trace T --concurrent K1,K2,..` (K copies of a small
trace at once on one GPU).
"""
import argparse
import json
import os
import sys
import threading
import time

if "--rehearse" in sys.argv:
    # W ranks as threads keep W persistent round kernels resident at once; HIP maps a process's streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing a queue run in order.  Before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))  # the other modes and the shared helpers: bench_common / bench_strong / bench_concurrent

from bench_common import SEED_C, SEED_P, PowerSampler, _golden_digests, live_pmc_traffic, emit, host_cpu_throttle  # noqa: E402

T_START = time.perf_counter()

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trace", default="lenet", help="lenet (L1..L7), one label (3_32, A, 7_256, E, L1..L7) or, with --scaling strong, "
                                                     "a comma-separated list of labels (a trace of one's own: L3,L1,L6,L7)")
    ap.add_argument("--label", default=None, help="alias of --trace for a single label")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mult", type=int, default=64, help="point-mults in the CPU baseline sample")
    ap.add_argument("--cpu-sample-add", type=int, default=512)
    ap.add_argument("--cpu-full-label", default="none", help="BASELINE configuration the CPU baseline ALSO proves at full size on all "
                    "host threads (A = configs[1], ~20 s more); none (default): the bounded sample only, the full-size figure of "
                    "the newest committed profile is carried in the detail file, labelled replayed")
    ap.add_argument("--cpu-single-thread", action="store_true", help="CPU baseline: the sample once more on ONE thread (~2.5x the time)")
    ap.add_argument("--table-slot", type=int, choices=[96, 128], default=None,
                    help="bytes per window-table entry (VPIN_TABLE_SLOT).  Default 128 when every rank has a GPU of its own (N = 1, or N > 1 "
                         "over nccl): one 128-byte line per gather, -2.2 %% on the step for 109 instead of 95 GiB of tables "
                         "(profiles/r06_ab_slot128.txt; the eval stream's budget goes to 100 GB and the free-memory share of a table to 0.45 "
                         "so that its 12-bit windows stay); 96 = the library's own default, kept where ranks share a GPU")
    ap.add_argument("--detail-out", default=None, help="path of the side file with the full record (default: "
                    "gpurun_out/bench_detail_n<N>.json under the repo root)")
    ap.add_argument("--serial", action="store_true", help="one instance at a time, one host thread")
    ap.add_argument("--host-buffers", action="store_true",
                    help="time vpin_sat_prove (instance + witness start in host memory: PCIe-inclusive; never the headline)")
    ap.add_argument("--pin-host-buffers", action="store_true", help="--host-buffers: the triplets and assignments are page-locked once "
                    "before the timed region (vpin_host_register), as a host that proves from the same buffers repeatedly would")
    ap.add_argument("--sat-only", action="store_true",
                    help="time only the R1CS satisfiability proof (SURVEY.md 8(a) rows H1-H10) instead of the whole SNARK")
    ap.add_argument("--snark", action="store_true", help="(default) whole SNARK: sat proof + inst_evals + SPARK R1CSEvalProof, "
                    "my_lib_prove in full; SNARK::encode runs once per instance before the timed region and is reported beside it")
    ap.add_argument("--mult-lanes", type=int, default=int(os.environ.get("VPIN_BENCH_MULT_LANES", "2")),
                    help="streams / host threads for the point-mult instances other than the largest")
    ap.add_argument("--lanes-spec", default=os.environ.get("VPIN_BENCH_LANES"),
                    help="explicit schedule: lanes separated by ';', instance names by ',' (first lane starts at once, the "
                         "others when its first instance's phase-1 sum-check is done); 'adds' = every point-add instance")
    ap.add_argument("--pipeline", action="store_true",
                    help="run the K timed steps back to back per lane, the smaller instances of all steps from one shared queue, "
                         "instead of finishing every step before the next starts (measured: 3 %% faster, +15 GB of pooled memory)")
    ap.add_argument("--concurrent", default=None, help="K or K1,K2,..: K independent copies of the (small) trace proven at once on one "
                    "GPU, one context each; reports traces/s and constraints/s at every K beside the single-trace latency")
    ap.add_argument("--gate", type=int, default=int(os.environ.get("VPIN_BENCH_GATE", "1")), choices=[0, 1, 2, 3],
                    help="when the other lanes start: 0 = at once (a spatial split: --cu-split), 1 = the largest instance's phase-1 sum-check is done (its MSMs then share every CU "
                         "with them), 2 = its derefs commitment is done, 3 = it is proven; with 2 and 3 it runs on an exclusive context until then")
    ap.add_argument("--cu-split", default=os.environ.get("VPIN_BENCH_CU_SPLIT"),
                    help="A,B[,layout[,after1]] or none: spatial split of the chip between the lanes (vpin_ctx_create_cumask) -- the largest "
                         "instance's stream gets A compute units of every XCD (0: every CU, unmasked), the other lanes' streams the LAST B of "
                         "every XCD; layout se (default): whole shader engines (8 CUs each), spread: consecutive CUs go round the four "
                         "engines; after1: the largest instance is confined only after its phase-1 sum-check "
                         "(vpin_ctx_set_cumask_after_phase1).  Default for the four-lane LeNet step: 24,8,se,after1")
    ap.add_argument("--small-queue", action="store_true", default=bool(os.environ.get("VPIN_BENCH_SMALL_QUEUE")),
                    help="the lanes other than the first take their instances of a step from ONE queue, longest first (durations measured "
                         "in the warm-up), instead of a fixed list per lane")
    ap.add_argument("--skip", default=None, help="comma-separated instance names left out of the trace (experiments: L5-mult)")
    ap.add_argument("--low-memory", action="store_true", help="vpin_ctx_set_low_memory on every context: the mem forests are built after "
                    "the ops forests are proven (LeNet step: 197 -> 181 GiB of HBM, +1.6 %% time)")
    ap.add_argument("--numa", choices=["local", "remote", "off"], default=os.environ.get("VPIN_BENCH_NUMA"),
                    help="pin the process to the host cores next to its GPU (local: default for N > 1), to the others (remote: measures "
                         "the sensitivity), or not at all (off: default for N = 1)")
    ap.add_argument("--span-lanes-repeats", type=int, default=1, help="reference-span pass on the lanes: this many passes (soak)")
    ap.add_argument("--no-span", action="store_true", help="skip the reference-span pass after the timed region")
    ap.add_argument("--digest", action="store_true", help="reference-span pass: MEASURE the zlib digest of bincode(A, B, C) (the "
                    "reference's unused Instance::new digest; ~30 s of one host core for the 2^25 instance); default: replayed from the "
                    "newest committed profile that measured it, labelled so")
    ap.add_argument("--no-digest", action="store_true", help="(accepted for older command lines; the digest is off unless --digest)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling sub-record (one trace over all ranks)")
    ap.add_argument("--strong-timeout", type=float, default=150.0, help="N > 1: seconds after which the strong sub-record is given up "
                    "(the line is printed with the failure recorded and every rank exits)")
    ap.add_argument("--only", choices=["mult", "add"], default=None, help="keep only the point-mult / point-add instances of the trace")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run verification of the last step's SNARKs")
    ap.add_argument("--no-prof", action="store_true", help="no HIP-event bracketing of kernels (no roofline object)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank proves its own trace; strong: ONE trace over all ranks -- the large instances "
                         "proven by all ranks together (vpin_comm), the small ones LPT-sharded")
    ap.add_argument("--rehearse", type=int, default=0,
                    help="with --scaling strong: W ranks as threads of this process on ONE GPU, serialised, to measure the "
                         "critical path of a W-GPU run (a model: no multi-GPU hardware involved)")
    ap.add_argument("--coop-all", action="store_true", help="--rehearse: every point-mult instance cooperatively (to see how each size scales)")
    ap.add_argument("--n1-step-ms", type=float, default=None, help="--rehearse: the measured N = 1 step of the same trace (bench.py default "
                    "line), so that the modelled speed-up is quoted against it")
    ap.add_argument("--rehearse-passes", type=int, default=3, help="serialised passes per cooperative instance; the quietest one is reported")
    ap.add_argument("--sub-coop-log2", type=float, default=22.0,
                    help="--scaling strong: instances of at least half this many (2^x) constraints, but below --coop-log2, are proven "
                         "by the first world/2 ranks together while the other ranks start on the small instances (world >= 4)")
    ap.add_argument("--coop-log2", type=int, default=24,
                    help="--scaling strong: instances of at least 2^this constraints are proven by all ranks together")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the N>1 runs (gloo: rehearsal with ranks sharing a GPU)")
    ap.add_argument("--no-roofline-pass", action="store_true",
                    help="skip the serial pass after the timed region that fills roofline.secondary (largest instance alone)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child passes at the "
                    "end of the default run (the value is then replayed from profiles/, and labelled so)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the roofline kernel from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


def ctypes_size_t():
    import ctypes
    return ctypes.c_size_t


def parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def pin_to_gpu_numa_node(local_rank, world, mode):
    """Pin this process (before the library's host threads exist: they inherit the mask) to the host cores next to its GPU:
    /sys/bus/pci/devices/<the card's PCI id>/local_cpulist.  A LeNet step makes ~2,500 host<->GPU round trips through pinned
    mailboxes; a rank whose threads and mailboxes sit on the other socket pays the inter-socket hop on each.
    mode: local | remote (the cores NOT next to the GPU: to measure the sensitivity) | off.  Ranks whose GPUs share a node
    split its cores evenly.  Returns a record for the bench line."""
    rec = {"mode": mode}
    if mode == "off" or not hasattr(os, "sched_setaffinity"):
        return rec
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        ndev = C.c_int(0)
        hip.hipGetDeviceCount(C.byref(ndev))
        lists = []
        for d in range(max(1, ndev.value)):
            buf = C.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, d) != 0:
                return dict(rec, error="hipDeviceGetPCIBusId failed")
            with open("/sys/bus/pci/devices/" + buf.value.decode().lower() + "/local_cpulist") as f:
                lists.append(parse_cpulist(f.read()))
        allowed = sorted(os.sched_getaffinity(0))
        mine = [c for c in lists[local_rank % len(lists)] if c in allowed]
        if mode == "remote":
            mine = [c for c in allowed if c not in set(lists[local_rank % len(lists)])]
        elif world > 1:
            # the ranks (local ranks 0..world-1 -> devices 0..) whose GPU has the same core list share it in equal slices
            peers = [r for r in range(world) if lists[r % len(lists)] == lists[local_rank % len(lists)]]
            k, n = peers.index(local_rank), len(peers)
            per = max(1, len(mine) // n)
            mine = mine[k * per:(k + 1) * per] or mine
        if not mine:
            return dict(rec, error="no allowed core in the selected set", node_cpulist=len(lists[local_rank % len(lists)]))
        os.sched_setaffinity(0, mine)
        rec.update({"cores": len(mine), "first_core": mine[0], "last_core": mine[-1],
                    "gpu_node_cores": len(lists[local_rank % len(lists)]), "allowed_before": len(allowed)})
    except (OSError, ValueError, IndexError) as e:
        rec["error"] = repr(e)
    return rec


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the newest round that has one (path, name) or (None, None)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return (c[-1], os.path.basename(c[-1])) if c else (None, None)


def lib_nnz(g):
    import vpin_amd
    L = vpin_amd.lib()
    return [int(L.vpin_dev_instance_nnz(g.h, m)) for m in range(3)]


def main():
    args = parse()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    # N > 1: every rank next to its own GPU (default for world > 1; --numa local / remote on one GPU measures the sensitivity)
    numa_mode = args.numa or ("local" if world_env > 1 else "off")
    affinity_rec = pin_to_gpu_numa_node(int(os.environ.get("LOCAL_RANK", "0")), world_env, numa_mode)
    if "VPIN_HOST_THREADS" not in os.environ:
        # The library's OpenMP teams (blind terms of the ZK rounds, generator derivation) are sized per context.  What bounds them
        # is the CPUs the job may use AT ONCE -- its cgroup's CFS quota (16 on a one-GPU box of a 256-thread host), not the
        # host's core count: a process that runs more threads than that for a while is stopped for the rest of every 100 ms
        # period.  Round 6 found this behind the ~20 ms stalls of the W = 8 rehearsal (8 rank-threads x teams of 4: 24 of ~90
        # periods throttled; with teams of 1 or 2 none, and the unfiltered model falls from 231 to 148-164 ms).
        quota = host_cpu_throttle().get("quota_cpus")
        if "cores" in affinity_rec:   # pinned: this rank's own cores
            cores = affinity_rec["cores"] * world_env
        else:
            cores = (os.cpu_count() or 16) if world_env > 1 else min(16, os.cpu_count() or 16)
        if quota:
            cores = min(cores, int(quota))
        if args.scaling == "strong" and args.rehearse:   # W rank-threads, one computing at a time, W teams alive
            os.environ["VPIN_HOST_THREADS"] = str(max(1, min(4, cores // (2 * args.rehearse))))
        else:
            # the four-lane step: 16 // 4 = 4 per lane (round 6 re-measured 2 against 4 on one box: 375.6 against 372.2 ms -- the
            # lanes' host sections are short bursts, the throttled periods they cause cost less than the narrower teams)
            os.environ["VPIN_HOST_THREADS"] = str(max(2, min(4, cores // (world_env * 4))))
    # window-table layout (before the library is loaded: the free-memory share is read once)
    own_gpu = world_env == 1 or args.backend == "nccl"
    slot = args.table_slot or (int(os.environ["VPIN_TABLE_SLOT"]) if os.environ.get("VPIN_TABLE_SLOT") else
                               (128 if (own_gpu and args.scaling == "weak" and not args.concurrent) else 96))
    os.environ["VPIN_TABLE_SLOT"] = str(slot)
    if slot == 128:
        os.environ.setdefault("VPIN_SPARK_GENS_BUDGET_GB", "100")
        os.environ.setdefault("VPIN_GENS_FREE_FRACTION", "0.45")
    if args.scaling == "strong":
        from bench_strong import main_strong
        return main_strong(args)
    if args.concurrent:
        ks = [int(x) for x in str(args.concurrent).split(",")]
        if len(ks) == 1:  # (a sweep's parent only starts children)
            if "VPIN_HOST_THREADS" not in os.environ:
                # K proving threads share the host's cores; a one-GPU box gets 16 of them whatever os.cpu_count() says, and
                # 8 contexts x 8 OpenMP threads on 16 cores measured 1.0x of the single-trace rate where 8 x 2 gives 3x
                cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8))
                os.environ["VPIN_HOST_THREADS"] = str(max(1, min(4, cores // max(1, ks[0]))))
            # one hardware queue per stream: with the runtime's default of 4, streams 5.. share a queue with another stream and a
            # resident round kernel of one proof holds up the other's launches
            os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(24, ks[0]))))
        from bench_concurrent import main_concurrent
        return main_concurrent(args)
    args.snark = not args.sat_only and not args.host_buffers
    host_snark = args.host_buffers and not args.sat_only   # the whole SNARK from host buffers: upload + SNARK::encode + prove per proof
    import vpin_amd
    from vpin_amd import gadgets as G
    from vpin_amd.dist import Group, env_rank

    rank, local_rank, world = env_rank()
    # torch is plumbing for the N > 1 control path (process group, barrier, max over ranks); one rank needs none of it and
    # talks to the HIP runtime directly (hipDeviceSynchronize / hipMemGetInfo), which also spares the N = 1 run the import
    torch = None
    if world > 1:
        import torch
        if args.backend != "nccl":  # rehearsal: ranks share the visible GPU(s), the control path goes over gloo
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
    import ctypes as C_
    hip_rt = C_.CDLL("libamdhip64.so")

    def device_synchronize():
        if torch is not None:
            torch.cuda.synchronize()
        else:
            assert hip_rt.hipSetDevice(local_rank) == 0 and hip_rt.hipDeviceSynchronize() == 0

    def mem_get_info():
        if torch is not None:
            return torch.cuda.mem_get_info(local_rank)
        f, t = C_.c_size_t(0), C_.c_size_t(0)
        assert hip_rt.hipSetDevice(local_rank) == 0 and hip_rt.hipMemGetInfo(C_.byref(f), C_.byref(t)) == 0
        return f.value, t.value
    grp = Group(backend=args.backend, device=torch.device("cuda", local_rank) if (world > 1 and args.backend == "nccl") else None)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    trace = args.label or args.trace
    labels = list(G.LENET) if trace == "lenet" else [trace]

    # ---- synthetic workload: witness inputs on the host, instances built on the device (outside the timed region) ----
    t0 = time.perf_counter()
    work = []  # (name, kind, inputs, unpadded constraints)
    for lab in labels:
        m = G.synthetic_mult_inputs(lab)
        if m is not None:
            work.append((f"{lab}-mult", "mult", m, 3464 * len(m[0])))
        a = G.synthetic_add_inputs(lab)
        work.append((f"{lab}-add", "add", a, 10 * len(a[4])))
    if args.only:
        work = [w for w in work if w[1] == args.only]
    if args.skip:
        work = [w for w in work if w[0] not in args.skip.split(",")]
    setup_s = time.perf_counter() - t0
    cons = {w[0]: w[3] for w in work}
    total_cons_step = sum(cons.values())
    inputs_of = {w[0]: (w[1], w[2]) for w in work}

    # lanes: mult instances largest first on lane 0; add instances on lane 1 (started after the
    # largest mult instance unless --serial)
    mults = sorted([w for w in work if w[1] == "mult"], key=lambda w: -w[3])
    adds = [w for w in work if w[1] == "add"]
    lanes = [mults + adds] if (args.serial or not mults) else [mults, adds]
    if args.snark and not args.serial and len(mults) > 2:
        # whole SNARKs are long enough to be worth three host threads / streams: the largest instance
        # alone, the other point-mult instances, the point-add instances; the other lanes start when the
        # largest instance's phase-1 sum-check is done, so its kernels (the roofline sample) are timed undisturbed
        rest = [[] for _ in range(max(1, min(args.mult_lanes, len(mults) - 1)))]
        for w in mults[1:]:  # largest first onto the least loaded lane
            min(rest, key=lambda l: sum(x[3] for x in l)).append(w)
        lanes = [mults[:1]] + rest + [adds]
    if args.lanes_spec:
        by_name = {w[0]: w for w in work}
        lanes = []
        for part in args.lanes_spec.split(";"):
            lane = []
            for nm in part.split(","):
                nm = nm.strip()
                lane += adds if nm == "adds" else [by_name[nm]]
            lanes.append(lane)
        assert sorted(w[0] for l in lanes for w in l) == sorted(w[0] for w in work), "--lanes-spec must name every instance once"
    # the lanes of small, latency-bound instances get high-priority streams: their one-workgroup round
    # kernels are then dispatched ahead of the large instance's queued workgroups instead of behind them
    prios = [0] + [-1] * (len(lanes) - 1) if not os.environ.get("VPIN_BENCH_NO_PRIO") else [0] * len(lanes)
    cu_split = None
    default_split = (args.cu_split is None and trace == "lenet" and args.snark and len(lanes) == 4 and not args.serial
                     and not args.only and not args.skip and not args.pipeline and (world == 1 or args.backend == "nccl"))
    if default_split:
        args.cu_split = "24,8,se,after1"
    if args.cu_split and args.cu_split != "none":
        # Spatial split (round 5).  Mask bit k of hipExtStreamCreateWithCUMask is, on this 8-XCD part, CU k/8 of XCD k%8, and CU n
        # of an XCD sits in shader engine n%4 (tools/ubench_cumask -> profiles/r05_cumask_map.txt); an XCD without any bit set
        # gets ALL its CUs, so whole XCDs cannot be taken away from a stream, only CUs of every XCD.
        parts = args.cu_split.split(",")
        ca, cb, layout = int(parts[0]), int(parts[1]), (parts[2] if len(parts) > 2 else "se")
        per_xcd, nx, nse = 32, 8, 4
        if layout in ("se", "b"):      # whole shader engines: CUs 0..ca-1 in SE-major order (SE = n // 8)
            bit = lambda n, x: x + nx * ((n // (per_xcd // nse)) + nse * (n % (per_xcd // nse)))
        else:                          # "i" / "spread": the runtime's own order, consecutive CUs go round the SEs
            bit = lambda n, x: n * nx + x
        big = [bit(n, x) for x in range(nx) for n in range(ca)] if ca > 0 else None
        small = [bit(n, x) for x in range(nx) for n in range(per_xcd - cb, per_xcd)]
        if len(parts) > 3 and parts[3] == "same":  # experiments: the first lane on the other lanes' CUs too
            big, ca = small, 0
        after1 = len(parts) > 3 and parts[3] == "after1" and big is not None
        cu_split = {"largest_instance_cus_per_xcd": ca or per_xcd, "other_lanes_cus_per_xcd": cb,
                    "layout": "whole shader engines" if layout in ("se", "b") else "spread over the shader engines",
                    "disjoint": bool(ca) and ca + cb <= per_xcd,
                    "largest_instance_masked": "after its phase-1 sum-check (vpin_ctx_set_cumask_after_phase1)" if after1 else "always"}
        ctxs = [vpin_amd.Context(local_rank, cu_mask=big) if (big and not after1) else vpin_amd.Context(local_rank, priority=0)] + \
               [vpin_amd.Context(local_rank, cu_mask=small) for _ in range(len(lanes) - 1)]
        big_mask = big
        if after1:
            ctxs[0].set_cumask_after_phase1(big)
    else:
        ctxs = [vpin_amd.Context(local_rank, priority=prios[li]) for li in range(len(lanes))]
    # the lanes share the GPU: the row-commitment MSM leaves room on every CU -- unless the largest instance has CUs of its own
    l0_shared = not (cu_split and cu_split["disjoint"]) and len(lanes) > 1
    if os.environ.get("VPIN_BENCH_L0_SHARED"):
        l0_shared = os.environ["VPIN_BENCH_L0_SHARED"] != "0"
    # ... and the other lanes' row commitments run three workgroups per CU on THEIR CUs when the split is disjoint (measured:
    # 394 -> 386 ms against one per CU, profiles/r05_ab_cumask.txt)
    small_shared = os.environ.get("VPIN_BENCH_SMALL_SHARED", "0" if (cu_split and cu_split["disjoint"]) else "1") != "0"
    for li, cx in enumerate(ctxs):
        cx.set_shared_device(l0_shared if li == 0 else small_shared)
        if args.low_memory:
            cx.set_low_memory(True)

    def barrier():
        device_synchronize()
        grp.barrier()
        device_synchronize()

    def build_instance(cx, kind, inp):
        """gadget + witness + Instance::new on the device (vpin_gadget_point_*_dev)"""
        return cx.gadget_point_mult_dev(*inp) if kind == "mult" else cx.gadget_point_add_dev(*inp)

    # instance + the three assignments resident in HBM before the timed region (the PCIe-inclusive
    # variant is vpin_sat_prove / --host-buffers; its rate is noted in DESIGN.md)
    resident, dicts, decomms, encode_ms, comm_bytes, verify_meta, last_proof, dev_insts = {}, {}, {}, {}, {}, {}, {}, {}
    pinned_tokens = []
    import numpy as np
    t0 = time.perf_counter()
    for li, lane in enumerate(lanes):
        for name, kind, inp, _ in lane:
            cx = ctxs[li]
            if args.host_buffers:
                inst = G.point_mult(*inp) if kind == "mult" else G.point_add(*inp)
                d = inst.as_dict()
                dicts[name] = d
                num_vars = d["num_vars"]
                inst.free()
                if args.pin_host_buffers:   # the host keeps its triplets and assignments page-locked (vpin_host_register)
                    from vpin_amd.capi import host_register
                    for key in ("A", "B", "C"):
                        d[key] = tuple(np.ascontiguousarray(x) for x in d[key])
                        pinned_tokens.extend(host_register(x) for x in d[key] if x.nbytes)
                    for key in ("vars_para", "vars_input", "vars"):
                        d[key] = np.ascontiguousarray(d[key])
                        pinned_tokens.append(host_register(d[key]))
                if host_snark:
                    cx.spark_prepare(d["num_cons"], d["num_vars"], max(len(d[k][0]) for k in "ABC"))
            else:
                g = build_instance(cx, kind, inp)
                assert g.num_cons_unpadded == cons[name]
                dev_insts[name] = g
                resident[name] = (g.r1cs, g.vars_para, g.vars_input, g.vars, g.inputs)
                num_vars = g.num_vars
            # generator sets before the timed region, largest polynomial first (lane 0 comes first), so every
            # lane shares the one window table per label
            cx.sat_prepare(num_vars)
            if args.snark:
                # SNARK::encode: once per circuit, outside the timed region (the computation commitment
                # does not depend on the witness); first call also builds the generator table
                g.spark_encode()[0].free()
                te = time.perf_counter()
                decomms[name], comm = g.spark_encode()
                encode_ms[name] = round((time.perf_counter() - te) * 1e3, 3)
                comm_bytes[name] = len(comm)
                verify_meta[name] = (comm, {"inputs": g.inputs, "num_inputs": g.num_inputs})
    upload_s = time.perf_counter() - t0
    if not os.environ.get("VPIN_BENCH_NO_TRIM"):
        for cx in ctxs:
            cx.pool_trim()   # the set-up's temporaries (gadget synthesis, SNARK::encode) are of no use to the proofs
    lane_names = [[w[0] for w in lane] for lane in lanes]

    last_spans, proof_bytes = {}, {}
    errors = {}   # sections after the timed region that failed (the line still comes out); 'parity' makes the exit code 1
    import numpy as np
    progress = np.zeros(4, dtype=np.int32)
    if args.snark and len(lanes) >= 3:
        ctxs[0].set_progress_flag(progress)

    proof_ms = {}   # last measured duration per instance (the lane plan after the warm-up, the shared queue's order)

    def prove(li, name):
        cx = ctxs[li]
        if args.snark:
            di, tp, ti, tv, inp = resident[name]
            t_p = time.perf_counter()
            r = cx.snark_prove_resident(di, decomms[name], tp, ti, tv, inp, SEED_C, SEED_P)
            proof_ms[name] = (time.perf_counter() - t_p) * 1e3
            last_spans[name] = dict(cx.sat_timings(), **{"spark_" + k: v for k, v in cx.spark_timings().items() if k != "_"})
            proof_bytes[name] = len(r["proof"])
            last_proof[name] = r
            return len(r["proof"])
        if host_snark:
            r = cx.snark_prove(dicts[name], SEED_C, SEED_P)
        elif args.host_buffers:
            r = cx.sat_prove(dicts[name], SEED_C, SEED_P)
        else:
            di, tp, ti, tv, inp = resident[name]
            r = cx.sat_prove_resident(di, tp, ti, tv, inp, SEED_C, SEED_P)
        last_spans[name] = cx.sat_timings()  # thread-local in the library: read on the proving thread
        proof_bytes[name] = len(r["proof"])
        return len(r["proof"])

    queue_state = {"lock": threading.Lock(), "order": [], "pos": 0, "all_on_every_lane": False}

    def run_lane(li, gate):
        if gate is not None:
            gate.wait()
        if args.small_queue and li > 0 and len(lanes) >= 3:
            if queue_state["all_on_every_lane"]:   # warm-up: every context meets every instance size once
                for name in queue_state["order"]:
                    prove(li, name)
                return
            while True:
                with queue_state["lock"]:
                    k = queue_state["pos"]
                    queue_state["pos"] += 1
                if k >= len(queue_state["order"]):
                    return
                name = queue_state["order"][k]
                t1 = time.perf_counter()
                prove(li, name)
                proof_ms[name] = (time.perf_counter() - t1) * 1e3
            return
        for name in lane_names[li]:
            prove(li, name)

    def step():
        if len(lanes) == 1:
            run_lane(0, None)
            return
        if len(lanes) >= 3:
            progress[0] = 0
            gate = threading.Event()
            if args.small_queue:
                small = [n for l in lane_names[1:] for n in l]
                queue_state["order"] = sorted(small, key=lambda n: -proof_ms.get(n, cons[n] * 1e-4))
                queue_state["pos"] = 0

            if args.gate > 1:
                ctxs[0].set_shared_device(False)  # the largest instance has the chip to itself until the gate opens

            def watch():
                while progress[0] < min(args.gate, 2) and not gate.is_set():
                    time.sleep(0.0005)
                if args.gate == 3:   # ... until the largest instance is proven
                    gate.wait()
                ctxs[0].set_shared_device(l0_shared)
                gate.set()
            ts = [threading.Thread(target=run_lane, args=(li, gate)) for li in range(1, len(lanes))] + [threading.Thread(target=watch)]
            for t in ts:
                t.start()
            run_lane(0, None)
            gate.set()
            for t in ts:
                t.join()
            return
        gate = threading.Event()
        t = threading.Thread(target=run_lane, args=(1, gate))
        t.start()  # ctypes releases the GIL inside the library calls
        if len(lane_names[0]) == 1:
            gate.set()  # single-network trace: nothing else would overlap the add instance
        for i, name in enumerate(lane_names[0]):
            prove(0, name)
            if i == 0:
                gate.set()  # the add lane starts once the largest instance is proven
        gate.set()
        t.join()

    # Pipelined schedule (whole SNARKs, >= 3 lanes): lane 0 proves the largest instance of step 0, 1, ... back to
    # back; the other lanes take the remaining instances of all K steps from one queue (largest first within a
    # step), so no lane idles at a step boundary while another finishes.  All K x 12 proofs start and end inside the
    # timed region.  Any context can prove any resident instance (read-only); the warm-up lets every context see
    # every instance size once, so no generator view is derived inside the timed region.  The other lanes start when
    # step 0's largest instance has finished its phase-1 sum-check: those launches are the roofline sample.
    pipelined = args.snark and len(lanes) >= 3 and args.pipeline and not args.lanes_spec
    others = sorted([w for lane in lanes[1:] for w in lane], key=lambda w: -w[3]) if pipelined else []
    other_names = [w[0] for w in others]
    lane0_snapshot = {}

    def run_pipelined(nsteps, warm):
        progress[0] = 0
        gate = threading.Event()
        jobs = [nm for _ in range(nsteps) for nm in other_names]
        lock = threading.Lock()
        pos = [0]

        def watch():
            while progress[0] == 0 and not gate.is_set():
                time.sleep(0.0005)
            gate.set()

        def worker(li):
            gate.wait()
            if warm:
                for nm in other_names:
                    prove(li, nm)
                return
            while True:
                with lock:
                    k = pos[0]
                    pos[0] += 1
                if k >= len(jobs):
                    return
                prove(li, jobs[k])

        ts = [threading.Thread(target=worker, args=(li,)) for li in range(1, len(lanes))] + [threading.Thread(target=watch)]
        for t in ts:
            t.start()
        for k in range(nsteps):
            for name in lane_names[0]:
                prove(0, name)
            if k == 0 and not warm:
                lane0_snapshot.update(ctxs[0].prof_read())  # step 0: the largest instance's phase 1 ran alone
        gate.set()
        for t in ts:
            t.join()

    if pipelined:
        for _ in range(args.warmup):
            run_pipelined(1, True)
    else:
        if args.small_queue and len(lanes) >= 3:
            queue_state["all_on_every_lane"] = True
            step()   # an extra warm-up pass: any lane may prove any of the small instances later
            queue_state["all_on_every_lane"] = False
        for wi in range(max(args.warmup, 2 if (args.snark and len(lanes) == 4 and not args.lanes_spec and not args.small_queue) else 0)):
            step()
            if wi == 0 and args.snark and len(lanes) == 4 and not args.lanes_spec and not args.small_queue and len(lane_names[1]) + len(lane_names[2]) > 2:
                # the two lanes of the smaller point-mult instances, re-planned from the durations just measured (longest
                # first onto the lane that is free first): constraint counts misjudge them -- padding, latency-bound phases
                names = sorted(lane_names[1] + lane_names[2], key=lambda n: -proof_ms.get(n, 0.0))
                plan, load = [[], []], [0.0, 0.0]
                for n in names:
                    k = 0 if load[0] <= load[1] else 1
                    plan[k].append(n)
                    load[k] += proof_ms.get(n, 0.0)
                lane_names[1], lane_names[2] = plan

    for cx in ctxs:
        cx.prof_reset()
        cx.prof_enable(not args.no_prof)
    sampler = PowerSampler(local_rank)
    barrier()
    sampler.start()
    thr0 = host_cpu_throttle()
    t0 = time.perf_counter()
    if pipelined:
        run_pipelined(args.steps, False)
    else:
        for _ in range(args.steps):
            step()
    for cx in ctxs:
        cx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    thr1 = host_cpu_throttle()
    host_cpu = dict(thr1, throttled_ms_in_timed_region=round(thr1.get("throttled_ms", 0.0) - thr0.get("throttled_ms", 0.0), 2),
                    nr_throttled_in_timed_region=thr1.get("nr_throttled", 0) - thr0.get("nr_throttled", 0),
                    cpu_count=os.cpu_count(), host_threads_per_lane=int(os.environ.get("VPIN_HOST_THREADS", "0"))) if thr1 else {}
    power_rec = sampler.stop()
    stats, stats_lane0 = {}, {}
    for ci, cx in enumerate(ctxs):
        per_ctx = cx.prof_read()
        if ci == 0:
            stats_lane0 = {n: dict(v) for n, v in (lane0_snapshot or per_ctx).items()}
        for name, v in per_ctx.items():
            a = stats.setdefault(name, {"launches": 0, "ms": 0.0, "alg_bytes": 0.0})
            for kk in a:
                a[kk] += v[kk]
        cx.prof_enable(False)

    free_b, total_b = mem_get_info()
    hbm_used_gb = round((total_b - free_b) / 2**30, 1)
    # what the memory in use is made of: the shared window tables; per lane the pool's blocks that back live handles
    # (instances, assignments, decommitments: the resident inputs of the step) and the blocks cached for the next proof's
    # temporaries (forests, derefs, partials); the rest = runtime, torch, the HIP contexts' own allocations
    L_ = vpin_amd.lib()
    L_.vpin_gens_shared_bytes.restype = ctypes_size_t()
    L_.vpin_gens_shared_bytes.argtypes = [__import__("ctypes").c_int]
    gib = lambda b: round(b / 2**30, 2)
    pools = [cx.pool_stats() for cx in ctxs]
    tables_b = int(L_.vpin_gens_shared_bytes(local_rank))
    hbm_breakdown = {
        "window_tables_gib": gib(tables_b),
        "resident_inputs_gib_per_lane": [gib(t - c_) for t, c_, _ in pools],
        "cached_temporaries_gib_per_lane": [gib(c_) for _, c_, _ in pools],
        "resident_inputs_gib": gib(sum(t - c_ for t, c_, _ in pools)), "cached_temporaries_gib": gib(sum(c_ for _, c_, _ in pools)),
        "other_gib": gib((total_b - free_b) - tables_b - sum(t for t, _, _ in pools)),
        "note": "after the timed region, no proof running: a lane's pool keeps the temporaries of its largest proof for the next one "
                "(hipMalloc / hipFree per proof cost more than the proofs of the small instances)"}
    elapsed = grp.max_over_ranks(elapsed)
    value = total_cons_step * args.steps * world / elapsed

    line = {
        "metric": ("R1CS constraints/sec, whole Spartan SNARK (sat proof + SPARK evaluation proof; vPIN point-mult + point-add instances)"
                   if (args.snark or host_snark) else "R1CS constraints/sec, Spartan sat proof (vPIN point-mult + point-add instances)"),
        "value": value,
        "unit": "constraints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u256 (integers mod q and mod p = 2^255-19, 32-bit limbs)",
        "data": "synthetic",
        "config": {
            "workload": (("vPIN LeNet trace (layers L1..L7): 12 " + ("SNARKs" if args.snark else "sat proofs") + " per step") if trace == "lenet"
                         else f"vPIN trace '{trace}': " + ("SNARKs" if args.snark else "sat proofs") + " of its point-mult and point-add instances"),
            "instances": cons,
            "constraints_unpadded_per_step": total_cons_step,
            "scope": ("my_lib_prove in full: R1CSProof + inst_evals + R1CSEvalProof (SPARK); SNARK::encode once per circuit "
                      "outside the timed region (encode_ms)" if args.snark else
                      "R1CSProof (commitments + both ZK sum-checks + evaluation proof); SPARK encode/eval proof not included"),
            "parallelism": f"one trace per rank x {world} rank(s), no collective; per rank "
                           + ("instances proven serially" if len(lanes) == 1 else
                              (f"{len(lanes)} streams, the K timed steps pipelined: the largest instance of every step back to back on one, the "
                               f"other instances of all steps from a shared queue on {len(lanes) - 1} (started after step 0's largest instance's phase-1 sum-check)"
                               if pipelined else
                               f"{len(lanes)} streams: largest instance | other mult instances on {len(lanes) - 2} | add instances (all but the first "
                               "start after the largest's phase-1 sum-check)")
                              if len(lanes) >= 3 else
                              "mult instances serially, add instances on a second stream after the largest"),
            "inputs": (("host buffers (PCIe-inclusive: triplets + assignments cross the bus, CSR/CSC" + (", SNARK::encode" if host_snark else "")
                        + " built per proof on the device" + ("; host buffers page-locked" if args.pin_host_buffers else "") + ")")
                       if args.host_buffers else "resident in HBM"),
        },
    }
    if cu_split:
        line["config"]["cu_split"] = cu_split
    line["config"]["parallelism_short"] = (f"one trace per rank x {world} rank(s), no collective; {len(lanes)} stream(s) per rank")
    detail_path = args.detail_out or os.path.join(ROOT, "gpurun_out", f"bench_detail_n{world}.json")
    line["config"]["lanes"] = lane_names
    line["config"]["low_memory"] = bool(args.low_memory)
    line["config"]["table_slot_bytes"] = int(os.environ.get("VPIN_TABLE_SLOT", "96"))
    line["host_affinity"] = affinity_rec
    line["host_cpu"] = host_cpu

    # ---- roofline of the fused sum-check round kernel ----
    # with several streams the event time of a kernel on one stream includes waiting for CUs taken by the
    # other streams' kernels; lane 0 (the largest instance) runs its phase-1 sum-check before the other lanes
    # start, so the roofline is taken over lane 0's launches only
    k = (stats_lane0 if len(lanes) > 1 else stats).get("sc_cubic_fused")
    k_all = stats.get("sc_cubic_fused")
    if k and k["ms"] > 0:
        achieved = k["alg_bytes"] / (k["ms"] * 1e-3) / 1e9
        traffic = args.pmc_traffic
        pmc_all, pmc_l5 = {}, {}
        pth, pmc = newest_profile("pmc_traffic.json")  # separate rocprofv3 --pmc passes, see the file's _how
        if pth:
            with open(pth) as f:
                doc = json.load(f)
            pmc_all, pmc_l5 = doc.get("bench_default", {}), doc.get("bench_L5_mult", {})
        traffic_source = "--pmc-traffic (this invocation's caller)" if traffic is not None else None
        if traffic is None and trace in ("lenet", "L5"):   # (the committed PMC passes are the 2^25 instance's: no other trace's launches)
            # the largest instance's launches alone (bench_L5_mult section): the default run's average mixes every instance's
            traffic = (pmc_l5 or pmc_all).get("sc_cubic3_kernel<true, true>", {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = f"replayed from profiles/{pmc} (separate rocprofv3 --pmc passes of the same command; NOT measured in this run)"
        # VALU-issue ceiling of the same launches: pairs per second the chip can issue (one wave-instruction per SIMD per 4
        # cycles, tools/isa_counts.py: VALU instructions per pair of this kernel's loop)
        valu_per_pair, valu_frac = None, None
        isa_path, isa_name = newest_profile("isa_counts.json")
        try:
            with open(isa_path) as f:
                valu_per_pair = json.load(f)["sc_cubic3_kernel<true, true>"]["valu_per_pair"]
            ncu, hz = ctxs[0].device_props()
            pairs_s = (k["alg_bytes"] / 384.0) / (k["ms"] * 1e-3)  # 4 tables x 32 B x 1.5 = 192 B per entry = 384 B per pair
            valu_frac = pairs_s / (ncu * 4 * 64 * hz / (4.0 * valu_per_pair))
        except (OSError, KeyError, ZeroDivisionError, TypeError):
            pass
        # bytes the eq-factored kernel really moves per launch: 3 tables read (len) and written (len/2), the suffix
        # table read once per pair (len/4): 152*len against the 192*len of the reference's 4-table formulation
        actual = k["alg_bytes"] * 152.0 / 192.0
        achieved_actual = actual / (k["ms"] * 1e-3) / 1e9
        line["roofline"] = {
            "kernel": "sc_cubic3_kernel<true, true> (fused fold + cubic round evaluation of phase 1, eq-factored, leading-coefficient form; "
                      "rounds with > 512 pairs)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
            "limiter": "valu-issue", "limiter_frac": None, "limiter_frac_static_model": valu_frac,
            "valu_instructions_per_pair_static": valu_per_pair, "static_model_source": isa_name,
            "limiter_note": "priced against HBM as the contract asks, but the kernel stops at VALU issue first.  limiter_frac (round 5) is "
                            "MEASURED: the share of SIMD-cycles in which a VALU instruction is in flight, from the SQ counters of the "
                            "kernel's own dispatches (4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x CUs x 4 SIMDs); a third rocprofv3 "
                            "--pmc child pass of this run, or replayed from profiles/ when that pass is skipped) -- at most 1 by "
                            "construction.  limiter_frac_static_model is the old yardstick (achieved pairs/s over CUs x 4 SIMDs x 64 lanes "
                            "x nominal clock / (4 cycles x the loop's static VALU instruction count)): it exceeded 1 in round 4 because the "
                            "static count of the loop body over-counts what a pair executes (measured_valu_instructions_per_pair); real "
                            "HBM traffic is frac_actual of peak",
            "achieved_note": "algorithmic bytes of the reference's 4-table formulation (SURVEY.md 8(d): 4*32*1.5*len per launch) / HIP-event time",
            "achieved_actual": achieved_actual, "frac_actual": achieved_actual / HBM_PEAK_GBPS,
            "actual_note": "bytes this kernel really moves (3 tables + the suffix table, 152*len: the eq table is never read; PMC "
                           "agrees, see traffic) / the same time; the kernel sits between the VALU-issue and the HBM limit",
            "launches": k["launches"], "avg_launch_us": k["ms"] * 1e3 / k["launches"],
            "alg_bytes_per_launch": k["alg_bytes"] / k["launches"],
            "scope": ((f"launches of the largest instance ({lane_names[0][0]}) in step 0 of the timed region, before the other lanes start: "
                       if pipelined else f"launches of the largest instance ({lane_names[0][0]}) only: ")
                      + f"{k['launches']} of {k_all['launches']} launches, "
                      f"{100.0 * k['alg_bytes'] / k_all['alg_bytes']:.1f}% of the kernel's algorithmic bytes in the timed region; the other "
                      "instances run concurrently on other streams" if len(lanes) > 1 else "all launches in the timed region"),
        }
    digest_jobs = []  # (instance, bincode(A, B, C)): compressed at the end of the run (reference_span.dead_work)
    line["kernels"] = {name: {"launches": v["launches"], "ms": round(v["ms"], 4),
                              "GBps_alg": (v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] else None}
                       for name, v in stats.items()}
    line["spans_ms_last_step"] = {n: {kk: round(vv * 1e3, 3) for kk, vv in sp.items()} for n, sp in last_spans.items()}
    if power_rec:
        line["power_during_timed_region"] = power_rec  # rank 0's card
    line["setup_s"] = {"synthetic_witness_inputs": round(setup_s, 3),
                       "device_gadgets_generator_tables_encode": round(upload_s, 3)}
    line["proof_bytes"] = proof_bytes
    line["hbm_in_use_gib_after_timed_region"] = hbm_used_gb  # instances, decommitments, generator tables, pooled temporaries
    line["hbm_breakdown"] = hbm_breakdown
    if args.snark:
        line["encode_ms"] = encode_ms
        line["comm_bytes"] = comm_bytes
        if not args.no_verify:
            # after the timed region: every proof of the last step through the product's verifier
            # (my_lib_verify: vpin_snark_verify), at full size
            tv = time.perf_counter()
            verified = {}
            for li, names in enumerate(lane_names):
                for name in names:
                    comm, meta = verify_meta[name]
                    verified[name] = bool(ctxs[li].snark_verify(meta, dict(last_proof[name], comm=comm)))
            line["verified"] = verified
            line["verify_s_first"] = round(time.perf_counter() - tv, 2)  # includes deriving the verifier's generator sets (once per context)
            tv = time.perf_counter()
            for li, names in enumerate(lane_names):
                for name in names:
                    comm, meta = verify_meta[name]
                    verified[name] = verified[name] and bool(ctxs[li].snark_verify(meta, dict(last_proof[name], comm=comm)))
            line["verify_s"] = round(time.perf_counter() - tv, 2)        # the same 12 verifications again: the steady state
            if not all(verified.values()):
                errors["parity"] = "rejected by the verifier: " + ",".join(n for n, v in verified.items() if not v)
            # and their bytes against the oracle's digests of the same instances and seeds (a committed fixture: data, no
            # oracle call) -- the proofs of the timed region, made with all lanes running, equal the oracle's byte for byte
            gold_path = os.path.join(ROOT, "tests", "golden", "config_digests.json")
            if os.path.exists(gold_path):
                import hashlib
                with open(gold_path) as f:
                    gold = json.load(f)["cases"]
                same = {n: hashlib.sha256(last_proof[n]["proof"]).hexdigest() == gold[n]["snark_sha256"]
                        for names in lane_names for n in names if n in gold and "snark_sha256" in gold[n]}
                line["bytes_equal_oracle_digest"] = same
                if not all(same.values()):
                    errors["parity"] = "proof bytes differ from the oracle's digest: " + ",".join(n for n, v in same.items() if not v)
        # ---- the reference's own span (proof_point_mult.rs:24-101: witness inputs -> gadget + witness ->
        # is_sat -> SNARK::encode -> my_lib_prove), one instance after the other, generator tables warm; the
        # resident copies are released first.  Same seeds, so the bytes must equal the timed region's proofs.
        # ---- roofline.secondary: the kernels that own the step, measured on the largest instance proven ALONE (one
        # stream, nothing else on the device) right after the timed region, so a kernel's event time is its own ----
        try:
            if "roofline" in line and not args.no_roofline_pass and not args.no_prof:
                big = lane_names[0][0]
                cx = ctxs[0]
                cx.set_shared_device(False)
                if cu_split and cu_split["largest_instance_masked"].startswith("after"):
                    cx.set_cumask_after_phase1(None)   # alone = on every CU of the chip
                cx.prof_reset()
                cx.prof_enable(2)  # level 2: also count the table additions of the row commitments
                di, tp, ti, tv_, inp = resident[big]
                psamp = PowerSampler(local_rank, interval_s=0.004)
                psamp.start()
                t1 = time.perf_counter()
                cx.snark_prove_resident(di, decomms[big], tp, ti, tv_, inp, SEED_C, SEED_P)
                cx.sync()
                alone_ms = (time.perf_counter() - t1) * 1e3
                tms = cx.spark_timings()
                # the derefs commitment (the proof's largest row commitment) runs from the end of the sat part for derefs_commit seconds
                msm_power = psamp.stop(t1 + tms.get("sat", 0.0) + 0.1 * tms.get("derefs_commit", 0.0), t1 + tms.get("sat", 0.0) + 0.9 * tms.get("derefs_commit", 0.0))
                power_series = [(round((t - t1) * 1e3, 1), None if x[0] is None else round(x[0] / 1e6), None if x[1] is None else round(x[1] / 1e6))
                                for t, x in zip(psamp.times, psamp.samples)]
                st = cx.prof_read()
                # The card's hwmon power (and clock) readings are running averages over roughly a second: inside one 0.3 s proof they
                # lag (the series above is kept as evidence).  So the row-commitment kernel's OPERATING POINT is measured on a loop:
                # the production kernel and window table over a polynomial of 2^24 uniformly random full-width scalars (the shape of
                # the derefs polynomial's regular rows), committed again and again for ~3 s; clock and power = medians over the
                # last 30 % of the loop, rate = counted table additions / HIP-event time of the same launches.
                msm_steady = None
                profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
                if not os.environ.get("VPIN_BENCH_NO_MSM_STEADY") and not profiled:   # (not inside a kernel trace: 4 s of one kernel would own its statistics)
                    rng = np.random.default_rng(7)
                    nz = 1 << 24
                    zr = rng.integers(0, 2**64, size=(nz, 4), dtype=np.uint64)
                    zr[:, 3] &= np.uint64((1 << 60) - 1)   # below 2^252 < q: every row is a canonical Montgomery image
                    tz = cx.upload(zr)
                    del zr
                    cx.dense_mlpoly_commit_sum(tz, SEED_C)   # warm
                    cx.prof_reset()
                    ps2 = PowerSampler(local_rank, interval_s=0.01)
                    ps2.start()
                    t_l = time.perf_counter()
                    n_loop = 0
                    while time.perf_counter() - t_l < 3.0:
                        cx.dense_mlpoly_commit_sum(tz, SEED_C)
                        n_loop += 1
                    t_e = time.perf_counter()
                    steady = ps2.stop(t_l + 0.7 * (t_e - t_l), t_e)   # the card's power reading is a running average over a second or more
                    sm = cx.prof_read().get("msm_rows")
                    tz.free()
                    if sm and sm["ms"] > 0 and steady:
                        msm_steady = dict(steady, G_adds_s=sm["units"] / (sm["ms"] * 1e-3) / 1e9, commitments=n_loop, scalars=nz,
                                          ms_per_commitment=sm["ms"] / max(1, sm["launches"]),
                                          what="msm_rows_kernel over 2^24 random full-width scalars (4096 rows x 4096), the production "
                                               "window table, looped for 3 s; medians over the last 30 %")
                cx.prof_enable(False)
                if len(lanes) > 1:
                    cx.set_shared_device(l0_shared)
                if cu_split and cu_split["largest_instance_masked"].startswith("after"):
                    cx.set_cumask_after_phase1(big_mask)
                sec = []
                cus, clk = cx.device_props()  # compute units, shader clock in Hz
                # Reference rates of the VALU-bound kernels.  (1) A static one: one wave-instruction per SIMD per 4 cycles over the
                # VALU instructions of the kernel's hot loop (profiles/r03_isa_counts.json); the SQ counters show the chip issues
                # slightly more than that on simple instructions and about half of it on v_mad_u64_u32
                # (profiles/r03_pmc_valu.json), so it is a yardstick, not a bound.  (2) A measured one for the MSM: the same point
                # addition on a register-resident dependent chain at the kernel's occupancy, no table loads, no digit logic
                # (tools/ubench_fpmul -> profiles/r03_ubench_fpmul.txt, its JSON line).
                isa, chain = {}, {}
                if isa_path:
                    with open(isa_path) as f:
                        isa = json.load(f)
                try:
                    with open(os.path.join(ROOT, "profiles", "r03_ubench_fpmul.txt")) as f:
                        for ln in f:
                            if ln.startswith("JSON "):
                                chain = json.loads(ln[5:])
                except OSError:
                    pass
                # (3) round 4: what the chip sustains under each instruction mix -- sclk and socket power sampled from the card's hwmon
                # files while the loop runs (tools/ubench_msm_variants -> profiles/r04_ubench_msm_variants.txt)
                power = {}
                try:
                    with open(os.path.join(ROOT, "profiles", "r04_ubench_msm_variants.txt")) as f:
                        for ln in f:
                            if ln.startswith("JSON "):
                                uj = json.loads(ln[5:])
                                pick = lambda k: {kk: uj[k][kk] for kk in ("G_per_s", "sclk_mhz", "watts")} if k in uj else None
                                power = {"point_addition_on_registers": pick("point addition, extended + affine entry (ref)"),
                                         "with_the_table_walk_gathers_shipped_layout": pick("walk [w][j][d], 96 B, prefetch 1 (shipped layout)"),
                                         "row_per_lane_walk": pick("walk ROW PER LANE (same (w,j) chip-wide), 96 B"),
                                         "source": "profiles/r04_ubench_msm_variants.txt (replayed; measured by tools/ubench_msm_variants, not in this run)"}
                except (OSError, ValueError):
                    pass
                m = st.get("msm_rows")
                if m and m["ms"] > 0 and m["units"] > 0:
                    ipa = isa.get("msm_rows_kernel", {}).get("valu_per_table_add", 1850)
                    peak_adds = cus * 4 * 64 * clk / (4.0 * ipa)
                    adds_s = m["units"] / (m["ms"] * 1e-3)
                    pm = pmc_l5.get("msm_rows_kernel (>= 1 GB fetched)", {})
                    sec.append({"kernel": "msm_rows_kernel (row commitments of >= 128 rows: witness, derefs, SNARK::encode shapes)",
                                "bound": "valu-issue", "limiter": "socket power: the walk's 96-byte gathers out of a 71 GB table push the card to its "
                                "~1.39 kW cap and the clock to ~1.8 GHz; the same additions without gathers run at 2.39 GHz (power)",
                                "power": power,
                                "achieved": adds_s / 1e9, "peak": peak_adds / 1e9, "unit": "G table adds/s",
                                "frac": adds_s / peak_adds, "launches": m["launches"], "ms": round(m["ms"], 3),
                                "table_adds": m["units"], "valu_instructions_per_add": ipa,
                                "frac_of_measured_chain": (adds_s / 1e9 / chain["point_adds_Gps_10x25"]) if chain.get("point_adds_Gps_10x25") else None,
                                "measured_chain_G_adds_s": chain.get("point_adds_Gps_10x25"),
                                "chain_note": "tools/ubench_fpmul: the kernel's point addition (ten-limb form, entry unpacked per addition) on a "
                                              "dependent register-resident chain, 12 waves per CU, no loads: what the arithmetic alone allows",
                                "peak_note": f"{cus} CUs x 4 SIMDs x 64 lanes x {clk / 1e9:.2f} GHz / (4 cycles per wave-instruction x {ipa} "
                                             "VALU instructions per affine table addition)",
                                "scalars_GBps": m["alg_bytes"] / (m["ms"] * 1e-3) / 1e9,
                                "traffic_bytes_per_add": pm.get("bytes_per_table_add"),
                                "traffic_note": "PMC FETCH_SIZE + WRITE_SIZE of the largest instance's row commitments / their table additions "
                                                "(profiles/r0x_pmc_traffic.json of the newest round, bench_L5_mult)"})
                p = st.get("spark_round_big")
                if p and p["ms"] > 0:
                    ach = p["alg_bytes"] / (p["ms"] * 1e-3) / 1e9
                    pp = pmc_l5.get("prod_round_kernel<*, true, true> (>= 2^20 pairs)", {})
                    ipp = isa.get("prod_round_kernel<true, true>", {}).get("valu_per_pair", 2224)
                    peak_pairs = cus * 4 * 64 * clk / (4.0 * ipp)
                    pairs_s = p["units"] / (p["ms"] * 1e-3)
                    sec.append({"kernel": "prod_round_kernel<*, true>, launches with >= 2^20 pairs per circuit (12 or 4 circuits per launch)",
                                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                                "valu_issue": {"achieved": pairs_s / 1e9, "peak": peak_pairs / 1e9, "unit": "G pair evaluations/s",
                                               "frac": pairs_s / peak_pairs, "valu_instructions_per_pair": ipp,
                                               "note": "the limiter: 8 products mod q per pair; same ceiling model as the MSM entry"},
                                "launches": p["launches"], "ms": round(p["ms"], 3), "alg_bytes_per_launch": p["alg_bytes"] / p["launches"],
                                "alg_note": "per launch: circuits x 2 tables x 32 B x 1.5 x len + the shared eq table (32 B x 1.5 x len)",
                                "traffic": pp.get("hbm_bytes_per_launch"),
                                "traffic_note": "PMC bytes per launch of the same launches (kernel names prod_round_kernel<*, true, true>), "
                                                "profiles/r02_pmc_traffic.json section bench_L5_mult"})
                # the same as scalars (the driver's record keeps scalar fields of `roofline`, not the nested list)
                rf = line["roofline"]
                rf["largest_instance_alone_ms"] = round(alone_ms, 2)
                for e in sec:
                    if e["kernel"].startswith("msm_rows_kernel"):
                        rf["msm_G_adds_s"] = e["achieved"]
                        rf["msm_frac_of_static_valu_peak"] = e["frac"]
                        rf["msm_frac_of_chain"] = e.get("frac_of_measured_chain")
                        rf["msm_ms_per_proof"] = e["ms"]
                        if msm_steady:
                            rf["msm_watts"] = msm_steady.get("watts_median")
                            rf["msm_sclk_mhz"] = msm_steady.get("sclk_mhz_median")
                            rf["msm_steady_G_adds_s"] = msm_steady.get("G_adds_s")
                            e["steady_state_measured_in_this_run"] = msm_steady
                        if msm_power:
                            e["power_measured_in_this_run"] = dict(msm_power, window="the middle 80 % of the derefs commitment of the "
                                                                   "largest instance proven alone (hwmon sampled every 4 ms)",
                                                                   series_ms_mhz_watts=power_series,
                                                                   phases_ms={"sat_until": round(tms.get("sat", 0.0) * 1e3, 1),
                                                                              "derefs_commit_until": round((tms.get("sat", 0.0) + tms.get("derefs_commit", 0.0)) * 1e3, 1)})
                    elif e["kernel"].startswith("prod_round_kernel"):
                        rf["prod_round_frac"] = e["frac"]
                        rf["prod_round_valu_frac"] = e["valu_issue"]["frac"]
                        rf["prod_round_ms_per_proof"] = e["ms"]
                line["roofline"]["secondary"] = sec
                line["roofline"]["secondary_scope"] = (f"{big} proven alone after the timed region ({alone_ms:.1f} ms, one stream, HIP events per launch, "
                                                       "table additions counted by vpin_prof_enable level 2)")
                line["kernels_largest_instance_alone"] = {name: {"launches": v["launches"], "ms": round(v["ms"], 4)} for name, v in st.items()}
        except Exception as e:  # noqa: BLE001 -- the measured line must come out; the failure is recorded in it
            errors['roofline_secondary'] = repr(e)[:300]
            print(f"[bench] roofline_secondary failed: {e!r}", file=sys.stderr, flush=True)
        for name in list(dev_insts):
            decomms.pop(name).free()
            dev_insts.pop(name).free()
        try:
            if not args.no_span:
                if os.environ.get("VPIN_POOL_TRACE"):
                    print("[bench] ==== reference-span pass starts ====", file=sys.stderr, flush=True)
                for cx in ctxs:
                    cx.set_shared_device(False)  # one proof at a time from here on
                if cu_split and cu_split["largest_instance_masked"].startswith("after"):
                    ctxs[0].set_cumask_after_phase1(None)
                for cx in ctxs:
                    # encode + prove per instance from here on, as a one-shot process does: SNARK::encode's 16N-scalar temporaries stay in
                    # the context's pool for the proof that follows instead of going back to the driver (a 17 GB hipMalloc per instance
                    # doubled this pass's time)
                    cx.set_expected_proofs(1)
                # one after the other = on the whole chip: the first lane's context (the other lanes' may be confined to their CUs)
                span_ctx = ctxs[0] if cu_split else None
                span, dead_commit, dead_digest_bytes, span_phases = {}, {}, {}, {}
                digest_prep_s = 0.0
                ts = time.perf_counter()
                for li, names in enumerate(lane_names):
                    for name in names:
                        t1 = time.perf_counter()
                        kind, inp = inputs_of[name]
                        cxs = span_ctx or ctxs[li]
                        g = build_instance(cxs, kind, inp)
                        t1a = time.perf_counter()
                        assert g.is_sat()
                        t1b = time.perf_counter()
                        r = g.snark_prove(SEED_C, SEED_P)
                        span[name] = round((time.perf_counter() - t1) * 1e3, 2)
                        stm = cxs.spark_timings()
                        span_phases[name] = {"gadget_witness_instance": round((t1a - t1) * 1e3, 2), "is_sat": round((t1b - t1a) * 1e3, 2),
                                             "encode_and_prove": round(span[name] - (t1b - t1) * 1e3, 2),
                                             **{"lib_" + kk: round(vv * 1e3, 2) for kk, vv in stm.items() if kk != "_"}}
                        if os.environ.get("VPIN_POOL_TRACE"):
                            print(f"[bench] span {name}: {span[name]} ms", file=sys.stderr, flush=True)
                        assert r["proof"] == last_proof[name]["proof"], f"{name}: reference-span proof differs from the timed region's"
                        # what the reference computes inside this span and never uses, timed beside it --
                        # (i) the third commitment my_dense_mlpoly_commit (proof_point_mult.rs:58-59; only row 0 is read, by an assert)
                        t2 = time.perf_counter()
                        third = cxs.dense_mlpoly_commit_sum(g.vars, SEED_C)
                        dead_commit[name] = round((time.perf_counter() - t2) * 1e3, 2)
                        assert bytes(third[0]) == bytes(cxs.points_add(r["comm_para"][:1], r["comm_input"][:1])[0]), name  # :69-73
                        # (ii) Instance::new's digest: zlib over bincode(A, B, C) (lib.rs:232-243, r1csinstance.rs:154-158), 48 B per
                        # entry, consumed by the NIZK path only.  Host work on one core per instance: the buffer is built here (the
                        # triplets come back from the device) and compressed at the end of the run, every instance at FULL size
                        # (round 5: 3.7 GB for L5-mult; round 4 priced it per byte from a small instance), in a thread of its own
                        # beside the PMC child passes.
                        t3 = time.perf_counter()
                        dead_digest_bytes[name] = 48 * sum(g.nnz) + 3 * 8 + 3 * (2 * 8 + 8)
                        if args.digest and rank == 0:   # (one rank measures it: the figure is per trace, not per rank)
                            buf = bytearray(np.array([g.num_cons, g.num_vars, g.num_inputs], dtype="<u8").tobytes())
                            for m in range(3):
                                row, col, val = g.triplets(m)
                                ent = np.zeros(len(row), dtype=[("row", "<u8"), ("col", "<u8"), ("val", "<u8", (4,))])
                                ent["row"], ent["col"], ent["val"] = row, col, val
                                buf += np.array([g.num_cons.bit_length() - 1, (2 * g.num_vars).bit_length() - 1, len(row)], dtype="<u8").tobytes()
                                buf += ent.tobytes()
                                del ent, row, col, val
                            assert len(buf) == dead_digest_bytes[name], (len(buf), dead_digest_bytes[name])
                            digest_jobs.append((name, buf))
                        digest_prep_s += time.perf_counter() - t3
                        g.free()
                span_s = time.perf_counter() - ts - sum(dead_commit.values()) / 1e3 - digest_prep_s
                line["value_reference_span"] = total_cons_step / span_s  # the reference's own timed span, see reference_span.scope
                line["reference_span"] = {
                    "ms_per_trace": round(span_s * 1e3, 1), "constraints_per_s": total_cons_step / span_s, "ms": span, "phases_ms": span_phases,
                    "scope": "per instance, serially: witness inputs in host memory -> gadget + witness synthesis + Instance::new "
                             "(device) -> is_sat -> SNARK::encode -> my_lib_prove -> proof bytes on the host; generator tables warm",
                }
                # the same span WITH the work the reference does inside it and never uses; the digest figures
                # are filled in when the compressions have finished (finish_dead_work below)
                line["reference_span"]["dead_work"] = {
                    "third_commitment_ms": dead_commit, "third_commitment_ms_total": round(sum(dead_commit.values()), 1),
                    "digest_bytes": dead_digest_bytes, "triplets_to_host_and_bincode_s": round(digest_prep_s, 2),
                    "ms_per_trace_without": round(span_s * 1e3, 1),
                    "note": "inside the reference's 'Proof generation time' and dropped by this build because no proof byte depends on it: "
                            "(i) my_dense_mlpoly_commit of the whole assignment (proof_point_mult.rs:58-59; its row 0 feeds an assert, "
                            "reproduced here) -- measured per instance on the device (vpin_dense_mlpoly_commit_sum); the poly_prime loop of "
                            ":61-67 is a vector addition nobody reads; (ii) the zlib digest of bincode(A, B, C) in Instance::new "
                            "(lib.rs:232-243) -- one host core per instance, MEASURED on every instance at full size (zlib level 6 = flate2's "
                            "default; digest_ms); is_sat (the other N4 item) is INSIDE the span on both sides",
                }
                if len(lanes) > 1:
                    # the same span with the trace's instances on the bench's lanes (streams) instead of one after the other
                    for li, cx in enumerate(ctxs):
                        cx.set_shared_device(l0_shared if li == 0 else small_shared)
                    if cu_split and cu_split["largest_instance_masked"].startswith("after"):
                        ctxs[0].set_cumask_after_phase1(big_mask)
                    errs = []

                    def span_lane(li):
                        try:
                            for name in lane_names[li]:
                                kind, inp = inputs_of[name]
                                g = build_instance(ctxs[li], kind, inp)
                                assert g.is_sat()
                                r = g.snark_prove(SEED_C, SEED_P)
                                assert r["proof"] == last_proof[name]["proof"], name
                                g.free()
                        except Exception as e:  # noqa: BLE001
                            errs.append(repr(e))
                    lanes_all = []
                    for _rep in range(max(1, args.span_lanes_repeats)):
                        ts = time.perf_counter()
                        th = [threading.Thread(target=span_lane, args=(li,)) for li in range(len(lanes))]
                        for t in th:
                            t.start()
                        for t in th:
                            t.join()
                        lanes_all.append(round((time.perf_counter() - ts) * 1e3, 1))
                        if os.environ.get("VPIN_POOL_TRACE"):
                            print(f"[bench] lanes pass {_rep}: {lanes_all[-1]} ms, free {mem_get_info()[0] / 2**30:.1f} GiB", file=sys.stderr, flush=True)
                        assert not errs, (errs, lanes_all)
                    lanes_s = lanes_all[0] / 1e3
                    line["reference_span"]["lanes"] = {
                        "ms_per_trace": round(lanes_s * 1e3, 1), "constraints_per_s": total_cons_step / lanes_s, "ms_every_pass": lanes_all,
                        "scope": f"the same per-instance span with the trace's instances on {len(lanes)} streams / host threads at once"}

        except Exception as e:  # noqa: BLE001 -- the measured line must come out; the failure is recorded in it
            errors['reference_span'] = repr(e)[:300]
            print(f"[bench] reference_span failed: {e!r}", file=sys.stderr, flush=True)
    # ---- CPU baseline: the oracle on a bounded sample, rank 0, N=1 only ----
    try:
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            import oracle_lib as O
            all_cores = min(os.cpu_count() or 1, 64)
            lab = "L1" if trace == "lenet" else trace
            sm = G.synthetic_mult_instance(lab, args.cpu_sample_mult) if G.CONFIGS[lab]["n_mult"] else None
            sa = G.synthetic_add_instance(lab, args.cpu_sample_add)
            sample_cons = (sm.num_cons_unpadded if sm else 0) + sa.num_cons_unpadded
            dm, da = (sm.as_dict() if sm else None), sa.as_dict()

            def cpu_run(threads, dm=dm, da=da):
                """the C oracle (restated reference prover) on the sample; SNARK::encode excluded on both sides"""
                os.environ["OMP_NUM_THREADS"] = str(threads)
                t0 = time.perf_counter()
                tm, enc_s = {}, 0.0
                prove = (lambda dd: O.snark_prove(dd, SEED_C, SEED_P, threads=threads)) if args.snark else \
                        (lambda dd: O.sat_prove(dd, SEED_C, SEED_P, threads=threads))
                if dm is not None:
                    assert len(prove(dm)["proof"])
                    tm = dict(O.sat_timings(), **({"spark_" + k: v for k, v in O.spark_timings().items()} if args.snark else {}))
                    enc_s += O.spark_timings()["encode"] if args.snark else 0.0
                assert len(prove(da)["proof"])
                enc_s += O.spark_timings()["encode"] if args.snark else 0.0
                return time.perf_counter() - t0 - enc_s, tm

            s_all, tm_all = cpu_run(all_cores)
            s_one, tm_one = (cpu_run(1) if args.cpu_single_thread else (None, {}))
            n_m, n_a = (args.cpu_sample_mult if sm else 0), sa.num_cons_unpadded // 10
            sample_short = (f"{n_m} point-mults + {n_a} point-adds drawn like layer {lab} ({sample_cons} constraints), C oracle, whole SNARKs, "
                            f"{s_all:.1f} s on {all_cores} threads")
            sample_txt = (f"{n_m} point-mults + {n_a} point-adds drawn like layer {lab} ({sample_cons} constraints), C oracle (restated "
                          f"reference prover): OpenMP over the commitment rows (as rayon in the reference), single-threaded sum-checks; "
                          f"{s_all:.1f} s on {all_cores} threads" + (f", {s_one:.1f} s on 1" if s_one else ""))
            value, full = sample_cons / s_all, None
            seconds = s_all
            flab = args.cpu_full_label
            if flab != "none" and args.snark:
                # a BASELINE configuration at full size beside it: CNN A's whole trace (configs[1]: 178 point-mults
                # = 616,592 constraints, 2^20 padded, + 2144 point-adds) on all host threads -- ~20 s of the box's cores
                fm = G.synthetic_mult_instance(flab) if G.CONFIGS[flab]["n_mult"] else None
                fa = G.synthetic_add_instance(flab)
                full_cons = (fm.num_cons_unpadded if fm else 0) + fa.num_cons_unpadded
                s_full, tm_full = cpu_run(all_cores, fm.as_dict() if fm else None, fa.as_dict())
                if fm:
                    fm.free()
                fa.free()
                value, seconds = full_cons / s_full, s_full
                full = {"label": flab, "constraints": full_cons, "seconds": round(s_full, 2), "cores": all_cores, "value": value,
                        "how": "measured in this run", "spans_ms_mult": {kk: round(vv * 1e3, 1) for kk, vv in tm_full.items()}}
                sample_short = (f"CNN {flab}'s whole trace at full size ({full_cons} constraints), C oracle, whole SNARKs, {s_full:.1f} s on "
                                f"{all_cores} threads")
                sample_txt = sample_short + "; and a bounded sample: " + sample_txt
            else:
                # not measured in this run: the full-size figure of the newest committed profile that holds one, labelled so
                import glob
                for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default*.json")), reverse=True):
                    try:
                        with open(pth) as f:
                            fc = (json.load(f).get("cpu_baseline") or {}).get("full_config")
                    except (OSError, ValueError):
                        fc = None
                    if fc and fc.get("how", "measured in this run") == "measured in this run":
                        full = dict(fc, how=f"replayed from profiles/{os.path.basename(pth)} (NOT measured in this run)")
                        break
            line["cpu_baseline"] = {
                "value": value, "unit": "constraints/s", "cores": all_cores, "kind": "port", "seconds": round(seconds, 2),
                "sample": sample_txt, "sample_short": sample_short,
                "full_config": full,
                "small_sample": {"value": sample_cons / s_all, "cores": all_cores, "seconds": round(s_all, 2), "constraints": sample_cons},
                "third_point": "the 2^25-constraint instance (L5-mult, 20,784,000 constraints) through the same oracle on 16 host threads: "
                               "profiles/r03_l5full_oracle_host.log (SNARK::encode + prove 644 s = 32 k constraints/s, 94 GB)",
                "spans_ms_mult": {kk: round(vv * 1e3, 1) for kk, vv in tm_all.items()},
            }
            if s_one:
                line["cpu_baseline"]["single_thread"] = {
                    "value": sample_cons / s_one, "cores": 1, "seconds": round(s_one, 2),
                    "spans_ms_mult": {kk: round(vv * 1e3, 1) for kk, vv in tm_one.items()},
                    "note": "beside the single-core profile of Spartan/README.md:338-377 (2^20 constraints: SNARK::prove 39.1 s = 26.8 k "
                            "constraints/s on one i7-1065G7 core) mind the shape: that instance has one non-zero entry per constraint and "
                            "matrix, vPIN's point-mult gadget 1.5 / 1.3 / 0.9 in A / B / C, so SPARK runs over 2.4 x the unpadded constraints"}

    except Exception as e:  # noqa: BLE001 -- the measured line must come out; the failure is recorded in it
        errors['cpu_baseline'] = repr(e)[:300]
        print(f"[bench] cpu_baseline failed: {e!r}", file=sys.stderr, flush=True)
    # ---- N > 1: the same trace ONCE over all ranks, beside the weak line ----
    # The default line shards independent traces (one per rank, no data-path collective).  BASELINE.json's configs[4] is the
    # other question -- one LeNet trace over the node, the 2^25 instance proven by all ranks together -- and the driver only
    # ever runs the default command, so the strong schedule of --scaling strong runs here as well and lands in `strong`.
    if world > 1 and args.snark and not args.no_strong and not args.host_buffers:
        for name in list(dev_insts):  # every rank's resident trace: the cooperative proofs build their own
            decomms.pop(name).free()
            dev_insts.pop(name).free()
        for cx in ctxs:
            cx.close()
        ctxs = []
        os.environ.setdefault("VPIN_COMM_TIMEOUT_S", "60")  # a rank that fails must not hold the others for long
        # RCCL at world > 1 has never run on this code (no multi-GPU hardware during the build): should its initialisation or a
        # collective hang, the weak line above must still come out.  Every rank arms a watchdog; when it fires rank 0 prints the
        # line with the failure recorded and every rank leaves at once (no rank can then be waited for).
        phase = {"name": "staged"}

        def bail():
            if rank == 0:
                msg = (f"timed out after {args.strong_timeout} s in the '{phase['name']}' pass: a rank did not return (RCCL initialisation "
                       "or a collective); the weak numbers above are unaffected")
                if "strong" in line:
                    line["strong"]["rccl_pass"] = {"error": msg}   # the staged pass had already delivered
                else:
                    line["strong"] = {"error": msg}
                emit(line, detail_path)
            os._exit(0)
        watchdog = threading.Timer(args.strong_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        if rank == 0:
            # ... and should the first outing of RCCL take the process down instead (a fault inside the library, the launcher's
            # SIGTERM after another rank died), the weak line is left with a signal handler that writes it out (vpin_crash_line_set)
            from bench_common import compact_line, dumps_line
            try:
                safe = dict(line, strong_error="the strong sub-record ended the process (fatal signal, or the launcher's SIGTERM after "
                            "another rank died); the weak numbers of this line are unaffected", run_s=time.perf_counter() - T_START)
                vpin_amd.Context.crash_line_set((dumps_line(compact_line(safe, None)) + "\n").encode())
            except (ValueError, vpin_amd.VpinError):
                pass
        ndev = max(1, torch.cuda.device_count())  # (world > 1: torch is imported)
        nsteps = max(1, min(args.steps, 5))

        def one_pass(with_rccl):
            err, rec = None, None
            try:
                from bench_strong import _strong_core
                rec = _strong_core(args, grp, rank, world, local_rank % ndev, with_rccl, ndev, nsteps, 1)
            except Exception as e:  # noqa: BLE001 -- reported in the line, the weak numbers above stand
                err = repr(e)
            errs = grp.gather_objects(err)
            if rank == 0 and not any(errs):
                rec["speedup_vs_one_gpu_step_of_this_run"] = line["ms_per_step"] / rec["ms_per_step"]
                rec["speedup_note"] = ("a rank's own trace on its own GPU (the weak line's ms_per_step: the four-lane step, what N = 1 "
                                       "delivers) over the time ONE trace takes on all ranks together")
            return rec, [e for e in errs if e]

        # Pass 1: device vectors staged through the shared-memory transport -- the path that has run (two and four processes on
        # one GPU).  Pass 2: the same with RCCL carrying them (ncclAllGather), which no hardware has run yet: if it fails or hangs,
        # pass 1's record stands.
        rec, errs = one_pass(False)
        if rank == 0:
            line["strong"] = rec if not errs else {"error": errs}
        if args.backend == "nccl" and ndev >= world and not errs:
            phase["name"] = "rccl"
            rec2, errs2 = one_pass(True)
            if rank == 0:
                line["strong"]["rccl_pass"] = ({"error": errs2} if errs2 else
                                               {k: rec2[k] for k in ("value", "ms_per_step", "rccl", "bytes_equal_oracle_digest", "comm",
                                                                     "speedup_vs_one_gpu_step_of_this_run")})
        watchdog.cancel()
        if rank == 0:
            vpin_amd.Context.crash_line_set(b"")
            # the same as scalars (the driver's record keeps scalar fields)
            st_rec = line.get("strong") or {}
            rp = st_rec.get("rccl_pass") or {}
            line["strong_ms_per_step"] = st_rec.get("ms_per_step")
            line["strong_speedup_vs_one_gpu_step"] = st_rec.get("speedup_vs_one_gpu_step_of_this_run")
            line["strong_bytes_ok"] = (all(st_rec.get("bytes_equal_oracle_digest", {}).values()) if st_rec.get("bytes_equal_oracle_digest") else None)
            line["strong_rccl_ms_per_step"] = rp.get("ms_per_step")
            line["strong_rccl_world"] = max(rp.get("rccl", {}).get("world_by_group", {}).values(), default=0) if rp.get("rccl") else 0
            line["strong_rccl_bytes_ok"] = (all(rp.get("bytes_equal_oracle_digest", {}).values()) if rp.get("bytes_equal_oracle_digest") else None)
            line["strong_error"] = st_rec.get("error") or rp.get("error")

    # ---- reference_span.dead_work: the zlib digests (one thread per instance, started here so that they run beside the PMC
    # child passes below: zlib releases the GIL) ----
    digest_ms, digest_threads = {}, []
    if digest_jobs:
        import zlib

        def compress(name, buf):
            t_z = time.perf_counter()
            z = zlib.compress(buf, 6)  # flate2's default level
            digest_ms[name] = (round((time.perf_counter() - t_z) * 1e3, 1), len(z))
        digest_threads = [threading.Thread(target=compress, args=job) for job in digest_jobs]
        for t in digest_threads:
            t.start()

    # ---- roofline.traffic and the VALU-issue occupancy measured in THIS run (replayed figures, a ceiling the
    # kernel exceeded) ----
    # Last GPU work before the line is printed: every context is closed and the shared window tables are released, so the three
    # rocprofv3 --pmc children have the GPU to themselves; this process does not touch the GPU afterwards.
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    try:
        if (rank == 0 and world == 1 and "roofline" in line and trace == "lenet" and not args.no_live_pmc and not args.no_roofline_pass
                and args.pmc_traffic is None and not args.only and not args.serial and not under_profiler):  # (no profiler inside a profiler)
            ncu_dev = ctxs[0].device_props()[0] if ctxs else 256
            for cx in ctxs:
                cx.close()
            ctxs = []
            vpin_amd.lib().vpin_gens_shared_clear()
            t_p = time.perf_counter()
            live, info = live_pmc_traffic(cus=ncu_dev)
            if live is not None:
                line["roofline"]["traffic_replayed"] = {"value": line["roofline"].get("traffic"), "source": line["roofline"].get("traffic_source")}
                line["roofline"]["traffic"] = live
                line["roofline"]["traffic_source"] = ("measured in this run: child processes under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                                                      "(separate passes, no tracing beside them) proving the largest instance alone; "
                                                      "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the kernel's dispatches "
                                                      f"({info['FETCH_SIZE']['dispatches']}); {time.perf_counter() - t_p:.0f} s with the SQ pass")
                line["roofline"]["traffic_counters"] = {k: v for k, v in info.items() if k != "VALU"}
                line["roofline"]["traffic_over_algorithmic"] = live / line["roofline"]["alg_bytes_per_launch"]
                v = info.get("VALU", {})
                if "valu_issue_frac" in v:
                    line["roofline"]["limiter_frac"] = v["valu_issue_frac"]
                    line["roofline"]["limiter_frac_source"] = "measured in this run (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE, third child pass)"
                    line["roofline"]["limiter_counters"] = v
                    pairs = line["roofline"]["alg_bytes_per_launch"] / 384.0 * v["dispatches"]
                    line["roofline"]["measured_valu_instructions_per_pair"] = v["SQ_INSTS_VALU"] * 64.0 / pairs
                elif v:
                    line["roofline"]["limiter_live_error"] = v
            else:
                line["roofline"]["traffic_live_error"] = info
    except Exception as e:  # noqa: BLE001 -- the measured line must come out; the failure is recorded in it
        errors['live_pmc'] = repr(e)[:300]
        print(f"[bench] live_pmc failed: {e!r}", file=sys.stderr, flush=True)
    if "roofline" in line and line["roofline"].get("limiter_frac") is None:
        # not measured in this run: the newest profile that holds the kernel's SQ counters (tools/pmc_valu.py), else the static model
        pth, nm = newest_profile("pmc_valu.json")
        ent = None
        if pth:
            with open(pth) as f:
                ent = json.load(f).get("kernels", {}).get("sc_cubic3_kernel<true, true>")
        if ent:
            line["roofline"]["limiter_frac"] = ent["valu_issue_frac"]
            line["roofline"]["limiter_frac_source"] = f"replayed from profiles/{nm} (NOT measured in this run)"
        else:
            line["roofline"]["limiter_frac"] = line["roofline"].get("limiter_frac_static_model")
            line["roofline"]["limiter_frac_source"] = "the static model (no SQ counters of this kernel at hand): a yardstick, not a bound"

    for t in digest_threads:
        t.join()
    if digest_ms and "reference_span" in line:
        dw = line["reference_span"]["dead_work"]
        dw["digest_ms"] = {n: v[0] for n, v in digest_ms.items()}
        dw["digest_compressed_bytes"] = {n: v[1] for n, v in digest_ms.items()}
        dw["digest_ms_total"] = round(sum(v[0] for v in digest_ms.values()), 1)
        big_n = max(digest_ms, key=lambda n: digest_ms[n][0])
        dw["digest_MB_per_s_largest"] = dw["digest_bytes"][big_n] / (digest_ms[big_n][0] * 1e-3) / 1e6
        dead_s = dw["third_commitment_ms_total"] / 1e3 + dw["digest_ms_total"] / 1e3
        span_s = dw["ms_per_trace_without"] / 1e3
        dw["ms_per_trace_with"] = round((span_s + dead_s) * 1e3, 1)
        dw["constraints_per_s_with"] = total_cons_step / (span_s + dead_s)
        dw["digest_how"] = "measured"
        dw["digest_note"] = ("every instance's bincode(A, B, C) compressed at full size, one Python thread (one host core) per instance, "
                             "run beside the PMC child passes; the sum of the per-instance times is what the reference's serial span pays")
    elif "reference_span" in line and trace == "lenet" and not args.only and not args.skip:
        # the digest was not measured in this run (--digest): the newest committed profile that measured it, labelled so
        import glob
        dw = line["reference_span"]["dead_work"]
        for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default*.json")), reverse=True):
            try:
                with open(pth) as f:
                    old = ((json.load(f).get("reference_span") or {}).get("dead_work") or {})
            except (OSError, ValueError):
                old = {}
            if old.get("digest_ms_total") and old.get("digest_how", "measured") == "measured":
                dw["digest_ms_total"] = old["digest_ms_total"]
                dw["digest_how"] = f"replayed:{os.path.basename(pth)}"
                dead_s = dw["third_commitment_ms_total"] / 1e3 + old["digest_ms_total"] / 1e3
                dw["ms_per_trace_with"] = round(dw["ms_per_trace_without"] + dead_s * 1e3, 1)
                dw["constraints_per_s_with"] = total_cons_step / (dw["ms_per_trace_without"] / 1e3 + dead_s)
                break

    # ---- guard (the span doubled late in a round and nobody looked): the reference span of this run against the
    # newest committed default line's; more than 1.3 x it -> a warning field in the printed line ----
    if "reference_span" in line and trace == "lenet" and not args.only and not args.skip:
        import glob
        for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default.json")), reverse=True):
            try:
                with open(pth) as f:
                    old = json.load(f)
                old_ms = old.get("reference_span_ms") or (old.get("reference_span") or {}).get("ms_per_trace")
            except (OSError, ValueError):
                old_ms = None
            if old_ms:
                now_ms = line["reference_span"]["ms_per_trace"]
                line["reference_span"]["previous"] = {"file": os.path.basename(pth), "ms_per_trace": old_ms}
                if now_ms > 1.3 * old_ms:
                    line["span_warning"] = f"reference span {now_ms:.0f} ms > 1.3 x {old_ms:.0f} ms of profiles/{os.path.basename(pth)}"
                break

    if rank == 0:
        line["run_s"] = time.perf_counter() - T_START
        if errors:
            line["errors"] = errors
        emit(line, detail_path)
    for cx in ctxs:
        cx.close()
    grp.close()
    if "parity" in errors:
        sys.exit(1)


if __name__ == "__main__":
    main()
