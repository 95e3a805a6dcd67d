"""CPU checks of the product's host-side protocol glue (no GPU): its generator derivation,
Merlin transcript and small Pedersen commitments must agree with the public vectors and with
the oracle, which was written separately in C."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as O
import pymodel as M
import vpin_amd


def test_host_merlin_matches_public_vector(golden_dir):
    with open(os.path.join(golden_dir, "ristretto_kat.json")) as f:
        k = json.load(f)["merlin_equivalence_simple"]
    out = (C.c_uint8 * 32)()
    data = k["data"].encode()
    rc = vpin_amd.lib().vpin_host_merlin_kat(k["protocol"].encode(), k["label"].encode(),
                                             (C.c_uint8 * len(data))(*data), len(data),
                                             k["challenge_label"].encode(), out, 32)
    assert rc == 0 and bytes(out).hex() == k["challenge32"]


def test_host_generators_match_oracle_and_rfc():
    nb = 40
    out = np.zeros((nb, 128), dtype=np.uint8)
    assert vpin_amd.lib().vpin_host_gens_derive(b"gens_r1cs_sat", nb, out.ctypes.data_as(C.c_void_p)) == 0
    exp, og = O.gens_stream_xyzt(nb)
    L = O.lib()
    for i in range(nb):
        # projective coordinates may differ by scaling; compare the canonical compressed forms
        g = O.Ge()
        L.ge_from_xyzt(C.byref(g), out[i].ctypes.data_as(C.c_void_p))
        a, b = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        L.ge_compress(a, C.byref(g))
        L.ge_compress(b, C.byref(og[i]))
        assert bytes(a) == bytes(b)


def test_host_commit_matches_oracle():
    rng = np.random.default_rng(3)
    L = O.lib()
    for n in (1, 3, 4):
        vals = [int(rng.integers(0, 2**62)) ** 4 % M.Q for _ in range(n)]
        vals[0] = M.Q - 1
        blind = int(rng.integers(0, 2**62)) ** 4 % M.Q
        v, b = M.ints_to_table(vals), M.ints_to_table([blind])
        out = (C.c_uint8 * 32)()
        assert vpin_amd.lib().vpin_host_commit(b"test-label", v.ctypes.data_as(C.c_void_p), n,
                                               b.ctypes.data_as(C.c_void_p), out) == 0
        gens = (O.Ge * (n + 1))()
        L.oracle_gens_new(gens, n, (C.c_uint8 * 10)(*b"test-label"), 10)
        exp = O.Ge()
        L.oracle_commit(C.byref(exp), O.ptr(v), n, O.ptr(b), gens, C.byref(gens[n]))
        e = (C.c_uint8 * 32)()
        L.ge_compress(e, C.byref(exp))
        assert bytes(out) == bytes(e)


def test_mailbox_framing_rejects_torn_and_stale_pieces():
    """mailbox_dev.h: a scalar crosses PCIe as three 16-byte pieces {seq, 3 words}, the spare word holds a checksum; a piece
    torn 8 + 8, pieces of two publications and a stale sequence number must all be refused (and are then read again)"""
    assert vpin_amd.lib().vpin_host_mailbox_selftest() == 0


def test_host_scalar_mul2_matches_oracle():
    """host/curve.h: width-5 NAF scalar multiplication and the shared-doubling a*P + b*Q of the verifier, against the oracle's
    double-and-add on generator-stream points; edge scalars: 0, 1, q - 1, 2^252, runs of ones, a lone top window"""
    rng = np.random.default_rng(11)
    L = O.lib()
    L.ge_scalarmul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ge_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    gens = (O.Ge * 2)()
    L.oracle_gens_new(gens, 1, (C.c_uint8 * 9)(*b"mul2-test"), 9)
    Pc, Qc = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    L.ge_compress(Pc, C.byref(gens[0]))
    L.ge_compress(Qc, C.byref(gens[1]))
    edge = [0, 1, 2, 15, 16, 17, 31, 32, 33, M.Q - 1, M.Q - 2, 1 << 252, (1 << 252) - 1, (1 << 200) - 1, 0xAAAAAAAAAAAAAAAA << 100,
            (1 << 64) - 1, 1 << 64, (1 << 128) - (1 << 60), 31 << 59, 31 << 60, 31 << 61, 17 << 62]
    pairs = [(a, b) for a in edge for b in (0, 1, M.Q - 1)] + [(0, b) for b in edge]
    pairs += [(int.from_bytes(rng.bytes(40), "little") % M.Q, int.from_bytes(rng.bytes(40), "little") % M.Q) for _ in range(40)]
    for a, b in pairs:
        sa, sb = M.ints_to_table([a]), M.ints_to_table([b])
        out = (C.c_uint8 * 32)()
        assert vpin_amd.lib().vpin_host_scalar_mul2(sa.ctypes.data_as(C.c_void_p), Pc, sb.ctypes.data_as(C.c_void_p), Qc, out) == 0, (a, b)
        ea, eb, es = O.Ge(), O.Ge(), O.Ge()
        L.ge_scalarmul(C.byref(ea), O.ptr(sa), C.byref(gens[0]))
        L.ge_scalarmul(C.byref(eb), O.ptr(sb), C.byref(gens[1]))
        L.ge_add(C.byref(es), C.byref(ea), C.byref(eb))
        e = (C.c_uint8 * 32)()
        L.ge_compress(e, C.byref(es))
        assert bytes(out) == bytes(e), (hex(a), hex(b))
    bad = (C.c_uint8 * 32)(*([0xff] * 32))
    one = M.ints_to_table([1])
    assert vpin_amd.lib().vpin_host_scalar_mul2(one.ctypes.data_as(C.c_void_p), bad, one.ctypes.data_as(C.c_void_p), Qc, out) == -6


def test_gadget_shape_matches_the_built_instances():
    """vpin_gadget_shape (what vpin_prove sizes the generator sets of its second instance with before reading the witness)
    against the instances the host builders really produce"""
    from vpin_amd import gadgets as G
    for kind, n in (("mult", 1), ("mult", 3), ("add", 1), ("add", 6), ("add", 100)):
        inp = G.synthetic_mult_inputs("3_32", n) if kind == "mult" else G.synthetic_add_inputs("3_32", n)
        inst = G.point_mult(*inp) if kind == "mult" else G.point_add(*inp)
        d = inst.as_dict()
        nc, nv, nnz = vpin_amd.gadget_shape(kind, n)
        assert (nc, nv) == (d["num_cons"], d["num_vars"]), (kind, n)
        assert nnz == [len(d[m][0]) for m in ("A", "B", "C")], (kind, n)
        inst.free()


def test_bench_cpulist_parser_and_overlap_bound():
    """bench.py's NUMA helper parses sysfs cpulists; dist.replay_trace's overlap flag is a lower bound of the sequential order"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    import sys
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("64-127,192-255")[:2] == [64, 65] and len(bench.parse_cpulist("64-127,192-255")) == 128
    assert bench.pin_to_gpu_numa_node(0, 1, "off") == {"mode": "off"}
    from vpin_amd.dist import plan_trace, replay_trace
    cons = [20_784_000, 2_771_200, 1_039_200, 831_360, 581_952, 70_560, 24_000, 7_680, 4_060, 2_880, 1_860]
    coop, small, _ = plan_trace(cons, 4, 0.5 * 2**24, 0.5 * 2**22)
    coop_ms = {k: 100.0 for k in coop}
    single = [10.0 + c / 4e4 for c in cons]
    seq = replay_trace(coop, small, 4, coop_ms, single)
    ov = replay_trace(coop, small, 4, coop_ms, single, overlap=True)
    assert all(o <= s for o, s in zip(ov, seq)) and max(ov) >= 100.0


def test_crash_line_is_written_when_a_fatal_signal_ends_the_process():
    """vpin_crash_line_set: bench.py leaves its weak line here before the strong sub-record's first outing of RCCL; a SIGSEGV / SIGTERM
    then ends the process with that line on stdout and exit code 0"""
    import subprocess
    import sys
    code = ("import os, signal, sys\n"
            f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
            "import vpin_amd\n"
            "vpin_amd.Context.crash_line_set(b'{\"value\":1}\\n')\n"
            "sys.stdout.flush()\n"
            "os.kill(os.getpid(), getattr(signal, sys.argv[1]))\n"
            "import time; time.sleep(5)\n"
            "print('not reached')\n")
    for sig in ("SIGSEGV", "SIGTERM", "SIGABRT"):
        r = subprocess.run([sys.executable, "-c", code, sig], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout == '{"value":1}\n', (sig, r.returncode, r.stdout, r.stderr[-300:])
    # disarmed: the default disposition is back
    code2 = code.replace("sys.stdout.flush()", "vpin_amd.Context.crash_line_set(b'')\nsys.stdout.flush()")
    r = subprocess.run([sys.executable, "-c", code2, "SIGTERM"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and r.stdout == ""
