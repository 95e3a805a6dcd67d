"""GPU parity of the device hash-to-group (vpin_gens_map_stream: RistrettoPoint::from_uniform_bytes for a whole generator
set, Spartan/src/commitments.rs:20-38 + RFC 9496 4.3.4) against the oracle's generator stream and the oracle's map of
arbitrary 64-byte strings: the real streams of both labels, and edge inputs (zero halves, all ones, the top bit that
FieldElement::from_bytes ignores, both branches of SQRT_RATIO_M1)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
P = 2**255 - 19
BASEPOINT_COMPRESSED = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


def map_on_device(ctx, stream):
    return ctx.gens_map_stream(stream)


def compress_xyzt(rows):
    L = O.lib()
    L.ge_from_xyzt.argtypes = [C.c_void_p, C.c_void_p]
    enc = []
    for r in rows:
        g, e = O.Ge(), (C.c_uint8 * 32)()
        L.ge_from_xyzt(C.byref(g), r.ctypes.data_as(C.c_void_p))
        L.ge_compress(e, C.byref(g))
        enc.append(bytes(e))
    return enc


def oracle_map(stream):
    L = O.lib()
    L.ge_from_uniform_bytes.argtypes = [C.c_void_p, C.c_void_p]
    enc = []
    for i in range(len(stream) // 64):
        g, e = O.Ge(), (C.c_uint8 * 32)()
        L.ge_from_uniform_bytes(C.byref(g), (C.c_uint8 * 64)(*stream[64 * i:64 * i + 64]))
        L.ge_compress(e, C.byref(g))
        enc.append(bytes(e))
    return enc


@pytest.mark.parametrize("label,nb", [(b"gens_r1cs_sat", 1030), (b"gens_r1cs_eval", 4098)])
def test_generator_streams_match_the_oracle(ctx, label, nb):
    stream = hashlib.shake_256(label + BASEPOINT_COMPRESSED).digest(64 * nb)
    dev = map_on_device(ctx, stream)
    for x in dev.reshape(-1, 32):  # canonical integers: what Point::from_xyzt / the table builder read
        assert int.from_bytes(bytes(x), "little") < P
    got = compress_xyzt(dev)
    xyzt, og = O.gens_stream_xyzt(nb, label)
    L = O.lib()
    for i in range(nb):
        e = (C.c_uint8 * 32)()
        L.ge_compress(e, C.byref(og[i]))
        assert got[i] == bytes(e), i


def test_edge_inputs_match_the_oracle(ctx):
    rng = np.random.default_rng(5)
    halves = [bytes(32), b"\xff" * 32, b"\x01" + bytes(31), bytes(31) + b"\x80", b"\xed" + b"\xff" * 30 + b"\x7f",
              b"\xec" + b"\xff" * 30 + b"\x7f", (P + 5).to_bytes(32, "little"), (2**255 - 1).to_bytes(32, "little"), (2).to_bytes(32, "little")]
    halves += [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(23)]
    stream = b"".join(a + b for a in halves for b in halves[:8])
    assert compress_xyzt(map_on_device(ctx, stream)) == oracle_map(stream)


def test_concurrent_requests_for_one_table_build_it_once(ctx):
    """vpin_gens_shared: the registry is unlocked while a table is built; a second thread asking for the same (or a shorter)
    stream of the label meanwhile waits for that table instead of building its own"""
    import threading
    import vpin_amd
    L = vpin_amd.lib()
    label = b"test-concurrent-build"
    nb = 700
    xyzt = np.zeros((nb, 128), dtype=np.uint8)
    L.vpin_host_gens_derive.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    assert L.vpin_host_gens_derive(label, nb, xyzt.ctypes.data_as(C.c_void_p)) == 0
    L.vpin_gens_shared.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    ctx2 = vpin_amd.Context(0)
    got, rcs = [None] * 4, [None] * 4

    def ask(i, cx, n):
        h = C.c_void_p()
        rcs[i] = L.vpin_gens_shared(cx.h, label, xyzt.ctypes.data_as(C.c_void_p), n, 1, C.byref(h))
        got[i] = h.value

    ts = [threading.Thread(target=ask, args=(0, ctx, nb)), threading.Thread(target=ask, args=(1, ctx2, nb)),
          threading.Thread(target=ask, args=(2, ctx2, nb - 100)), threading.Thread(target=ask, args=(3, ctx, nb))]
    # contexts are single-threaded objects: one thread per context at a time
    ts[0].start(); ts[1].start(); ts[0].join(); ts[1].join()
    ts[2].start(); ts[3].start(); ts[2].join(); ts[3].join()
    assert rcs == [0, 0, 0, 0]
    assert got[0] == got[1] == got[2] == got[3] and got[0]
    ctx2.close()
