"""CU-masked contexts (vpin_ctx_create_cumask, vpin_ctx_set_cumask_after_phase1: the spatial split of bench.py's four-lane
step) and the A/B switch of the hash layer's one-pass slice evaluation: the proofs are the oracle's, byte for byte
(tests/golden/config_digests.json), whichever CUs run them, alone or side by side.
"""
import hashlib
import json
import os
import threading

import pytest

pytestmark = pytest.mark.gpu

SEED_C = bytes(range(64))
SEED_P = bytes((7 * i + 3) % 256 for i in range(64))

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as _f:
    GOLD = json.load(_f)["cases"]


def _cus(per_xcd_from, per_xcd_to):
    """CUs [from, to) of every XCD, whole shader engines first (mask bit k = CU k/8 of XCD k%8, CU n in engine n%4)"""
    return [x + 8 * ((n // 8) + 4 * (n % 8)) for x in range(8) for n in range(per_xcd_from, per_xcd_to)]


def _prove(ctx, key):
    from vpin_amd import gadgets as G
    g = GOLD[key]
    inp = G.synthetic_mult_inputs(g["label"]) if g["kind"] == "mult" else G.synthetic_add_inputs(g["label"])
    d = ctx.gadget_point_mult_dev(*inp) if g["kind"] == "mult" else ctx.gadget_point_add_dev(*inp)
    try:
        return hashlib.sha256(d.snark_prove(SEED_C, SEED_P)["proof"]).hexdigest()
    finally:
        d.free()


@pytest.mark.parametrize("lo,hi", [(0, 24), (24, 32), (0, 8)])
def test_masked_context_gives_the_oracle_bytes(lo, hi):
    import vpin_amd
    with vpin_amd.Context(0, cu_mask=_cus(lo, hi)) as ctx:
        cus, _ = ctx.device_props()
        for key in ("3_32-add", "3_32-mult", "A-mult"):
            assert _prove(ctx, key) == GOLD[key]["snark_sha256"], key


def test_second_stream_after_phase1_gives_the_oracle_bytes_and_is_left_again():
    import vpin_amd
    with vpin_amd.Context(0) as ctx:
        s0 = ctx.stream
        ctx.set_cumask_after_phase1(_cus(0, 24))
        for key in ("3_32-mult", "A-mult", "A-add"):
            assert _prove(ctx, key) == GOLD[key]["snark_sha256"], key
            assert ctx.stream == s0, "the proof must return on the context's first stream"
        ctx.set_cumask_after_phase1(None)
        assert _prove(ctx, "3_32-mult") == GOLD["3_32-mult"]["snark_sha256"]
    with vpin_amd.Context(0, cu_mask=_cus(0, 8)) as ctx:
        with pytest.raises(vpin_amd.VpinError):
            ctx.set_cumask_after_phase1(_cus(8, 16))   # a masked context has no second stream


def test_disjoint_contexts_side_by_side():
    """the split of bench.py's default step in miniature: a context on 24 CUs per XCD (second stream after phase 1) and two on
    the other 8, proving at the same time"""
    import vpin_amd
    big = vpin_amd.Context(0)
    big.set_cumask_after_phase1(_cus(0, 24))
    small = [vpin_amd.Context(0, cu_mask=_cus(24, 32)) for _ in range(2)]
    got, errs = {}, []

    def run(cx, keys, tag):
        try:
            for k in keys:
                got[(tag, k)] = _prove(cx, k)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(big, ["A-mult", "7_256-mult"], "big")),
          threading.Thread(target=run, args=(small[0], ["3_32-mult", "A-add", "3_32-add"], "s0")),
          threading.Thread(target=run, args=(small[1], ["7_256-add", "3_32-mult"], "s1"))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for cx in [big] + small:
        cx.close()
    assert not errs, errs
    for (tag, k), sha in got.items():
        assert sha == GOLD[k]["snark_sha256"], (tag, k)


def test_invalid_masks_are_refused():
    import ctypes as C
    import numpy as np
    import vpin_amd
    L = vpin_amd.lib()
    h = C.c_void_p()
    zero = np.zeros(8, dtype=np.uint32)
    assert L.vpin_ctx_create_cumask(0, zero.ctypes.data_as(C.c_void_p), 8, C.byref(h)) == -1   # VPIN_EINVAL: no CU enabled
    assert L.vpin_ctx_create_cumask(0, None, 0, C.byref(h)) == -1


def test_hash_layer_two_pass_switch_gives_the_same_bytes():
    """VPIN_HASH_TWO_PASS: DensePolynomial::evaluate per slice and DensePolynomial::bound as two passes over the combined
    polynomials (the path before round 5, still the one a proof split over several GPUs takes) against the one-pass default"""
    import vpin_amd
    with vpin_amd.Context(0) as ctx:
        for key in ("3_32-add", "A-mult", "7_256-mult"):
            os.environ["VPIN_HASH_TWO_PASS"] = "1"
            try:
                a = _prove(ctx, key)
            finally:
                del os.environ["VPIN_HASH_TWO_PASS"]
            assert a == GOLD[key]["snark_sha256"] == _prove(ctx, key), key


def test_low_memory_mode_gives_the_same_bytes():
    """vpin_ctx_set_low_memory: the mem forest built after the ops forest is proven, its roots from a product reduction over the
    leaves (spark_mem_roots) -- the same field elements, the same SNARK"""
    import vpin_amd
    with vpin_amd.Context(0) as ctx:
        ctx.set_low_memory(True)
        for key in ("3_32-add", "3_32-mult", "A-mult", "7_256-add", "E-mult", "L5-mult"):
            assert _prove(ctx, key) == GOLD[key]["snark_sha256"], key
        ctx.set_low_memory(False)
        assert _prove(ctx, "A-mult") == GOLD["A-mult"]["snark_sha256"]
