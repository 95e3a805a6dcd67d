"""GPU parity: vpin_hyrax_commit_pippenger (bucket accumulation staged in LDS, msm_pip.hip -- the MSM north_star names, kept as a
measured alternative) against the CPU oracle's Pippenger and against the window-table walk (vpin_hyrax_commit) on the same
generators and scalars: bit-exact compressed points for every window width."""
import numpy as np
import pytest

import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
Q = M.Q


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


def _full_width(rng, n):
    return [int.from_bytes(rng.bytes(40), "little") % Q for _ in range(n)]


def _witness_like(rng, n):
    vals = []
    for _ in range(n):
        k = rng.random()
        vals.append(0 if k < 0.35 else 1 if k < 0.45 else int(rng.integers(0, 2**16)) if k < 0.5
                    else Q - 1 if k < 0.53 else int(rng.integers(0, 2**62)) ** 4 % Q)
    return vals


def _edge_scalars(c):
    """digits equal to 2^(c-1) (kept positive), just above it (negated, carry), runs of carries, the top window's largest values"""
    half, full = 1 << (c - 1), 1 << c
    vals = [0, 1, half, half + 1, full - 1, full, Q - 1, Q - 2, (Q - 1) // 2, 1 << 252, (1 << 252) - 1,
            sum(half << (c * w) for w in range(252 // c)),              # every digit exactly 2^(c-1)
            sum((half + 1) << (c * w) for w in range(252 // c)),        # every digit negated with a carry
            sum((full - 1) << (c * w) for w in range(252 // c)) % Q,    # carries rippling through every window
            (1 << 252) + (1 << 124)]
    return [v % Q for v in vals]


@pytest.mark.parametrize("c_bits", [0, 9, 10, 11, 12])
def test_small_rows_vs_oracle(ctx, c_bits):
    Rs, Ls = 32, 8
    xyzt, og = O.gens_stream_xyzt(Rs + 2)
    g = ctx.gens_create(xyzt)
    rng = np.random.default_rng(100 + c_bits)
    vals = _witness_like(rng, Ls * Rs)
    edges = _edge_scalars(c_bits or 9)
    vals[:len(edges)] = edges
    Z = M.ints_to_table(vals)
    blinds = M.ints_to_table(_full_width(rng, Ls))
    dZ = ctx.upload(Z)
    exp = O.hyrax_commit(Z, Ls, blinds, og, Rs + 1)
    assert np.array_equal(ctx.hyrax_commit_pippenger(g, dZ, blinds, Rs + 1, c_bits=c_bits), exp)
    # commit(gens, None): no blind
    zero = np.zeros((Ls, 4), dtype=np.uint64)
    assert np.array_equal(ctx.hyrax_commit_pippenger(g, dZ, None, 0, Ls=Ls, c_bits=c_bits), O.hyrax_commit(Z, Ls, zero, og, Rs + 1))
    dZ.free()
    g.free()


@pytest.mark.parametrize("Rs,Ls,c_bits", [(256, 8, 0), (256, 8, 12), (1024, 4, 10), (4096, 16, 0), (4096, 4, 11), (4096, 4, 12)])
def test_rows_vs_table_walk_and_oracle(ctx, Rs, Ls, c_bits):
    xyzt, og = O.gens_stream_xyzt(Rs + 2)
    g = ctx.gens_create(xyzt)
    rng = np.random.default_rng(Rs + Ls + c_bits)
    rows = [_full_width(rng, Rs) if i % 2 == 0 else _witness_like(rng, Rs) for i in range(Ls)]
    rows[-1] = [rows[-1][0]] * Rs          # a constant row (the table walk's prefix-sum shortcut; plain buckets here)
    if Ls > 2:
        rows[1] = [0] * Rs                 # an all-zero row: every window empty
        rows[2] = [int(rng.integers(0, 2)) for _ in range(Rs)]   # bits: only window 0 is populated
    Z = M.ints_to_table([v for r in rows for v in r])
    blinds = M.ints_to_table(_full_width(rng, Ls))
    dZ = ctx.upload(Z)
    walk = ctx.hyrax_commit(g, dZ, blinds, Rs + 1)
    got = ctx.hyrax_commit_pippenger(g, dZ, blinds, Rs + 1, c_bits=c_bits)
    assert np.array_equal(got, walk)
    k = min(Ls, 3)
    assert np.array_equal(got[:k], O.hyrax_commit(Z[:k * Rs], k, blinds[:k], og, Rs + 1))
    dZ.free()
    g.free()


def test_more_pairs_than_the_persistent_grid(ctx):
    """768 workgroups loop over rows x W (row, window) pairs: 64 rows x 29 windows = 1856 pairs, every workgroup takes several"""
    Rs, Ls = 512, 64
    xyzt, og = O.gens_stream_xyzt(Rs + 2)
    g = ctx.gens_create(xyzt)
    rng = np.random.default_rng(77)
    Z = M.ints_to_table(_full_width(rng, Ls * Rs))
    blinds = M.ints_to_table(_full_width(rng, Ls))
    dZ = ctx.upload(Z)
    got = ctx.hyrax_commit_pippenger(g, dZ, blinds, Rs + 1, c_bits=9)
    assert np.array_equal(got, ctx.hyrax_commit(g, dZ, blinds, Rs + 1))
    # the rows in chunks (long commitments bound their digit buffer to 2 GiB): 64 rows as 7 + 7 + .. + 1
    import os
    os.environ["VPIN_PIP_DIGIT_BYTES"] = str(7 * 29 * (Rs + 1) * 2)
    try:
        assert np.array_equal(ctx.hyrax_commit_pippenger(g, dZ, blinds, Rs + 1, c_bits=9), got)
    finally:
        del os.environ["VPIN_PIP_DIGIT_BYTES"]
    assert np.array_equal(got[:2], O.hyrax_commit(Z[:2 * Rs], 2, blinds[:2], og, Rs + 1))
    dZ.free()
    g.free()


def test_refused_arguments(ctx):
    import vpin_amd
    xyzt, _ = O.gens_stream_xyzt(34)
    g = ctx.gens_create(xyzt)
    dZ = ctx.upload(np.zeros((64, 4), dtype=np.uint64))
    with pytest.raises(vpin_amd.VpinError):
        ctx.hyrax_commit_pippenger(g, dZ, None, 0, Ls=2, c_bits=8)       # fewer buckets than lanes
    with pytest.raises(vpin_amd.VpinError):
        ctx.hyrax_commit_pippenger(g, dZ, None, 0, Ls=1, c_bits=0)       # R = 64 > 34 generators
    with pytest.raises(vpin_amd.VpinError):
        ctx.hyrax_commit_pippenger(g, dZ, None, 0, Ls=3, c_bits=0)       # 64 scalars are not 3 rows
    dZ.free()
    g.free()


@pytest.mark.parametrize("mode", ["1", "11"])
def test_whole_snarks_with_every_row_commitment_by_buckets(ctx, mode):
    """VPIN_MSM_PIPPENGER: the provers' row commitments (witness, the SPARK polynomials, the derefs polynomial without its
    hot-column shortcut) by the bucket method instead of the table walk -- the SNARKs are the oracle's, byte for byte"""
    import hashlib
    import json
    import os
    from vpin_amd import gadgets as G
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        gold = json.load(f)["cases"]
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    os.environ["VPIN_MSM_PIPPENGER"] = mode
    try:
        for key in ("3_32-add", "3_32-mult", "A-mult", "7_256-mult") + (("E-mult",) if mode == "1" else ()):
            g = gold[key]
            inp = G.synthetic_mult_inputs(g["label"]) if g["kind"] == "mult" else G.synthetic_add_inputs(g["label"])
            d = ctx.gadget_point_mult_dev(*inp) if g["kind"] == "mult" else ctx.gadget_point_add_dev(*inp)
            try:
                res = d.snark_prove(seed_c, seed_p)
            finally:
                d.free()
            assert hashlib.sha256(res["proof"]).hexdigest() == g["snark_sha256"], key
            assert hashlib.sha256(res["comm"]).hexdigest() == g["comm_sha256"], key
    finally:
        del os.environ["VPIN_MSM_PIPPENGER"]


def test_whole_snarks_by_buckets_at_config_size_take_their_rows_in_chunks(ctx):
    """L3-mult (2^22) and L5-mult (2^25 constraints) with every row commitment by the bucket method.  The 2^25 instance's
    derefs polynomial is 2^14 rows x 2^14 scalars: at c = 9 a row's digits take 29 x 16385 x 2 B, so the 2 GiB digit buffer
    holds ~2260 rows and the row loop of pip_rows_c really runs 8 times for that one commitment (the chunking had
    only been exercised under a forced cap on a 64-row polynomial).  Its 2^14 x 2^15 polynomials (SNARK::encode's) have 32769 terms
    per row with the blind: one more than a list entry's 15-bit column holds, so those stay on the table walk (msm.hip
    msm_rows).  Bytes = the oracle's (config digests)."""
    import hashlib
    import json
    import os
    from vpin_amd import gadgets as G
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")) as f:
        gold = json.load(f)["cases"]
    seed_c, seed_p = bytes(range(64)), bytes((7 * i + 3) % 256 for i in range(64))
    os.environ["VPIN_MSM_PIPPENGER"] = "1"
    try:
        for key, min_chunks in (("L3-mult", 3), ("L5-mult", 12)):
            g = gold[key]
            d = ctx.gadget_point_mult_dev(*G.synthetic_mult_inputs(g["label"]))
            before = ctx.pip_row_chunks()
            try:
                res = d.snark_prove(seed_c, seed_p)
            finally:
                d.free()
            assert hashlib.sha256(res["proof"]).hexdigest() == g["snark_sha256"], key
            assert hashlib.sha256(res["comm"]).hexdigest() == g["comm_sha256"], key
            assert ctx.pip_row_chunks() - before >= min_chunks, (key, ctx.pip_row_chunks() - before)
    finally:
        del os.environ["VPIN_MSM_PIPPENGER"]
