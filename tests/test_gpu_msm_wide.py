"""GPU parity for the MSM paths only large instances reach, against the oracle's Pippenger
(oracle/group.c ge_msm: Spartan/src/group.rs:103-122) on the same generators and scalars, bit-exact:
  * msm_wide_kernel<4> (few-row MSMs of >= 8192 scalars: the bullet-reduction rows of the 2^25 instance),
  * the two-segment table (12-bit windows for generators < 16386, narrow ones beyond; 80 GB budget),
  * the kSeg = 8192 zero-compaction segment boundary inside a row,
  * a 1024 x 1024 commitment (CNN A's witness shape), sampled rows.
The table is the shared one of label b"gens_r1cs_eval" with the real SHAKE stream (32770 generators = what the 2^25
instance's SNARK::encode needs), so later tests in the session reuse it instead of building a second 77 GB table.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
Q = M.Q
NB = 32770  # R = 2^15 columns + gens_1 base + h (PolyCommitmentGens::new(29, ..), dense_mlpoly.rs:26-40)


@pytest.fixture(scope="module")
def ctx():
    import vpin_amd
    c = vpin_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def big(ctx):
    xyzt, og = O.gens_stream_xyzt(NB, b"gens_r1cs_eval")
    g = ctx.gens_shared("gens_r1cs_eval", xyzt, 80 if ctx.device_total_bytes() >= (200 << 30) else 24)  # the library's own budgets (spark.cpp)
    import vpin_amd
    lay = (C.c_size_t * 6)()
    L = vpin_amd.lib()
    L.vpin_gens_layout.argtypes = [C.c_void_p, C.c_void_p]
    assert L.vpin_gens_layout(g.h, lay) == 0
    return g, og, list(lay)


def scalars(rng, n, zero_frac=0.35):
    """witness-like mix (Montgomery form): zeros, small values, q-1, full width"""
    t = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    t[:, 3] &= np.uint64((1 << 60) - 1)
    k = rng.random(n)
    t[k < zero_frac] = 0
    small = M.ints_to_table([1, 2, 3, 2**16 - 1, Q - 1, Q - 2])
    idx = np.nonzero((k >= zero_frac) & (k < zero_frac + 0.15))[0]
    t[idx] = small[rng.integers(0, len(small), size=len(idx))]
    return t


def oracle_msm(og, sc, base0=0):
    """compressed sum_i sc[i] * g[base0 + i] via the oracle's Pippenger"""
    L = O.lib()
    n = sc.shape[0]
    pts = (O.Ge * n)(*[og[base0 + i] for i in range(n)])
    r = O.Ge()
    L.ge_msm(C.byref(r), O.ptr(np.ascontiguousarray(sc)), pts, n)
    out = (C.c_uint8 * 32)()
    L.ge_compress(out, C.byref(r))
    return bytes(out)


def sum_parts(parts_row):
    L = O.lib()
    acc = O.Ge()
    L.ge_identity(C.byref(acc))
    for p in parts_row:
        q = O.Ge()
        L.ge_from_xyzt(C.byref(q), p.ctypes.data_as(C.c_void_p))
        L.ge_add(C.byref(acc), C.byref(acc), C.byref(q))
    out = (C.c_uint8 * 32)()
    L.ge_compress(out, C.byref(acc))
    return bytes(out)


def test_table_has_two_segments_on_a_288gb_part(ctx, big):
    g, og, lay = big
    c, W, split, c_hi, W_hi, bases = lay
    assert bases >= NB
    if ctx.device_total_bytes() >= (200 << 30):
        assert split == 16386 and c == 12 and 6 < c_hi < 12, lay  # the layout DESIGN.md section 3 describes


@pytest.mark.parametrize("ncols", [8192, 16384, 32768])
def test_wide_rows_parts_vs_oracle(ctx, big, ncols):
    """vpin_gens_msm_parts with >= 8192 columns: msm_wide_kernel<4> + parts_reduce_kernel; 32768 columns cross into
    the narrow-window segment"""
    g, og, _ = big
    rng = np.random.default_rng(ncols)
    rows = 2
    sc = scalars(rng, rows * ncols)
    parts = ctx.gens_msm_parts(g, sc, rows, ncols)
    for i in range(rows):
        assert sum_parts(parts[i]) == oracle_msm(og, sc[i * ncols:(i + 1) * ncols]), (ncols, i)


@pytest.mark.parametrize("ncols", [8192, 16384, 32768])
def test_few_long_rows_vs_oracle(ctx, big, ncols):
    """vpin_gens_msm (msm_rows_kernel, chunked because rows < 128) on long rows"""
    g, og, _ = big
    rng = np.random.default_rng(100 + ncols)
    sc = scalars(rng, 3 * ncols)
    got = ctx.gens_msm(g, sc, 3, ncols)
    for i in range(3):
        assert bytes(got[i]) == oracle_msm(og, sc[i * ncols:(i + 1) * ncols]), (ncols, i)


def test_nonzeros_straddling_the_compaction_segment(ctx, big):
    """rows >= 128 take the un-chunked path where one workgroup compacts the row's non-zeros 8192 scalars at a time:
    rows whose only non-zeros sit around index 8192, at the very ends, and a dense one"""
    g, og, _ = big
    rng = np.random.default_rng(77)
    Ls, Rs = 128, 16384
    Z = np.zeros((Ls * Rs, 4), dtype=np.uint64)
    dense = scalars(rng, Rs, zero_frac=0.0)
    pat = {0: range(8185, 8200), 1: [8191], 2: [8192], 3: [0, Rs - 1], 4: range(8192 - 40, 8192 + 40, 3), 5: range(Rs), 127: [8191, 8192]}
    for row, idx in pat.items():
        idx = list(idx)
        Z[row * Rs + np.array(idx)] = dense[idx]
    blinds = scalars(rng, Ls, zero_frac=0.0)
    got = ctx.hyrax_commit(g, ctx.upload(Z), blinds, Rs + 1)
    for row in list(pat) + [6, 64]:
        exp = O.hyrax_commit(Z[row * Rs:(row + 1) * Rs], 1, blinds[row:row + 1], og, Rs + 1)
        assert np.array_equal(got[row], exp[0]), row


def test_1024x1024_commitment_sampled_rows(ctx, big):
    g, og, _ = big
    rng = np.random.default_rng(1024)
    Ls = Rs = 1024
    Z = scalars(rng, Ls * Rs)
    Z[Ls * Rs - 300 * Rs:] = 0  # padded tail rows
    blinds = scalars(rng, Ls, zero_frac=0.0)
    got = ctx.hyrax_commit(g, ctx.upload(Z), blinds, Rs + 1)
    for row in (0, 1, 511, 723, 724, 1023):
        exp = O.hyrax_commit(Z[row * Rs:(row + 1) * Rs], 1, blinds[row:row + 1], og, Rs + 1)
        assert np.array_equal(got[row], exp[0]), row


def test_full_width_commitment_rows_in_both_segments(ctx, big):
    """4 rows x 32768 columns through vpin_hyrax_commit with blinds at g[32769]: SNARK::encode's shape for the 2^25
    instance (columns 16386.. use the narrow-window segment)"""
    g, og, _ = big
    rng = np.random.default_rng(4)
    Ls, Rs = 4, 32768
    Z = scalars(rng, Ls * Rs, zero_frac=0.2)
    Z[3 * Rs:] = M.ints_to_table([5])[0]  # a constant row: the prefix-sum base path
    blinds = scalars(rng, Ls, zero_frac=0.0)
    got = ctx.hyrax_commit(g, ctx.upload(Z), blinds, Rs + 1)
    exp = O.hyrax_commit(Z, Ls, blinds, og, Rs + 1)
    assert np.array_equal(got, exp)
