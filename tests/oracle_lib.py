"""ctypes binding for the CPU oracle (oracle/_build/liboracle.so).

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module.  The product package (vpin_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class Fq(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]

    def limbs(self):
        return [int(x) for x in self.l]


class Fe(C.Structure):
    _fields_ = [("l", C.c_uint64 * 5)]


class Ge(C.Structure):
    _fields_ = [("X", Fe), ("Y", Fe), ("Z", Fe), ("T", Fe)]


class Shake(C.Structure):
    _fields_ = [("st", C.c_uint64 * 25), ("pos", C.c_size_t), ("squeezing", C.c_int)]


class Merlin(C.Structure):
    _fields_ = [("st", C.c_uint8 * 200), ("pos", C.c_uint8), ("pos_begin", C.c_uint8), ("cur_flags", C.c_uint8)]


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    P = C.POINTER(Fq)
    for name in ("fq_add", "fq_sub", "fq_mul"):
        getattr(L, name).restype = Fq
        getattr(L, name).argtypes = [P, P]
    for name in ("fq_neg", "fq_square", "fq_invert"):
        getattr(L, name).restype = Fq
        getattr(L, name).argtypes = [P]
    L.fq_from_u64.restype = Fq
    L.fq_from_u64.argtypes = [C.c_uint64]
    L.fq_from_raw.restype = Fq
    L.fq_from_raw.argtypes = [_u64p]
    L.fq_from_bytes.restype = C.c_int
    L.fq_from_bytes.argtypes = [P, _u8p]
    L.fq_to_bytes.restype = None
    L.fq_to_bytes.argtypes = [_u8p, P]
    L.fq_from_bytes_wide.restype = Fq
    L.fq_from_bytes_wide.argtypes = [_u8p]
    L.fq_from_bytes_mod_order.restype = Fq
    L.fq_from_bytes_mod_order.argtypes = [_u8p]
    L.fq_pow_vartime.restype = Fq
    L.fq_pow_vartime.argtypes = [P, _u64p]
    L.fq_batch_invert.restype = Fq
    L.fq_batch_invert.argtypes = [C.c_void_p, C.c_size_t]
    L.oracle_eq_evals.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_bound_poly_var_top.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.oracle_sc_cubic_round.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    L.oracle_sc_quad_round.argtypes = [C.c_void_p] * 2 + [C.c_size_t, C.c_void_p]
    L.oracle_poly_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.oracle_dotproduct.restype = Fq
    L.oracle_dotproduct.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.oracle_poly_evaluate.restype = Fq
    L.oracle_poly_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.oracle_unipoly_from_evals.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_unipoly_evaluate.restype = Fq
    L.oracle_unipoly_evaluate.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tr_challenge_scalar.restype = Fq
    L.ge_eq.restype = C.c_int
    L.ge_decompress.restype = C.c_int
    L.ge_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.oracle_gens_new.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.oracle_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_hyrax_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]
    L.merlin_init.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.merlin_append_message.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.merlin_challenge_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.shake256_absorb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.shake256_squeeze.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for fn in ("oracle_eq_evals", "oracle_bound_poly_var_top", "oracle_sc_cubic_round",
               "oracle_sc_quad_round", "oracle_poly_bound", "oracle_unipoly_from_evals"):
        getattr(L, fn).restype = None


# ---- numpy helpers: a table is an (n,4) uint64 C-contiguous array (Montgomery limbs) ----

def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def fq_from_limbs(limbs):
    f = Fq()
    for i in range(4):
        f.l[i] = int(limbs[i])
    return f


def fq_arr(f):
    return np.array(f.limbs(), dtype=np.uint64)


def eq_evals(r):
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((1 << r.shape[0], 4), dtype=np.uint64)
    lib().oracle_eq_evals(ptr(r), r.shape[0], ptr(out))
    return out


def bound_top(Z, r):
    """returns the folded copy (length n/2)."""
    Z = np.array(Z, dtype=np.uint64).reshape(-1, 4).copy()
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    lib().oracle_bound_poly_var_top(ptr(Z), Z.shape[0], ptr(r))
    return Z[: Z.shape[0] // 2].copy()


def sc_cubic_round(A, B, Cc, D):
    out = np.zeros((3, 4), dtype=np.uint64)
    lib().oracle_sc_cubic_round(ptr(A), ptr(B), ptr(Cc), ptr(D), A.shape[0], ptr(out))
    return out


def sc_quad_round(A, B):
    out = np.zeros((2, 4), dtype=np.uint64)
    lib().oracle_sc_quad_round(ptr(A), ptr(B), A.shape[0], ptr(out))
    return out


# ---- R1CS / sat proof -------------------------------------------------------------------------

class R1CS(C.Structure):
    _fields_ = [("num_cons", C.c_size_t), ("num_vars", C.c_size_t), ("num_inputs", C.c_size_t),
                ("nnz", C.c_size_t * 3), ("row", C.c_void_p * 3), ("col", C.c_void_p * 3), ("val", C.c_void_p * 3)]


def make_r1cs(inst):
    """inst: dict from gadgets_model.instance_new. Keeps references alive on the struct."""
    r = R1CS()
    r.num_cons, r.num_vars, r.num_inputs = inst["num_cons"], inst["num_vars"], inst["num_inputs"]
    keep = []
    for m, name in enumerate("ABC"):
        rows, cols, vals = (np.ascontiguousarray(x) for x in inst[name])
        keep += [rows, cols, vals]
        r.nnz[m] = len(rows)
        r.row[m] = rows.ctypes.data
        r.col[m] = cols.ctypes.data
        r.val[m] = vals.ctypes.data
    r._keep = keep
    return r


def log2(n):
    return int(n).bit_length() - 1


def sat_prove(inst, seed_commit, seed_proof, threads=8):
    L = lib()
    L.oracle_sat_proof_max_bytes.restype = C.c_size_t
    L.oracle_sat_proof_max_bytes.argtypes = [C.c_size_t, C.c_size_t]
    L.oracle_vpin_sat_prove.restype = C.c_size_t
    L.oracle_vpin_sat_prove.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 5
    r = make_r1cs(inst)
    nv, nc = inst["num_vars"], inst["num_cons"]
    ell = log2(nv)
    Lsz = 1 << (ell // 2)
    cap = L.oracle_sat_proof_max_bytes(nc, nv)
    proof = np.zeros(cap, dtype=np.uint8)
    comm_para = np.zeros((Lsz, 32), dtype=np.uint8)
    comm_input = np.zeros((Lsz, 32), dtype=np.uint8)
    evals = np.zeros((3, 4), dtype=np.uint64)
    rx = np.zeros((log2(nc), 4), dtype=np.uint64)
    ry = np.zeros((log2(nv) + 1, 4), dtype=np.uint64)
    sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
    sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
    inputs = np.ascontiguousarray(inst["inputs"])
    n = L.oracle_vpin_sat_prove(C.byref(r), ptr(inst["vars_para"]), ptr(inst["vars_input"]), ptr(inst["vars"]),
                                inputs.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                sp.ctypes.data_as(C.c_void_p), threads, proof.ctypes.data_as(C.c_void_p), cap,
                                comm_para.ctypes.data_as(C.c_void_p), comm_input.ctypes.data_as(C.c_void_p),
                                ptr(evals), ptr(rx), ptr(ry))
    return dict(proof=bytes(proof[:n]), comm_para=comm_para, comm_input=comm_input, inst_evals=evals, rx=rx, ry=ry)


def sat_verify(inst, res, proof=None):
    L = lib()
    L.oracle_vpin_sat_verify.restype = C.c_int
    L.oracle_vpin_sat_verify.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    pb = np.frombuffer(proof if proof is not None else res["proof"], dtype=np.uint8).copy()
    rx = np.zeros((64, 4), dtype=np.uint64)
    ry = np.zeros((64, 4), dtype=np.uint64)
    inputs = np.ascontiguousarray(inst["inputs"])
    return L.oracle_vpin_sat_verify(pb.ctypes.data_as(C.c_void_p), len(pb), inst["num_cons"], inst["num_vars"],
                                    inputs.ctypes.data_as(C.c_void_p), inst["num_inputs"],
                                    ptr(np.ascontiguousarray(res["inst_evals"])),
                                    res["comm_para"].ctypes.data_as(C.c_void_p),
                                    res["comm_input"].ctypes.data_as(C.c_void_p), ptr(rx), ptr(ry))


def is_sat(inst):
    L = lib()
    L.oracle_r1cs_is_sat.restype = C.c_int
    r = make_r1cs(inst)
    inputs = np.ascontiguousarray(inst["inputs"])
    return L.oracle_r1cs_is_sat(C.byref(r), ptr(inst["vars"]), inputs.ctypes.data_as(C.c_void_p))


def sat_timings():
    out = (C.c_double * 5)()
    lib().oracle_sat_last_timings(out)
    return dict(zip(("polycommit", "sc_phase_one", "sc_phase_two", "polyeval", "total"), out))


def gens_stream_xyzt(nb, label=b"gens_r1cs_sat"):
    """First nb points of MultiCommitGens::new's stream under `label`, as (nb,128) uint8 X|Y|Z|T,
    plus the oracle's own ge_t array (for oracle-side commits)."""
    L = lib()
    gens = (Ge * nb)()
    lab = (C.c_uint8 * len(label))(*label)
    L.oracle_gens_new(gens, nb - 1, lab, len(label))
    out = np.zeros((nb, 128), dtype=np.uint8)
    for i in range(nb):
        L.ge_to_xyzt(out[i].ctypes.data_as(C.c_void_p), C.byref(gens[i]))
    return out, gens


def hyrax_commit(Z, Ls, blinds, gens, blind_index, threads=8):
    Z = np.ascontiguousarray(Z, dtype=np.uint64).reshape(-1, 4)
    Rs = Z.shape[0] // Ls
    out = np.zeros((Ls, 32), dtype=np.uint8)
    lib().oracle_hyrax_commit(out.ctypes.data_as(C.c_void_p), ptr(Z), Ls, Rs,
                              ptr(np.ascontiguousarray(blinds, dtype=np.uint64)), gens, C.byref(gens[blind_index]),
                              threads)
    return out


# ---- SPARK / whole SNARK ------------------------------------------------------------------------

def _spark_decl(L):
    L.oracle_spark_comm_bytes.restype = C.c_size_t
    L.oracle_spark_comm_bytes.argtypes = [C.c_void_p]
    L.oracle_snark_proof_max_bytes.restype = C.c_size_t
    L.oracle_snark_proof_max_bytes.argtypes = [C.c_void_p]
    L.oracle_spark_encode.restype = C.c_void_p
    L.oracle_spark_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.oracle_spark_decomm_free.restype = None
    L.oracle_spark_decomm_free.argtypes = [C.c_void_p]
    L.oracle_vpin_snark_prove.restype = C.c_size_t
    L.oracle_vpin_snark_prove.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.oracle_vpin_snark_verify.restype = C.c_int
    L.oracle_vpin_snark_verify.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p]


def snark_prove(inst, seed_commit, seed_proof, threads=8):
    """SNARK::encode + the whole my_lib_prove: returns dict(proof, comm, comm_para, comm_input)."""
    L = lib()
    _spark_decl(L)
    r = make_r1cs(inst)
    ccap = L.oracle_spark_comm_bytes(C.byref(r))
    comm = np.zeros(ccap, dtype=np.uint8)
    clen = C.c_size_t(0)
    dec = L.oracle_spark_encode(C.byref(r), threads, comm.ctypes.data_as(C.c_void_p), ccap, C.byref(clen))
    assert dec, "oracle_spark_encode failed"
    try:
        nv = inst["num_vars"]
        Lsz = 1 << (log2(nv) // 2)
        cap = L.oracle_snark_proof_max_bytes(C.byref(r))
        proof = np.zeros(cap, dtype=np.uint8)
        comm_para = np.zeros((Lsz, 32), dtype=np.uint8)
        comm_input = np.zeros((Lsz, 32), dtype=np.uint8)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
        inputs = np.ascontiguousarray(inst["inputs"])
        n = L.oracle_vpin_snark_prove(C.byref(r), dec, ptr(inst["vars_para"]), ptr(inst["vars_input"]),
                                      ptr(inst["vars"]), inputs.ctypes.data_as(C.c_void_p),
                                      sc.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p), threads,
                                      proof.ctypes.data_as(C.c_void_p), cap,
                                      comm_para.ctypes.data_as(C.c_void_p), comm_input.ctypes.data_as(C.c_void_p))
    finally:
        L.oracle_spark_decomm_free(dec)
    return dict(proof=bytes(proof[:n]), comm=bytes(comm[:clen.value]), comm_para=comm_para, comm_input=comm_input)


def snark_verify(inst, res, proof=None, comm=None):
    L = lib()
    _spark_decl(L)
    pb = np.frombuffer(proof if proof is not None else res["proof"], dtype=np.uint8).copy()
    cb = np.frombuffer(comm if comm is not None else res["comm"], dtype=np.uint8).copy()
    inputs = np.ascontiguousarray(inst["inputs"])
    return L.oracle_vpin_snark_verify(pb.ctypes.data_as(C.c_void_p), len(pb), cb.ctypes.data_as(C.c_void_p), len(cb),
                                      inputs.ctypes.data_as(C.c_void_p), inst["num_inputs"],
                                      res["comm_para"].ctypes.data_as(C.c_void_p),
                                      res["comm_input"].ctypes.data_as(C.c_void_p))


def spark_timings():
    out = (C.c_double * 7)()
    lib().oracle_spark_last_timings(out)
    return dict(zip(("encode", "sat", "derefs_commit", "network_build", "product_layer", "hash_layer", "total"), out))
