"""ctypes binding of include/vpin_hip.h (the C ABI of libvpin_hip.so)."""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "vpin_hip.h")

KERNEL_CLASSES = {
    0: "sc_cubic", 1: "sc_quad", 2: "sc_bind", 3: "sc_cubic_fused", 4: "sc_quad_fused", 5: "eq", 6: "msm", 7: "sc_tail",
    8: "spark_round", 9: "spark_build", 10: "spark_round_big", 11: "msm_rows", 12: "spark_tail",
}
K_COUNT = 16


class VpinError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        L = lib()
        msg = L.vpin_strerror(code).decode()
        last = L.vpin_last_error().decode()
        super().__init__(f"{where}: {msg} ({code})" + (f" [{last}]" if last else ""))


class R1CS(C.Structure):
    _fields_ = [("num_cons", C.c_size_t), ("num_vars", C.c_size_t), ("num_inputs", C.c_size_t),
                ("nnz", C.c_size_t * 3), ("row", C.c_void_p * 3), ("col", C.c_void_p * 3), ("val", C.c_void_p * 3)]


def make_r1cs(inst):
    """inst: dict with num_cons/num_vars/num_inputs and A/B/C = (rows u32, cols u32, vals (n,4) u64)."""
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


class KStat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms", C.c_double), ("alg_bytes", C.c_double), ("units", C.c_double)]


def lib_path():
    return os.path.join(HERE, "lib", "libvpin_hip.so")


_lib = None


def lib():
    """Load libvpin_hip.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(
            f"{p} is missing: build it with `python vpin_amd/build.py` (hipcc, gfx950). "
            "vpin_amd has no CPU fallback.")
    L = C.CDLL(p)
    vp = C.c_void_p
    L.vpin_strerror.restype = C.c_char_p
    L.vpin_strerror.argtypes = [C.c_int]
    L.vpin_last_error.restype = C.c_char_p
    L.vpin_abi_version.restype = C.c_int
    L.vpin_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vpin_ctx_create_prio.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.vpin_ctx_create_cumask.argtypes = [C.c_int, vp, C.c_uint32, C.POINTER(vp)]
    L.vpin_ctx_set_cumask_after_phase1.argtypes = [vp, vp, C.c_uint32]
    L.vpin_ctx_destroy.argtypes = [vp]
    L.vpin_ctx_destroy.restype = None
    L.vpin_ctx_stream.argtypes = [vp]
    L.vpin_ctx_stream.restype = vp
    L.vpin_ctx_sync.argtypes = [vp]
    L.vpin_ctx_set_shared_device.argtypes = [vp, C.c_int]
    L.vpin_ctx_set_progress_flag.argtypes = [vp, vp]
    L.vpin_table_upload.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_table_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_table_wrap.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_table_clone.argtypes = [vp, vp, C.POINTER(vp)]
    L.vpin_table_free.argtypes = [vp, vp]
    L.vpin_table_free.restype = None
    L.vpin_table_len.argtypes = [vp]
    L.vpin_table_len.restype = C.c_size_t
    L.vpin_table_device_ptr.argtypes = [vp]
    L.vpin_table_device_ptr.restype = vp
    L.vpin_table_read.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    L.vpin_sc_cubic_round.argtypes = [vp, vp, vp, vp, vp, vp]
    L.vpin_sc_quad_round.argtypes = [vp, vp, vp, vp]
    L.vpin_sc_bind.argtypes = [vp, C.POINTER(vp), C.c_int, vp]
    L.vpin_sc_cubic_bind_round.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.vpin_sc_quad_bind_round.argtypes = [vp, vp, vp, vp, vp]
    L.vpin_eq_table.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.vpin_gens_create.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_gens_free.argtypes = [vp, vp]
    L.vpin_gens_free.restype = None
    L.vpin_gens_count.argtypes = [vp]
    L.vpin_gens_entry_bytes.restype = C.c_size_t
    L.vpin_gens_entry_bytes.argtypes = []
    L.vpin_gens_count.restype = C.c_size_t
    L.vpin_hyrax_commit.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp]
    L.vpin_hyrax_commit_pair.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp]
    L.vpin_hyrax_commit_pippenger.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp]
    L.vpin_gens_msm.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp]
    L.vpin_poly_bound.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.vpin_sat_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp, vp, vp, vp]
    L.vpin_sat_prove_resident.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp, vp, vp, vp]
    L.vpin_r1cs_upload.argtypes = [vp, vp, C.POINTER(vp)]
    L.vpin_r1cs_free.argtypes = [vp, vp]
    L.vpin_r1cs_free.restype = None
    L.vpin_r1cs_dims.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.vpin_r1cs_dims.restype = None
    L.vpin_r1cs_build_z.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
    L.vpin_r1cs_multiply_vec.argtypes = [vp, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.vpin_r1cs_eval_table.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
    L.vpin_r1cs_evaluate.argtypes = [vp, vp, vp, vp, vp]
    L.vpin_sat_proof_max_bytes.argtypes = [C.c_size_t, C.c_size_t]
    L.vpin_sat_proof_max_bytes.restype = C.c_size_t
    L.vpin_sat_last_timings.argtypes = [C.POINTER(C.c_double)]
    L.vpin_sat_prepare.argtypes = [vp, C.c_size_t]
    L.vpin_gens_shared.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vpin_gens_shared_clear.restype = None
    L.vpin_gens_shared_clear.argtypes = []
    L.vpin_spark_comm_bytes.restype = C.c_size_t
    L.vpin_spark_comm_bytes.argtypes = [vp]
    L.vpin_snark_proof_max_bytes.restype = C.c_size_t
    L.vpin_snark_proof_max_bytes.argtypes = [vp]
    L.vpin_spark_encode.argtypes = [vp, vp, C.POINTER(vp), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vpin_spark_decomm_free.restype = None
    L.vpin_spark_decomm_free.argtypes = [vp, vp]
    L.vpin_snark_prove_resident.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp]
    L.vpin_snark_prove.argtypes = [vp] * 9 + [C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp]
    L.vpin_sat_verify.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp, vp, vp]
    L.vpin_snark_verify.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, vp]
    L.vpin_spark_last_timings.restype = None
    L.vpin_spark_last_timings.argtypes = [C.POINTER(C.c_double)]
    L.vpin_sat_last_timings.restype = None
    L.vpin_host_gens_derive.argtypes = [C.c_char_p, C.c_size_t, vp]
    L.vpin_host_merlin_kat.argtypes = [C.c_char_p, C.c_char_p, vp, C.c_size_t, C.c_char_p, vp, C.c_size_t]
    L.vpin_host_commit.argtypes = [C.c_char_p, vp, C.c_size_t, vp, vp]
    L.vpin_gadget_point_add_dev.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_gadget_point_mult_dev.argtypes = [vp, vp, vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_dev_instance_free.argtypes = [vp, vp]
    L.vpin_dev_instance_free.restype = None
    for name in ("r1cs", "vars_para", "vars_input", "vars", "inputs"):
        f = getattr(L, "vpin_dev_instance_" + name)
        f.argtypes, f.restype = [vp], vp
    for name in ("num_cons_unpadded", "num_vars_unpadded", "comm_bytes", "proof_max_bytes"):
        f = getattr(L, "vpin_dev_instance_" + name)
        f.argtypes, f.restype = [vp], C.c_size_t
    L.vpin_dev_instance_nnz.argtypes = [vp, C.c_int]
    L.vpin_dev_instance_nnz.restype = C.c_size_t
    L.vpin_dev_instance_triplets.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.vpin_dev_instance_is_sat.argtypes = [vp, vp]
    L.vpin_spark_encode_dev.argtypes = [vp, vp, C.POINTER(vp), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.vpin_snark_prove_dev.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t), vp, vp]
    L.vpin_prof_enable.argtypes = [vp, C.c_int]
    L.vpin_prof_reset.argtypes = [vp]
    L.vpin_prof_read.argtypes = [vp, C.POINTER(KStat)]
    _lib = L
    return L


def host_register(arr):
    """page-lock a numpy array's buffer (vpin_host_register); returns a token for host_unregister"""
    L = lib()
    L.vpin_host_register.argtypes = [C.c_void_p, C.c_size_t]
    a = np.ascontiguousarray(arr)
    assert a.ctypes.data == arr.ctypes.data, "host_register needs a contiguous array (the buffer itself is pinned)"
    _chk(L.vpin_host_register(C.c_void_p(a.ctypes.data), a.nbytes), "vpin_host_register")
    return a.ctypes.data


def host_unregister(token):
    L = lib()
    L.vpin_host_unregister.argtypes = [C.c_void_p]
    _chk(L.vpin_host_unregister(C.c_void_p(token)), "vpin_host_unregister")


def declared_symbols():
    """Every function name declared in include/vpin_hip.h."""
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vpin_[a-z0-9_]+)\s*\(", text)))


def exported_symbols():
    L = lib()
    return [s for s in declared_symbols() if hasattr(L, s)]


def _chk(code, where):
    if code != 0:
        raise VpinError(code, where)


def _u8(a):
    a = np.ascontiguousarray(a)
    return a, a.ctypes.data_as(C.c_void_p)


class Table:
    """Device-resident Vec<Scalar> (DensePolynomial.Z of the reference)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def __len__(self):
        return int(lib().vpin_table_len(self.h))

    @property
    def device_ptr(self):
        return lib().vpin_table_device_ptr(self.h)

    def read(self, off=0, n=None):
        """-> (n,4) uint64 Montgomery limbs"""
        if n is None:
            n = len(self) - off
        out = np.zeros((n, 4), dtype=np.uint64)
        _chk(lib().vpin_table_read(self.ctx.h, self.h, off, n, out.ctypes.data_as(C.c_void_p)), "vpin_table_read")
        return out

    def write(self, off, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
        L = lib()
        L.vpin_table_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        _chk(L.vpin_table_write(self.ctx.h, self.h, off, a.shape[0], a.ctypes.data_as(C.c_void_p)), "vpin_table_write")

    def clone(self):
        h = C.c_void_p()
        _chk(lib().vpin_table_clone(self.ctx.h, self.h, C.byref(h)), "vpin_table_clone")
        return Table(self.ctx, h)

    def free(self):
        if self.h:
            lib().vpin_table_free(self.ctx.h, self.h)
            self.h = None


class SparkDecomm:
    """ComputationDecommitment resident in HBM (vpin_spark_encode)."""

    def __init__(self, ctx, h, proof_cap):
        self.ctx, self.h, self.proof_cap = ctx, h, proof_cap

    def hot_cols(self):
        """vpin_spark_decomm_hot_cols: per matrix the column taken out of the derefs commitment's table walks, or None"""
        out = (C.c_uint32 * 3)()
        L = lib()
        L.vpin_spark_decomm_hot_cols.restype = None
        L.vpin_spark_decomm_hot_cols.argtypes = [C.c_void_p, C.c_void_p]
        L.vpin_spark_decomm_hot_cols(self.h, out)
        return [None if v == 0xffffffff else int(v) for v in out]

    def free(self):
        if self.h:
            lib().vpin_spark_decomm_free(self.ctx.h, self.h)
            self.h = None


class R1csDev:
    """Device-resident (A,B,C): CSR + CSC copies in HBM."""

    def __init__(self, ctx, handle, num_cons, num_vars, num_inputs):
        self.ctx, self.h = ctx, handle
        self.num_cons, self.num_vars, self.num_inputs = num_cons, num_vars, num_inputs

    def free(self):
        if self.h:
            lib().vpin_r1cs_free(self.ctx.h, self.h)
            self.h = None


class DevInstance:
    """A gadget instance built on the device (vpin_gadget_point_*_dev): R1CS (CSR + CSC), the three assignments and
    the per-operation template, all resident in HBM."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        L = lib()
        nc, nv, ni = C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.vpin_r1cs_dims(L.vpin_dev_instance_r1cs(handle), C.byref(nc), C.byref(nv), C.byref(ni))
        self.num_cons, self.num_vars, self.num_inputs = nc.value, nv.value, ni.value
        self.num_cons_unpadded = L.vpin_dev_instance_num_cons_unpadded(handle)
        self.num_vars_unpadded = L.vpin_dev_instance_num_vars_unpadded(handle)
        self.nnz = [int(L.vpin_dev_instance_nnz(handle, m)) for m in range(3)]
        # borrowed views: freed with the instance
        self.r1cs = R1csDev(ctx, C.c_void_p(L.vpin_dev_instance_r1cs(handle)), self.num_cons, self.num_vars, self.num_inputs)
        self.vars_para = Table(ctx, C.c_void_p(L.vpin_dev_instance_vars_para(handle)))
        self.vars_input = Table(ctx, C.c_void_p(L.vpin_dev_instance_vars_input(handle)))
        self.vars = Table(ctx, C.c_void_p(L.vpin_dev_instance_vars(handle)))
        ip = L.vpin_dev_instance_inputs(handle)
        self.inputs = (np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_uint64)), shape=(self.num_inputs, 4)).copy()
                       if ip else np.zeros((0, 4), dtype=np.uint64))

    def triplets(self, m):
        n = lib().vpin_dev_instance_nnz(self.h, m)
        row, col, val = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros((n, 4), np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(lib().vpin_dev_instance_triplets(self.ctx.h, self.h, m, p(row), p(col), p(val)), "vpin_dev_instance_triplets")
        return row, col, val

    def is_sat(self):
        rc = lib().vpin_dev_instance_is_sat(self.ctx.h, self.h)
        if rc < 0:
            _chk(rc, "vpin_dev_instance_is_sat")
        return rc == 1

    def spark_encode(self):
        cap = lib().vpin_dev_instance_comm_bytes(self.h)
        comm = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        h = C.c_void_p()
        _chk(lib().vpin_spark_encode_dev(self.ctx.h, self.h, C.byref(h), comm.ctypes.data_as(C.c_void_p), cap, C.byref(n)),
             "vpin_spark_encode_dev")
        return SparkDecomm(self.ctx, h, lib().vpin_dev_instance_proof_max_bytes(self.h)), bytes(comm[:n.value])

    def snark_prove(self, seed_commit, seed_proof):
        """SNARK::encode + my_lib_prove: dict(proof, comm, comm_para, comm_input)"""
        L = lib()
        cap, ccap = L.vpin_dev_instance_proof_max_bytes(self.h), L.vpin_dev_instance_comm_bytes(self.h)
        Ls = 1 << ((self.num_vars.bit_length() - 1) // 2)
        proof, comm = np.zeros(cap, np.uint8), np.zeros(ccap, np.uint8)
        cp, ci = np.zeros((Ls, 32), np.uint8), np.zeros((Ls, 32), np.uint8)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
        n, cn = C.c_size_t(0), C.c_size_t(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(L.vpin_snark_prove_dev(self.ctx.h, self.h, p(sc), p(sp), p(proof), cap, C.byref(n), p(comm), ccap, C.byref(cn), p(cp), p(ci)),
             "vpin_snark_prove_dev")
        return dict(proof=bytes(proof[:n.value]), comm=bytes(comm[:cn.value]), comm_para=cp, comm_input=ci)

    def free(self):
        if self.h:
            lib().vpin_dev_instance_free(self.ctx.h, self.h)
            self.h = None


class Gens:
    """Device window table over a Pedersen generator stream (MultiCommitGens)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def __len__(self):
        return int(lib().vpin_gens_count(self.h))

    def free(self):
        if self.h:
            lib().vpin_gens_free(self.ctx.h, self.h)
            self.h = None


class CommStats(C.Structure):
    _fields_ = [("collectives", C.c_uint64), ("bytes", C.c_double), ("wait_s", C.c_double), ("busy_s", C.c_double),
                ("crit_s", C.c_double)]


_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def _comm_decl():
    L = lib()
    if getattr(L, "_comm_declared", False):
        return L
    vp = C.c_void_p
    L.vpin_comm_create_shm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(vp)]
    L.vpin_comm_create_local.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.vpin_comm_create_callbacks.argtypes = [C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
    L.vpin_comm_destroy.argtypes = [vp]
    L.vpin_comm_destroy.restype = None
    L.vpin_comm_rank.argtypes = [vp]
    L.vpin_comm_world.argtypes = [vp]
    L.vpin_comm_allgather.argtypes = [vp, vp, vp, C.c_size_t]
    L.vpin_comm_allgather_dev.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.vpin_comm_enable_rccl.argtypes = [vp, vp]
    L.vpin_comm_set_serialize.argtypes = [vp, C.c_int]
    L.vpin_comm_stats_read.argtypes = [vp, C.POINTER(CommStats), C.c_int]
    L.vpin_comm_stats_tags.argtypes = [vp, vp, C.c_size_t]
    L.vpin_comm_stats_tags.restype = C.c_size_t
    L.vpin_dist_plan.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L._comm_declared = True
    return L


class Comm:
    """vpin_comm: the exchange layer of one proof over several GPUs (SPMD; see include/vpin_hip.h)."""

    def __init__(self, handle, keep=None):
        self.h, self._keep = handle, keep
        L = _comm_decl()
        self.rank, self.world = L.vpin_comm_rank(handle), L.vpin_comm_world(handle)

    @staticmethod
    def shm(name, rank, world, slot_bytes=0):
        """ranks = processes of one node; `name` ("/...") unique per job, the same on every rank"""
        h = C.c_void_p()
        _chk(_comm_decl().vpin_comm_create_shm(name.encode(), rank, world, slot_bytes, C.byref(h)), "vpin_comm_create_shm")
        return Comm(h)

    @staticmethod
    def local(world, slot_bytes=0):
        """`world` handles for the threads of this process"""
        hs = (C.c_void_p * world)()
        _chk(_comm_decl().vpin_comm_create_local(world, slot_bytes, hs), "vpin_comm_create_local")
        return [Comm(C.c_void_p(hs[r])) for r in range(world)]

    @staticmethod
    def callbacks(rank, world, allgather):
        """allgather(send: bytes) -> bytes of world x len(send): the caller's fabric (e.g. torch.distributed over gloo)"""
        err = []

        def _fn(_user, send, recv, n):
            try:
                out = allgather(C.string_at(send, n))
                assert len(out) == n * world
                C.memmove(recv, out, n * world)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                err.append(e)
                return -7

        cb = _ALLGATHER_FN(_fn)
        h = C.c_void_p()
        _chk(_comm_decl().vpin_comm_create_callbacks(rank, world, C.cast(cb, C.c_void_p), None, C.byref(h)),
             "vpin_comm_create_callbacks")
        cm = Comm(h, keep=(cb, err))
        cm.errors = err
        return cm

    def allgather(self, data):
        data = bytes(data)
        out = C.create_string_buffer(len(data) * self.world)
        _chk(_comm_decl().vpin_comm_allgather(self.h, data, out, len(data)), "vpin_comm_allgather")
        return out.raw

    def allgather_dev(self, ctx, d_send, d_recv, nbytes):
        _chk(_comm_decl().vpin_comm_allgather_dev(self.h, ctx.h, d_send, d_recv, nbytes), "vpin_comm_allgather_dev")

    def enable_rccl(self, ctx):
        _chk(_comm_decl().vpin_comm_enable_rccl(self.h, ctx.h), "vpin_comm_enable_rccl")

    def set_serialize(self, on=True):
        _chk(_comm_decl().vpin_comm_set_serialize(self.h, 1 if on else 0), "vpin_comm_set_serialize")

    def stats(self, reset=False):
        st = CommStats()
        _chk(_comm_decl().vpin_comm_stats_read(self.h, C.byref(st), 1 if reset else 0), "vpin_comm_stats_read")
        return dict(collectives=int(st.collectives), bytes=st.bytes, wait_s=st.wait_s, busy_s=st.busy_s, crit_s=st.crit_s)

    def latency(self, nbytes=1728, iters=2000):
        """seconds per all-gather, measured inside the library (collective)"""
        L = _comm_decl()
        L.vpin_comm_latency.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        out = C.c_double(0.0)
        _chk(L.vpin_comm_latency(self.h, nbytes, iters, C.byref(out)), "vpin_comm_latency")
        return out.value

    def tag_stats(self):
        """{tag: dict(collectives, busy_s, crit_s)}: the critical path per step of the protocol (serialized rehearsal)"""
        L = _comm_decl()
        n = L.vpin_comm_stats_tags(self.h, None, 0)
        buf = C.create_string_buffer(n + 16)
        L.vpin_comm_stats_tags(self.h, buf, n + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            tag, cnt, busy, crit = line.split()
            out[tag] = dict(collectives=int(cnt), busy_s=float(busy), crit_s=float(crit))
        return out

    def abort(self):
        """marks the group dead: every peer's current or next wait fails with VPIN_ECOMM (vpin_comm_abort)"""
        L = _comm_decl()
        L.vpin_comm_abort.argtypes = [C.c_void_p]
        L.vpin_comm_abort.restype = None
        L.vpin_comm_abort(self.h)

    def destroy(self):
        if self.h:
            _comm_decl().vpin_comm_destroy(self.h)
            self.h = None


def gadget_shape(kind, n_ops):
    """(num_cons, num_vars, [nnz A, B, C]) of the instance a point-mult ("mult") / point-add gadget builds for n_ops
    operations (vpin_gadget_shape)"""
    L = lib()
    L.vpin_gadget_shape.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    nc, nv, nnz = C.c_size_t(), C.c_size_t(), (C.c_size_t * 3)()
    _chk(L.vpin_gadget_shape(1 if kind == "mult" else 0, n_ops, C.byref(nc), C.byref(nv), nnz), "vpin_gadget_shape")
    return nc.value, nv.value, [nnz[0], nnz[1], nnz[2]]


def dist_plan(world):
    """owners of the 12 ops circuits, the 6 dot-product halves and the 4 mem circuits (vpin_dist_plan)"""
    a, b, c = (C.c_int * 12)(), (C.c_int * 6)(), (C.c_int * 4)()
    _chk(_comm_decl().vpin_dist_plan(world, a, b, c), "vpin_dist_plan")
    return list(a), list(b), list(c)


class Context:
    def __init__(self, device=0, priority=0, cu_mask=None):
        """priority < 0: high-priority stream (latency-bound small proofs beside a large one); > 0: low.
        cu_mask: iterable of enabled CU numbers (the runtime's numbering, see vpin_ctx_create_cumask): the stream is
        confined to them (normal priority)."""
        self.h = C.c_void_p()
        if cu_mask is not None:
            words = self._mask_words(cu_mask)
            _chk(lib().vpin_ctx_create_cumask(device, words.ctypes.data_as(C.c_void_p), len(words), C.byref(self.h)), "vpin_ctx_create_cumask")
            return
        _chk(lib().vpin_ctx_create_prio(device, priority, C.byref(self.h)), "vpin_ctx_create_prio")

    @staticmethod
    def _mask_words(cu_mask):
        cus = sorted(set(int(x) for x in cu_mask))
        words = np.zeros((cus[-1] // 32) + 1, dtype=np.uint32)
        for x in cus:
            words[x // 32] |= np.uint32(1 << (x % 32))
        return words

    def set_cumask_after_phase1(self, cu_mask):
        """second stream confined to the CUs of cu_mask; proofs move to it after their phase-1 sum-check (None: remove it)"""
        if cu_mask is None:
            _chk(lib().vpin_ctx_set_cumask_after_phase1(self.h, None, 0), "vpin_ctx_set_cumask_after_phase1")
            return
        words = self._mask_words(cu_mask)
        _chk(lib().vpin_ctx_set_cumask_after_phase1(self.h, words.ctypes.data_as(C.c_void_p), len(words)), "vpin_ctx_set_cumask_after_phase1")

    def close(self):
        if self.h:
            lib().vpin_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def stream(self):
        return lib().vpin_ctx_stream(self.h)

    def sync(self):
        _chk(lib().vpin_ctx_sync(self.h), "vpin_ctx_sync")

    def device_props(self):
        """(compute units, shader clock in Hz)"""
        cus, khz = C.c_int(), C.c_int()
        L = lib()
        L.vpin_ctx_device_props.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _chk(L.vpin_ctx_device_props(self.h, C.byref(cus), C.byref(khz)), "vpin_ctx_device_props")
        return cus.value, khz.value * 1e3

    def device_total_bytes(self):
        f, t = C.c_size_t(), C.c_size_t()
        L = lib()
        L.vpin_ctx_mem_info.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _chk(L.vpin_ctx_mem_info(self.h, C.byref(f), C.byref(t)), "vpin_ctx_mem_info")
        return t.value

    def pool_stats(self):
        """(bytes held from the driver, bytes of them cached in the free lists, blocks) of this context's device pool"""
        L = lib()
        L.vpin_ctx_pool_stats.argtypes = [C.c_void_p, C.POINTER(C.c_size_t * 3)]
        out = (C.c_size_t * 3)()
        _chk(L.vpin_ctx_pool_stats(self.h, C.byref(out)), "vpin_ctx_pool_stats")
        return int(out[0]), int(out[1]), int(out[2])

    @staticmethod
    def crash_line_set(line_bytes):
        """leave `line_bytes` to be written to stdout should a fatal signal or SIGTERM end the process (vpin_crash_line_set);
        b"" disarms"""
        L = lib()
        L.vpin_crash_line_set.argtypes = [C.c_char_p, C.c_size_t]
        _chk(L.vpin_crash_line_set(line_bytes if line_bytes else None, len(line_bytes)), "vpin_crash_line_set")

    @staticmethod
    def driver_alloc_stats():
        """(calls, bytes) the library has taken from the driver's allocator since the process started (vpin_driver_alloc_stats)"""
        L = lib()
        L.vpin_driver_alloc_stats.argtypes = [C.POINTER(C.c_ulonglong * 2)]
        out = (C.c_ulonglong * 2)()
        _chk(L.vpin_driver_alloc_stats(C.byref(out)), "vpin_driver_alloc_stats")
        return int(out[0]), int(out[1])

    def pool_trim(self):
        """cached blocks of this context's device pool back to the driver (vpin_ctx_pool_trim)"""
        L = lib()
        L.vpin_ctx_pool_trim.argtypes = [C.c_void_p]
        _chk(L.vpin_ctx_pool_trim(self.h), "vpin_ctx_pool_trim")

    def pip_row_chunks(self):
        """row chunks the bucket method has launched on this context (vpin_ctx_pip_row_chunks)"""
        L = lib()
        L.vpin_ctx_pip_row_chunks.argtypes = [C.c_void_p]
        L.vpin_ctx_pip_row_chunks.restype = C.c_ulonglong
        return int(L.vpin_ctx_pip_row_chunks(self.h))

    def strip_rows_taken(self):
        """rows handed to the row-per-lane commitment kernel so far (vpin_ctx_strip_rows_taken)"""
        L = lib()
        L.vpin_ctx_strip_rows_taken.argtypes = [C.c_void_p]
        L.vpin_ctx_strip_rows_taken.restype = C.c_ulonglong
        return int(L.vpin_ctx_strip_rows_taken(self.h))

    def set_shared_device(self, on=True):
        """other contexts prove on this device concurrently: leave them a share of every CU"""
        _chk(lib().vpin_ctx_set_shared_device(self.h, 1 if on else 0), "vpin_ctx_set_shared_device")

    def set_low_memory(self, on=True):
        """vpin_ctx_set_low_memory: a third less working set for ~1 % of a large proof's time"""
        L = lib()
        L.vpin_ctx_set_low_memory.argtypes = [C.c_void_p, C.c_int]
        _chk(L.vpin_ctx_set_low_memory(self.h, 1 if on else 0), "vpin_ctx_set_low_memory")

    def set_expected_proofs(self, n):
        """generator window tables built through this context will serve n proofs (0 = a service: widest windows)"""
        L = lib()
        L.vpin_ctx_set_expected_proofs.argtypes = [C.c_void_p, C.c_int]
        _chk(L.vpin_ctx_set_expected_proofs(self.h, int(n)), "vpin_ctx_set_expected_proofs")

    def gens_map_stream(self, stream):
        """RistrettoPoint::from_uniform_bytes on the device: 64 bytes of a label's SHAKE256 stream per generator ->
        (n, 128) uint8 X|Y|Z|T (vpin_gens_map_stream)"""
        L = lib()
        L.vpin_gens_map_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
        n = buf.size // 64
        out = np.zeros((n, 128), dtype=np.uint8)
        _chk(L.vpin_gens_map_stream(self.h, buf.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)), "vpin_gens_map_stream")
        return out

    # ---- tables ----
    def upload(self, arr):
        """arr: (n,4) uint64 Montgomery limbs (or n*32 bytes)."""
        a = np.ascontiguousarray(arr)
        n = a.nbytes // 32
        h = C.c_void_p()
        _chk(lib().vpin_table_upload(self.h, a.ctypes.data_as(C.c_void_p), n, C.byref(h)), "vpin_table_upload")
        return Table(self, h)

    def alloc(self, n):
        h = C.c_void_p()
        _chk(lib().vpin_table_alloc(self.h, n, C.byref(h)), "vpin_table_alloc")
        return Table(self, h)

    def wrap(self, device_ptr, n):
        h = C.c_void_p()
        _chk(lib().vpin_table_wrap(self.h, C.c_void_p(device_ptr), n, C.byref(h)), "vpin_table_wrap")
        return Table(self, h)

    def eq_table(self, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        _chk(lib().vpin_eq_table(self.h, r.ctypes.data_as(C.c_void_p), r.shape[0], C.byref(h)), "vpin_eq_table")
        return Table(self, h)

    # ---- sum-check ----
    def sc_cubic_round(self, tau, A, B, Cc):
        out = np.zeros((3, 4), dtype=np.uint64)
        _chk(lib().vpin_sc_cubic_round(self.h, tau.h, A.h, B.h, Cc.h, out.ctypes.data_as(C.c_void_p)),
             "vpin_sc_cubic_round")
        return out

    def sc_quad_round(self, A, B):
        out = np.zeros((2, 4), dtype=np.uint64)
        _chk(lib().vpin_sc_quad_round(self.h, A.h, B.h, out.ctypes.data_as(C.c_void_p)), "vpin_sc_quad_round")
        return out

    def sc_bind(self, tables, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        arr = (C.c_void_p * len(tables))(*[t.h for t in tables])
        _chk(lib().vpin_sc_bind(self.h, arr, len(tables), r.ctypes.data_as(C.c_void_p)), "vpin_sc_bind")

    def sc_cubic_bind_round(self, tau, A, B, Cc, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        out = np.zeros((3, 4), dtype=np.uint64)
        _chk(lib().vpin_sc_cubic_bind_round(self.h, tau.h, A.h, B.h, Cc.h, r.ctypes.data_as(C.c_void_p),
                                            out.ctypes.data_as(C.c_void_p)), "vpin_sc_cubic_bind_round")
        return out

    def sc_quad_bind_round(self, A, B, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        out = np.zeros((2, 4), dtype=np.uint64)
        _chk(lib().vpin_sc_quad_bind_round(self.h, A.h, B.h, r.ctypes.data_as(C.c_void_p),
                                           out.ctypes.data_as(C.c_void_p)), "vpin_sc_quad_bind_round")
        return out

    # ---- generators / MSM ----
    def gens_create(self, xyzt):
        """xyzt: (nb,128) uint8 generator stream (X|Y|Z|T canonical LE)."""
        a = np.ascontiguousarray(xyzt, dtype=np.uint8).reshape(-1, 128)
        h = C.c_void_p()
        _chk(lib().vpin_gens_create(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], C.byref(h)), "vpin_gens_create")
        return Gens(self, h)

    def gens_shared(self, label, xyzt, budget_gb=0):
        """vpin_gens_shared: the process-wide window table of `label`'s generator stream (a non-owning handle)."""
        a = np.ascontiguousarray(xyzt, dtype=np.uint8).reshape(-1, 128)
        h = C.c_void_p()
        _chk(lib().vpin_gens_shared(self.h, label.encode(), a.ctypes.data_as(C.c_void_p), a.shape[0], budget_gb, C.byref(h)),
             "vpin_gens_shared")
        g = Gens(self, h)
        g.free = lambda: None  # owned by the registry
        return g

    def gens_msm_parts(self, gens, scalars, rows, ncols):
        """vpin_gens_msm_parts: (rows, parts, 128) uint8 partial points X|Y|Z|T; the caller adds them."""
        L = lib()
        L.vpin_gens_msm_parts_count.restype = C.c_size_t
        L.vpin_gens_msm_parts_count.argtypes = [C.c_size_t]
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(rows * ncols, 4)
        np_ = L.vpin_gens_msm_parts_count(ncols)
        out = np.zeros((rows, np_, 128), dtype=np.uint8)
        L.vpin_gens_msm_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        _chk(L.vpin_gens_msm_parts(self.h, gens.h, s.ctypes.data_as(C.c_void_p), rows, ncols, out.ctypes.data_as(C.c_void_p)),
             "vpin_gens_msm_parts")
        return out

    def hyrax_commit(self, gens, Z, blinds, blind_base):
        b = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(-1, 4)
        Ls = b.shape[0]
        out = np.zeros((Ls, 32), dtype=np.uint8)
        _chk(lib().vpin_hyrax_commit(self.h, gens.h, Z.h, b.ctypes.data_as(C.c_void_p), Ls, blind_base,
                                     out.ctypes.data_as(C.c_void_p)), "vpin_hyrax_commit")
        return out

    def hyrax_commit_pippenger(self, gens, Z, blinds, blind_base, Ls=None, c_bits=0):
        """vpin_hyrax_commit_pippenger: the same rows by bucket accumulation (blinds None: commit(gens, None) of Ls rows)"""
        b = None if blinds is None else np.ascontiguousarray(blinds, dtype=np.uint64).reshape(-1, 4)
        Ls = b.shape[0] if b is not None else int(Ls)
        out = np.zeros((Ls, 32), dtype=np.uint8)
        _chk(lib().vpin_hyrax_commit_pippenger(self.h, gens.h, Z.h, b.ctypes.data_as(C.c_void_p) if b is not None else None, Ls,
                                               blind_base, c_bits, out.ctypes.data_as(C.c_void_p)), "vpin_hyrax_commit_pippenger")
        return out

    def hyrax_commit_pair(self, gens, Za, Zb, blinds_a, blinds_b, blind_base):
        ba = np.ascontiguousarray(blinds_a, dtype=np.uint64).reshape(-1, 4)
        bb = np.ascontiguousarray(blinds_b, dtype=np.uint64).reshape(-1, 4)
        Ls = ba.shape[0]
        outs = [np.zeros((Ls, 32), dtype=np.uint8) for _ in range(3)]
        _chk(lib().vpin_hyrax_commit_pair(self.h, gens.h, Za.h, Zb.h, ba.ctypes.data_as(C.c_void_p),
                                          bb.ctypes.data_as(C.c_void_p), Ls, blind_base,
                                          *[o.ctypes.data_as(C.c_void_p) for o in outs]), "vpin_hyrax_commit_pair")
        return outs

    def gens_msm(self, gens, scalars, rows, ncols, want_xyzt=False):
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(rows * ncols, 4)
        out = np.zeros((rows, 32), dtype=np.uint8)
        ox = np.zeros((rows, 128), dtype=np.uint8) if want_xyzt else None
        _chk(lib().vpin_gens_msm(self.h, gens.h, s.ctypes.data_as(C.c_void_p), rows, ncols,
                                 out.ctypes.data_as(C.c_void_p),
                                 ox.ctypes.data_as(C.c_void_p) if want_xyzt else None), "vpin_gens_msm")
        return (out, ox) if want_xyzt else out

    def msm(self, scalars, points_compressed, want_xyzt=False):
        """vpin_msm: sum_i s_i * decompress(P_i); scalars (n,4) uint64 Montgomery, points (n,32) uint8"""
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        pts = np.ascontiguousarray(points_compressed, dtype=np.uint8).reshape(-1, 32)
        assert s.shape[0] == pts.shape[0]
        out, xyzt = np.zeros(32, np.uint8), np.zeros(128, np.uint8)
        L = lib()
        L.vpin_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(L.vpin_msm(self.h, p(s), p(pts), s.shape[0], p(out), p(xyzt) if want_xyzt else None), "vpin_msm")
        return (out, xyzt) if want_xyzt else out

    def dense_mlpoly_commit_sum(self, t_vars, seed_commit):
        """vpin_dense_mlpoly_commit_sum: my_dense_mlpoly_commit of the whole assignment (the reference's third commitment,
        proof_point_mult.rs:58-59) -> (L, 32) uint8"""
        L = lib()
        L.vpin_dense_mlpoly_commit_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        nv = len(t_vars)
        Ls = 1 << ((nv.bit_length() - 1) // 2)
        out = np.zeros((Ls, 32), dtype=np.uint8)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        _chk(L.vpin_dense_mlpoly_commit_sum(self.h, t_vars.h, sc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
             "vpin_dense_mlpoly_commit_sum")
        return out

    def points_add(self, a, b):
        """vpin_points_add: compress(decompress(a[i]) + decompress(b[i]))"""
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros_like(a)
        L = lib()
        L.vpin_points_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        _chk(L.vpin_points_add(self.h, p(a), p(b), a.shape[0], p(out)), "vpin_points_add")
        return out

    def poly_bound(self, Z, Lvec):
        lv = np.ascontiguousarray(Lvec, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((len(Z) // lv.shape[0], 4), dtype=np.uint64)
        _chk(lib().vpin_poly_bound(self.h, Z.h, lv.ctypes.data_as(C.c_void_p), lv.shape[0],
                                   out.ctypes.data_as(C.c_void_p)), "vpin_poly_bound")
        return out

    def poly_slices_bound(self, Z, nbits, used, r, ch=None):
        """vpin_poly_slices_bound: (evaluations of the first `used` of Z's 2^nbits slices at r, LZ of the whole Z at (ch, r) or None)"""
        rr = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        ell = rr.shape[0] + nbits
        ev = np.zeros((used, 4), dtype=np.uint64)
        lz = np.zeros((1 << (ell - ell // 2), 4), dtype=np.uint64) if ch is not None else None
        cc = np.ascontiguousarray(ch, dtype=np.uint64).reshape(-1, 4) if ch is not None else None
        L = lib()
        L.vpin_poly_slices_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.vpin_poly_slices_bound(self.h, Z.h, nbits, used, rr.ctypes.data_as(C.c_void_p), rr.shape[0],
                                      cc.ctypes.data_as(C.c_void_p) if cc is not None else None, ev.ctypes.data_as(C.c_void_p),
                                      lz.ctypes.data_as(C.c_void_p) if lz is not None else None), "vpin_poly_slices_bound")
        return ev, lz

    def poly_slices_bound_u32(self, u32, Zfq, nbits, used, r, ch=None):
        """vpin_poly_slices_bound_u32: u32 = (n32, N) uint32 slices, Zfq = Table of the other used - n32 slices (or None)"""
        u = np.ascontiguousarray(u32, dtype=np.uint32)
        rr = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        ell = rr.shape[0] + nbits
        ev = np.zeros((used, 4), dtype=np.uint64)
        lz = np.zeros((1 << (ell - ell // 2), 4), dtype=np.uint64) if ch is not None else None
        cc = np.ascontiguousarray(ch, dtype=np.uint64).reshape(-1, 4) if ch is not None else None
        L = lib()
        L.vpin_poly_slices_bound_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(L.vpin_poly_slices_bound_u32(self.h, u.ctypes.data_as(C.c_void_p), u.shape[0], Zfq.h if Zfq is not None else None, nbits, used,
                                          rr.ctypes.data_as(C.c_void_p), rr.shape[0], cc.ctypes.data_as(C.c_void_p) if cc is not None else None,
                                          ev.ctypes.data_as(C.c_void_p), lz.ctypes.data_as(C.c_void_p) if lz is not None else None),
             "vpin_poly_slices_bound_u32")
        return ev, lz

    # ---- sat proof ----
    def sat_prove(self, inst, seed_commit, seed_proof):
        """inst: dict as produced by the gadget builders (padded instance + three assignments)."""
        r = make_r1cs(inst)
        nv, nc = inst["num_vars"], inst["num_cons"]
        ell = nv.bit_length() - 1
        Ls = 1 << (ell // 2)
        cap = lib().vpin_sat_proof_max_bytes(nc, nv)
        proof = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        cp = np.zeros((Ls, 32), dtype=np.uint8)
        ci = np.zeros((Ls, 32), dtype=np.uint8)
        ev = np.zeros((3, 4), dtype=np.uint64)
        rx = np.zeros((nc.bit_length() - 1, 4), dtype=np.uint64)
        ry = np.zeros((ell + 1, 4), dtype=np.uint64)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
        p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        keep = [np.ascontiguousarray(inst[k]) for k in ("vars_para", "vars_input", "vars", "inputs")]
        _chk(lib().vpin_sat_prove(self.h, C.byref(r), p(keep[0]), p(keep[1]), p(keep[2]),
                                  p(keep[3]) if keep[3].size else None, p(sc), p(sp), p(proof), cap, C.byref(n),
                                  p(cp), p(ci), p(ev), p(rx), p(ry)), "vpin_sat_prove")
        return dict(proof=bytes(proof[:n.value]), comm_para=cp, comm_input=ci, inst_evals=ev, rx=rx, ry=ry)

    def r1cs_upload(self, inst):
        r = make_r1cs(inst)
        h = C.c_void_p()
        _chk(lib().vpin_r1cs_upload(self.h, C.byref(r), C.byref(h)), "vpin_r1cs_upload")
        return R1csDev(self, h, inst["num_cons"], inst["num_vars"], inst["num_inputs"])

    def r1cs_build_z(self, dinst, vars_t, inputs):
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        h = C.c_void_p()
        _chk(lib().vpin_r1cs_build_z(self.h, dinst.h, vars_t.h, inp.ctypes.data_as(C.c_void_p) if inp.size else None,
                                     C.byref(h)), "vpin_r1cs_build_z")
        return Table(self, h)

    def r1cs_multiply_vec(self, dinst, z):
        hs = [C.c_void_p() for _ in range(3)]
        _chk(lib().vpin_r1cs_multiply_vec(self.h, dinst.h, z.h, *[C.byref(h) for h in hs]), "vpin_r1cs_multiply_vec")
        return [Table(self, h) for h in hs]

    def r1cs_eval_table(self, dinst, evals_rx, r_abc):
        r = np.ascontiguousarray(r_abc, dtype=np.uint64).reshape(3, 4)
        h = C.c_void_p()
        _chk(lib().vpin_r1cs_eval_table(self.h, dinst.h, evals_rx.h, r.ctypes.data_as(C.c_void_p), C.byref(h)),
             "vpin_r1cs_eval_table")
        return Table(self, h)

    def r1cs_evaluate(self, dinst, evals_rx, evals_ry):
        out = np.zeros((3, 4), dtype=np.uint64)
        _chk(lib().vpin_r1cs_evaluate(self.h, dinst.h, evals_rx.h, evals_ry.h, out.ctypes.data_as(C.c_void_p)),
             "vpin_r1cs_evaluate")
        return out

    def sat_prove_resident(self, dinst, t_para, t_input, t_vars, inputs, seed_commit, seed_proof):
        nv, nc = dinst.num_vars, dinst.num_cons
        ell = nv.bit_length() - 1
        Ls = 1 << (ell // 2)
        cap = lib().vpin_sat_proof_max_bytes(nc, nv)
        proof = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        cp = np.zeros((Ls, 32), dtype=np.uint8)
        ci = np.zeros((Ls, 32), dtype=np.uint8)
        ev = np.zeros((3, 4), dtype=np.uint64)
        rx = np.zeros((nc.bit_length() - 1, 4), dtype=np.uint64)
        ry = np.zeros((ell + 1, 4), dtype=np.uint64)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(lib().vpin_sat_prove_resident(self.h, dinst.h, t_para.h, t_input.h, t_vars.h, p(inp) if inp.size else None,
                                           p(sc), p(sp), p(proof), cap, C.byref(n), p(cp), p(ci), p(ev), p(rx), p(ry)),
             "vpin_sat_prove_resident")
        return dict(proof=bytes(proof[:n.value]), comm_para=cp, comm_input=ci, inst_evals=ev, rx=rx, ry=ry)

    def set_progress_flag(self, arr):
        """arr: np.int32 array of length >= 1 kept alive by the caller, or None."""
        _chk(lib().vpin_ctx_set_progress_flag(self.h, arr.ctypes.data_as(C.c_void_p) if arr is not None else None),
             "vpin_ctx_set_progress_flag")

    def sat_prepare(self, num_vars):
        _chk(lib().vpin_sat_prepare(self.h, num_vars), "vpin_sat_prepare")

    # ---- one proof over several GPUs (include/vpin_hip.h: vpin_comm) ----
    def set_comm(self, comm):
        """attach a Comm (or None): the prove calls on this context become collective calls over its ranks"""
        Lb = lib()
        Lb.vpin_ctx_set_comm.argtypes = [C.c_void_p, C.c_void_p]
        _chk(Lb.vpin_ctx_set_comm(self.h, comm.h if comm is not None else None), "vpin_ctx_set_comm")
        self._comm_keep = comm

    def spark_prepare(self, num_cons, num_vars, max_nnz):
        L = lib()
        L.vpin_spark_prepare.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        _chk(L.vpin_spark_prepare(self.h, num_cons, num_vars, max_nnz), "vpin_spark_prepare")

    # ---- SPARK / whole SNARK ----
    def spark_encode(self, inst):
        """SNARK::encode: returns (SparkDecomm, bincode(R1CSCommitment) bytes)."""
        r = make_r1cs(inst)
        cap = lib().vpin_spark_comm_bytes(C.byref(r))
        comm = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        h = C.c_void_p()
        _chk(lib().vpin_spark_encode(self.h, C.byref(r), C.byref(h), comm.ctypes.data_as(C.c_void_p), cap, C.byref(n)),
             "vpin_spark_encode")
        return SparkDecomm(self, h, lib().vpin_snark_proof_max_bytes(C.byref(r))), bytes(comm[:n.value])

    def snark_prove_resident(self, dinst, decomm, t_para, t_input, t_vars, inputs, seed_commit, seed_proof):
        nv = dinst.num_vars
        Ls = 1 << ((nv.bit_length() - 1) // 2)
        cap = decomm.proof_cap
        proof = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        cp = np.zeros((Ls, 32), dtype=np.uint8)
        ci = np.zeros((Ls, 32), dtype=np.uint8)
        sc = np.frombuffer(bytes(seed_commit), dtype=np.uint8).copy()
        sp = np.frombuffer(bytes(seed_proof), dtype=np.uint8).copy()
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(lib().vpin_snark_prove_resident(self.h, dinst.h, decomm.h, t_para.h, t_input.h, t_vars.h,
                                             p(inp) if inp.size else None, p(sc), p(sp), p(proof), cap, C.byref(n),
                                             p(cp), p(ci)), "vpin_snark_prove_resident")
        return dict(proof=bytes(proof[:n.value]), comm_para=cp, comm_input=ci)

    def snark_prove(self, inst, seed_commit, seed_proof):
        """encode + prove from host buffers (test convenience): dict(proof, comm, comm_para, comm_input)."""
        dinst = self.r1cs_upload(inst)
        decomm, comm = self.spark_encode(inst)
        tabs = [self.upload(inst[k]) for k in ("vars_para", "vars_input", "vars")]
        try:
            res = self.snark_prove_resident(dinst, decomm, tabs[0], tabs[1], tabs[2], inst["inputs"], seed_commit, seed_proof)
        finally:
            for t in tabs:
                t.free()
            decomm.free()
            dinst.free()
        res["comm"] = comm
        return res

    def snark_verify(self, inst, res, proof=None, comm=None):
        """my_lib_verify: True = accept, False = rejected by the verifier."""
        pb = np.frombuffer(proof if proof is not None else res["proof"], dtype=np.uint8).copy()
        cb = np.frombuffer(comm if comm is not None else res["comm"], dtype=np.uint8).copy()
        inp = np.ascontiguousarray(inst["inputs"], dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().vpin_snark_verify(self.h, p(pb), len(pb), p(cb), len(cb), p(inp) if inp.size else None, inst["num_inputs"],
                                     p(np.ascontiguousarray(res["comm_para"])), p(np.ascontiguousarray(res["comm_input"])))
        if rc not in (0, -6):
            _chk(rc, "vpin_snark_verify")
        return rc == 0

    def sat_verify(self, inst, res, proof=None):
        pb = np.frombuffer(proof if proof is not None else res["proof"], dtype=np.uint8).copy()
        inp = np.ascontiguousarray(inst["inputs"], dtype=np.uint64)
        ev = np.ascontiguousarray(res["inst_evals"], dtype=np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().vpin_sat_verify(self.h, p(pb), len(pb), inst["num_cons"], inst["num_vars"], p(inp) if inp.size else None,
                                   inst["num_inputs"], p(ev), p(np.ascontiguousarray(res["comm_para"])),
                                   p(np.ascontiguousarray(res["comm_input"])))
        if rc not in (0, -6):
            _chk(rc, "vpin_sat_verify")
        return rc == 0

    @staticmethod
    def spark_timings():
        out = (C.c_double * 8)()
        lib().vpin_spark_last_timings(out)
        names = ("encode", "derefs_commit", "network_build", "product_layer", "hash_layer", "sat", "total", "_")
        return dict(zip(names, out))

    @staticmethod
    def sat_timings():
        out = (C.c_double * 8)()
        lib().vpin_sat_last_timings(out)
        names = ("polycommit", "sc_phase_one", "sc_phase_two", "polyeval", "total", "gens", "host_spmv", "inst_evaluate")
        return dict(zip(names, out))

    # ---- gadgets on the device ----
    def gadget_point_add_dev(self, px, py, rx, ry, rz):
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (px, py, rx, ry, rz)]
        h = C.c_void_p()
        _chk(lib().vpin_gadget_point_add_dev(self.h, *[a.ctypes.data_as(C.c_void_p) for a in arrs], len(arrs[4]), C.byref(h)),
             "vpin_gadget_point_add_dev")
        return DevInstance(self, h)

    def gadget_point_mult_dev(self, weights, px, py):
        """weights: Python ints < 2^128"""
        w = np.frombuffer(b"".join(int(v).to_bytes(16, "little") for v in weights), dtype=np.uint8).copy()
        x, y = np.ascontiguousarray(px, dtype=np.uint8), np.ascontiguousarray(py, dtype=np.uint8)
        h = C.c_void_p()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _chk(lib().vpin_gadget_point_mult_dev(self.h, p(w), p(x), p(y), len(weights), C.byref(h)), "vpin_gadget_point_mult_dev")
        return DevInstance(self, h)

    # ---- profiling ----
    def prof_enable(self, on=True):
        """on = 2: also count the table additions of the row commitments (vpin_prof_enable level 2)"""
        _chk(lib().vpin_prof_enable(self.h, int(on)), "vpin_prof_enable")

    def prof_reset(self):
        _chk(lib().vpin_prof_reset(self.h), "vpin_prof_reset")

    def prof_read(self):
        arr = (KStat * K_COUNT)()
        _chk(lib().vpin_prof_read(self.h, arr), "vpin_prof_read")
        out = {}
        for k, name in KERNEL_CLASSES.items():
            if arr[k].launches:
                out[name] = {"launches": int(arr[k].launches), "ms": float(arr[k].ms),
                             "alg_bytes": float(arr[k].alg_bytes), "units": float(arr[k].units)}
        return out
