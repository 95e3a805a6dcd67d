"""ctypes binding of the gadget builders / synthetic witness generator of include/vpin_hip.h
(vPIN_proof_generation/src/point_addition.rs, point_mult.rs; Spartan/src/lib.rs Instance::new)."""
import ctypes as C

import numpy as np

from .capi import R1CS, VpinError, lib


def _decl():
    L = lib()
    vp = C.c_void_p
    if getattr(L, "_gadgets_declared", False):
        return L
    L.vpin_gadget_point_add.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_gadget_point_mult.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(vp)]
    L.vpin_instance_free.argtypes = [vp]
    L.vpin_instance_free.restype = None
    L.vpin_instance_r1cs.argtypes = [vp]
    L.vpin_instance_r1cs.restype = C.POINTER(R1CS)
    for f in ("vpin_instance_num_cons_unpadded", "vpin_instance_num_vars_unpadded"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_size_t
    for f in ("vpin_instance_vars_para", "vpin_instance_vars_input", "vpin_instance_vars", "vpin_instance_inputs"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = vp
    L.vpin_instance_is_sat.argtypes = [vp]
    L.vpin_synthetic_points.argtypes = [C.c_uint64, C.c_size_t, vp, vp]
    L._gadgets_declared = True
    return L


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class Instance:
    """A padded R1CS instance with vPIN's three assignments (owned by the C++ side)."""

    def __init__(self, handle):
        self.h = handle
        L = _decl()
        r = L.vpin_instance_r1cs(self.h).contents
        self.num_cons, self.num_vars, self.num_inputs = int(r.num_cons), int(r.num_vars), int(r.num_inputs)
        self.num_cons_unpadded = int(L.vpin_instance_num_cons_unpadded(self.h))
        self.num_vars_unpadded = int(L.vpin_instance_num_vars_unpadded(self.h))
        self.nnz = [int(r.nnz[m]) for m in range(3)]

    def as_dict(self):
        """numpy copies in the layout the tests' oracle binding and Context.sat_prove take."""
        L = _decl()
        r = L.vpin_instance_r1cs(self.h).contents
        d = dict(num_cons=self.num_cons, num_vars=self.num_vars, num_inputs=self.num_inputs,
                 num_cons_unpadded=self.num_cons_unpadded, num_vars_unpadded=self.num_vars_unpadded)
        for m, name in enumerate("ABC"):
            n = int(r.nnz[m])
            d[name] = (_arr(r.row[m], (n,), np.uint32), _arr(r.col[m], (n,), np.uint32), _arr(r.val[m], (n, 4), np.uint64))
        d["vars_para"] = _arr(L.vpin_instance_vars_para(self.h), (self.num_vars, 4), np.uint64)
        d["vars_input"] = _arr(L.vpin_instance_vars_input(self.h), (self.num_vars, 4), np.uint64)
        d["vars"] = _arr(L.vpin_instance_vars(self.h), (self.num_vars, 4), np.uint64)
        d["inputs"] = _arr(L.vpin_instance_inputs(self.h), (self.num_inputs, 4), np.uint64)
        return d

    def is_sat(self):
        return _decl().vpin_instance_is_sat(self.h) == 1

    def free(self):
        if self.h:
            _decl().vpin_instance_free(self.h)
            self.h = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def point_add(px, py, rx, ry, rz):
    """px..ry: (N,32) uint8 little-endian; rz: (N,) uint8"""
    px, py, rx, ry = (np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32) for a in (px, py, rx, ry))
    rz = np.ascontiguousarray(rz, dtype=np.uint8).reshape(-1)
    h = C.c_void_p()
    rc = _decl().vpin_gadget_point_add(_p(px), _p(py), _p(rx), _p(ry), _p(rz), px.shape[0], C.byref(h))
    if rc:
        raise VpinError(rc, "vpin_gadget_point_add")
    return Instance(h)


def point_mult(weights, px, py):
    """weights: iterable of ints < 2^128; px, py: (N,32) uint8"""
    px, py = (np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32) for a in (px, py))
    w = np.frombuffer(b"".join(int(x).to_bytes(16, "little") for x in weights), dtype=np.uint8).copy()
    h = C.c_void_p()
    rc = _decl().vpin_gadget_point_mult(_p(w), _p(px), _p(py), px.shape[0], C.byref(h))
    if rc:
        raise VpinError(rc, "vpin_gadget_point_mult")
    return Instance(h)


def synthetic_points(seed, count):
    x = np.zeros((count, 32), dtype=np.uint8)
    y = np.zeros((count, 32), dtype=np.uint8)
    rc = _decl().vpin_synthetic_points(seed, count, _p(x), _p(y))
    if rc:
        raise VpinError(rc, "vpin_synthetic_points")
    return x, y


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return state, z ^ (z >> 31)


SEED = 0x5650494E  # generator seed, + config index

# SURVEY.md 8(d) / BASELINE.md section 2: op counts of the reference's configurations
CONFIGS = {
    "3_32": dict(index=1, n_mult=18, n_add=16, weights="conv3", rz_one_every=3),
    "A": dict(index=2, n_mult=178, n_add=2144, weights="wide", rz_one_every=300),
    "7_256": dict(index=3, n_mult=98, n_add=96, weights="conv7", rz_one_every=0),
    "E": dict(index=4, n_mult=658, n_add=2336, weights="wide", rz_one_every=300),
    # LeNet layers (src/LeNet/Server.py:690-698,753-761): conv layers use the 5x5 diagonal filter
    # {2,2,1,2,2} (Server.py:300), FC layers wide products; L2/L4 (pooling) have no point-mults
    "L1": dict(index=11, n_mult=300, n_add=288, weights="conv5", rz_one_every=6),
    "L2": dict(index=12, n_mult=0, n_add=7056, weights="conv5", rz_one_every=0),
    "L3": dict(index=13, n_mult=800, n_add=768, weights="conv5", rz_one_every=6),
    "L4": dict(index=14, n_mult=0, n_add=2400, weights="conv5", rz_one_every=0),
    "L5": dict(index=5, n_mult=6000, n_add=5760, weights="wide", rz_one_every=6),
    "L6": dict(index=16, n_mult=240, n_add=406, weights="wide", rz_one_every=300),
    "L7": dict(index=17, n_mult=168, n_add=186, weights="wide", rz_one_every=300),
}

LENET = ("L1", "L2", "L3", "L4", "L5", "L6", "L7")


def synthetic_mult_inputs(label, n_override=None):
    """(weights as Python ints, px, py) of a configuration's point multiplications, or None"""
    cfg = CONFIGS[label]
    n = n_override or cfg["n_mult"]
    if n == 0:
        return None
    x, y = synthetic_points(SEED + cfg["index"], n)
    if cfg["weights"] == "conv3":  # filter entries {1,0,1,2,0,2,1,0,1} x 2 (src/convolution/Server.py:453-455)
        base = [1, 0, 1, 2, 0, 2, 1, 0, 1]
        w = [base[i % 9] for i in range(n)]
    elif cfg["weights"] == "conv5":  # 5x5 filter with {2,2,1,2,2} on the diagonal, zeros elsewhere
        diag = {0: 2, 6: 2, 12: 1, 18: 2, 24: 2}
        w = [diag.get(i % 25, 0) for i in range(n)]
    elif cfg["weights"] == "conv7":  # 6 non-zeros in {1,2} of 49 (Server.py:463-469)
        w = [(1 + (i % 2)) if (i % 49) % 8 == 0 and (i % 49) < 48 else 0 for i in range(n)]
    else:  # FC layers: ~2^120..2^128 products of HMAC prefixes and scaled weights; uniform in [0, 2^127)
        st, w = SEED ^ 0xABCDEF ^ cfg["index"], []
        for _ in range(n):
            st, a = splitmix64(st)
            st, b = splitmix64(st)
            w.append(((a << 64) | b) >> 1)
    return w, x, y


def synthetic_mult_instance(label, n_override=None):
    inp = synthetic_mult_inputs(label, n_override)
    return None if inp is None else point_mult(*inp)


def synthetic_add_inputs(label, n_override=None):
    cfg = CONFIGS[label]
    n = n_override or cfg["n_add"]
    x, y = synthetic_points(SEED + 100 + cfg["index"], 2 * n)
    px, py, rx, ry = x[0::2].copy(), y[0::2].copy(), x[1::2].copy(), y[1::2].copy()
    rz = np.zeros(n, dtype=np.uint8)
    k = cfg["rz_one_every"]
    if k:
        idx = np.arange(0, n, k)
        rz[idx] = 1
        rx[idx] = 0
        ry[idx] = 0
    return px, py, rx, ry, rz


def synthetic_add_instance(label, n_override=None):
    return point_add(*synthetic_add_inputs(label, n_override))


def write_witness_files(root, label, n_mult=None, n_add=None):
    """The 8 JSON files of one network label under root/rust_files/<label>/, in the format the reference's
    Python service writes (src/convolution/Server.py:324-417) and VP/load_data*.rs read: N x 32 arrays of
    byte values, rz flags as ints, weights as decimal strings.  L2/L4 get no pointMult directory content."""
    import json
    import os
    pa = os.path.join(root, "rust_files", label, "pointAdd")
    pm = os.path.join(root, "rust_files", label, "pointMult")
    os.makedirs(pa, exist_ok=True)
    os.makedirs(pm, exist_ok=True)
    px, py, rx, ry, rz = synthetic_add_inputs(label, n_add)
    for name, a in (("px", px), ("py", py), ("rx", rx), ("ry", ry)):
        with open(os.path.join(pa, f"point_add_{name}_byte.json"), "w") as f:
            json.dump(a.tolist(), f)
    with open(os.path.join(pa, "point_add_rz_byte.json"), "w") as f:
        json.dump([int(v) for v in rz], f)
    m = synthetic_mult_inputs(label, n_mult)
    if m is not None:
        w, x, y = m
        with open(os.path.join(pm, "weight.json"), "w") as f:
            json.dump([str(v) for v in w], f)
        with open(os.path.join(pm, "point_mult_px_byte.json"), "w") as f:
            json.dump(x.tolist(), f)
        with open(os.path.join(pm, "point_mult_py_byte.json"), "w") as f:
            json.dump(y.tolist(), f)
