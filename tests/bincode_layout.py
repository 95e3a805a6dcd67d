"""A schema walk over bincode(SNARK) -- test infrastructure, independent of both the oracle's and the product's writers.

bincode 1.3.3 defaults: little-endian, fixed-width integers, Vec<T> = u64 length + items, tuples / arrays /
structs = concatenation of their fields in declaration order.  Scalar = [u64; 4] (32 B), CompressedRistretto = 32 raw bytes.
The schemas below restate the reference's struct declarations (paths relative to src/proof_generation/Spartan/src/):

  lib.rs:330-338              SNARK { r1cs_sat_proof, inst_evals: (Scalar, Scalar, Scalar), r1cs_eval_proof }
  r1csproof.rs:22-47          R1CSProof
  sumcheck.rs:64-69           ZKSumcheckInstanceProof { comm_polys, comm_evals, proofs: Vec<DotProductProof> }
  sumcheck.rs:18-20           SumcheckInstanceProof { compressed_polys: Vec<CompressedUniPoly> }
  unipoly.rs:18-20            CompressedUniPoly { coeffs_except_linear_term: Vec<Scalar> }
  nizk/mod.rs                 KnowledgeProof, EqualityProof, ProductProof, DotProductProof, DotProductProofLog
  nizk/bullet.rs:16-19        BulletReductionProof { L_vec, R_vec }
  dense_mlpoly.rs:317-319     PolyEvalProof { proof: DotProductProofLog };  PolyCommitment { C: Vec<CompressedGroup> }
  r1csinstance.rs:326-328     R1CSEvalProof { proof: SparseMatPolyEvalProof }
  sparse_mlpoly.rs:1438-1441  SparseMatPolyEvalProof { comm_derefs, poly_eval_network_proof }
  sparse_mlpoly.rs:1326-1329  PolyEvalNetworkProof { proof_prod_layer, proof_hash_layer }
  sparse_mlpoly.rs:1036-1042  ProductLayerProof
  sparse_mlpoly.rs:699-707    HashLayerProof
  product_tree.rs:135-139,163-166  LayerProofBatched, ProductCircuitEvalProofBatched

`walk(buf, schema)` returns a tree of (name, start, end, children); a proof "parses" when the walk ends exactly at
len(buf) and every Vec length is plausible.
"""
import struct

S = ("S",)  # Scalar
P = ("P",)  # CompressedGroup


def Vec(t):
    return ("V", t)


def T(*fields):
    """struct / tuple: (name, type) pairs in declaration order"""
    return ("T", fields)


def Arr(t, n):
    return ("A", t, n)


PolyCommitment = T(("C", Vec(P)))
DotProductProof = T(("delta", P), ("beta", P), ("z", Vec(S)), ("z_delta", S), ("z_beta", S))
ZKSumcheckInstanceProof = T(("comm_polys", Vec(P)), ("comm_evals", Vec(P)), ("proofs", Vec(DotProductProof)))
KnowledgeProof = T(("alpha", P), ("z1", S), ("z2", S))
ProductProof = T(("alpha", P), ("beta", P), ("delta", P), ("z", Arr(S, 5)))
EqualityProof = T(("alpha", P), ("z", S))
BulletReductionProof = T(("L_vec", Vec(P)), ("R_vec", Vec(P)))
DotProductProofLog = T(("bullet_reduction_proof", BulletReductionProof), ("delta", P), ("beta", P), ("z1", S), ("z2", S))
PolyEvalProof = T(("proof", DotProductProofLog))
R1CSProof = T(
    ("comm_vars", PolyCommitment),
    ("sc_proof_phase1", ZKSumcheckInstanceProof),
    ("claims_phase2", Arr(P, 4)),
    ("pok_claims_phase2", T(("knowledge", KnowledgeProof), ("product", ProductProof))),
    ("proof_eq_sc_phase1", EqualityProof),
    ("sc_proof_phase2", ZKSumcheckInstanceProof),
    ("comm_vars_at_ry", P),
    ("proof_eval_vars_at_ry", PolyEvalProof),
    ("proof_eq_sc_phase2", EqualityProof),
)
CompressedUniPoly = T(("coeffs_except_linear_term", Vec(S)))
SumcheckInstanceProof = T(("compressed_polys", Vec(CompressedUniPoly)))
LayerProofBatched = T(("proof", SumcheckInstanceProof), ("claims_prod_left", Vec(S)), ("claims_prod_right", Vec(S)))
ProductCircuitEvalProofBatched = T(("proof", Vec(LayerProofBatched)),
                                   ("claims_dotp", T(("left", Vec(S)), ("right", Vec(S)), ("weight", Vec(S)))))
_eval_rc = T(("init", S), ("read", Vec(S)), ("write", Vec(S)), ("audit", S))
ProductLayerProof = T(
    ("eval_row", _eval_rc),
    ("eval_col", _eval_rc),
    ("eval_val", T(("left", Vec(S)), ("right", Vec(S)))),
    ("proof_mem", ProductCircuitEvalProofBatched),
    ("proof_ops", ProductCircuitEvalProofBatched),
)
_hash_rc = T(("addr", Vec(S)), ("read_ts", Vec(S)), ("audit_ts", S))
HashLayerProof = T(
    ("eval_row", _hash_rc),
    ("eval_col", _hash_rc),
    ("eval_val", Vec(S)),
    ("eval_derefs", T(("row", Vec(S)), ("col", Vec(S)))),
    ("proof_ops", PolyEvalProof),
    ("proof_mem", PolyEvalProof),
    ("proof_derefs", T(("proof_derefs", PolyEvalProof))),
)
PolyEvalNetworkProof = T(("proof_prod_layer", ProductLayerProof), ("proof_hash_layer", HashLayerProof))
SparseMatPolyEvalProof = T(("comm_derefs", T(("comm_ops_val", PolyCommitment))),
                           ("poly_eval_network_proof", PolyEvalNetworkProof))
R1CSEvalProof = T(("proof", SparseMatPolyEvalProof))
SNARK = T(("r1cs_sat_proof", R1CSProof), ("inst_evals", Arr(S, 3)), ("r1cs_eval_proof", R1CSEvalProof))


class Node:
    __slots__ = ("name", "start", "end", "kids", "count")

    def __init__(self, name, start):
        self.name, self.start, self.end, self.kids, self.count = name, start, start, [], None

    def __len__(self):
        return self.end - self.start

    def __getitem__(self, name):
        for k in self.kids:
            if k.name == name:
                return k
        raise KeyError(name)


def walk(buf, schema, name="root", off=0, max_vec=1 << 24):
    node = Node(name, off)
    kind = schema[0]
    if kind in ("S", "P"):
        off += 32
    elif kind == "V":
        (n,) = struct.unpack_from("<Q", buf, off)
        if n > max_vec:
            raise ValueError(f"{name}: implausible Vec length {n} at offset {off}")
        off += 8
        node.count = n
        for i in range(n):
            kid = walk(buf, schema[1], f"{i}", off, max_vec)
            if schema[1][0] not in ("S", "P"):
                node.kids.append(kid)
            off = kid.end
    elif kind == "A":
        node.count = schema[2]
        for i in range(schema[2]):
            off = walk(buf, schema[1], f"{i}", off, max_vec).end
    elif kind == "T":
        for fname, ftype in schema[1]:
            kid = walk(buf, ftype, fname, off, max_vec)
            node.kids.append(kid)
            off = kid.end
    else:
        raise ValueError(kind)
    if off > len(buf):
        raise ValueError(f"{name}: runs past the end of the buffer ({off} > {len(buf)})")
    node.end = off
    return node


def snark_sections(proof):
    """lengths the reference's profiler prints (Spartan/README.md:363,372,375) for one bincode(SNARK)"""
    root = walk(proof, SNARK)
    if root.end != len(proof):
        raise ValueError(f"trailing bytes: parsed {root.end} of {len(proof)}")
    ev = root["r1cs_eval_proof"]
    prod = ev["proof"]["poly_eval_network_proof"]["proof_prod_layer"]
    return dict(len_r1cs_sat_proof=len(root["r1cs_sat_proof"]), len_product_layer_proof=len(prod),
                len_r1cs_eval_proof=len(ev), total=len(proof), tree=root)
