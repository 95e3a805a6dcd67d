#!/usr/bin/env python3
"""Pin vPIN's two R1CS gadgets to the TEXT of the reference's Rust sources, mechanically.

Run in the build container (the reference does not travel):

    python tests/golden/make_gadget_pins.py            # writes tests/golden/gadget_pins.json

What it does.  The bodies of `point_mult` (vPIN_proof_generation/src/point_mult.rs:20-651, with its helpers
`pa`, `pd`, `u128_to_128_bit_string`, `process_bit_string`, :667-729) and of `point_addition`
(point_addition.rs:18-313) are straight-line Rust over a tiny subset of the language: `let`, indexed
assignments, `for a..b`, `if / else if / else`, `Vec::push`, dalek `Scalar` arithmetic.  This script
translates that subset, line by line, into Python (every line must match a rule: an unknown line is an
error, nothing is dropped silently; the lines that are skipped on purpose are listed in SKIP), executes the
translation on explicit witness inputs with a 20-line big-integer `Scalar`, and records what the REFERENCE's
statements produced:

  * the (row, col, value) triplets of A, B, C in push order               -> count + SHA-256 per matrix
  * num_cons / num_vars / num_inputs / num_non_zero_entries (the param_1..3 chain of :27-67 / :38-70)
  * the three assignment vectors vars_para / vars_input / vars and inputs  -> SHA-256 each

Nothing here is a restatement by hand: tests/gadgets_model.py, the product's host builders and its device
builders are then checked against these digests (tests/test_gadget_pins.py, tests/test_gpu_gadget_pins.py).
The output file holds inputs and digests only -- no reference text.
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/proof_generation/vPIN_proof_generation/src"
Q = 2**252 + 27742317777372353535851937790883648493


# ---------------------------------------------------------------------------------------------------------
# dalek's Scalar as the gadgets use it (curve25519-dalek 3.2.0: arithmetic mod the group order l = q,
# invert = x^(l-2) so that 0 -> 0, to_bytes = 32 canonical little-endian bytes)

class Scalar:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v % Q

    @staticmethod
    def zero():
        return Scalar(0)

    @staticmethod
    def one():
        return Scalar(1)

    @staticmethod
    def from_int(x):
        return Scalar(int(x))

    @staticmethod
    def from_bytes_mod_order(b):
        return Scalar(int.from_bytes(bytes(b), "little"))

    def __add__(self, o):
        return Scalar(self.v + o.v)

    def __sub__(self, o):
        return Scalar(self.v - o.v)

    def __mul__(self, o):
        return Scalar(self.v * o.v)

    def invert(self):
        return Scalar(pow(self.v, Q - 2, Q))

    def to_bytes(self):
        return self.v.to_bytes(32, "little")


# ---------------------------------------------------------------------------------------------------------
# Rust subset -> Python

SKIP = [
    r"^println!\(",                       # stdout only
    r"^let \([\w, ]+\);$",                # `let (param_1, param_2, param_3);` -- a declaration without a value
    r"^let inst_1 = Instance::new\(",     # libspartan: padding / column remap, not part of the gadget
    r"^let assignment_\w+ = \w+::new\(",  # libspartan wrappers around the byte vectors recorded here
]
STOP = r"^let res(_1)? = inst_1\.is_sat\("  # the self-check: everything of interest is assigned before it


def _split_top(s, sep):
    """split s at top-level occurrences of the one-character separator"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def _vec_macros(e):
    """vec![X; n] -> [X for _ in range(n)], innermost first; vec![] -> []"""
    while True:
        k = e.rfind("vec![")
        if k < 0:
            return e
        depth, j = 0, k + 4
        while True:
            if e[j] == "[":
                depth += 1
            elif e[j] == "]":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        inner = e[k + 5:j]
        parts = _split_top(inner, ";")
        if inner.strip() == "":
            rep = "[]"
        elif len(parts) == 2:
            rep = "[%s for _ in range(%s)]" % (parts[0].strip(), parts[1].strip())
        else:
            raise ValueError("vec! form: " + e)
        e = e[:k] + rep + e[j + 1:]


def expr(e):
    e = e.strip()
    e = re.sub(r"\s+as\s+(usize|i32|i64|u8|u64|u128)\b", "", e)
    e = re.sub(r"\b(\d+)(u8|u64|u128|usize|i32)\b", r"\1", e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"&(?=\w)", "", e)                                   # borrows
    e = re.sub(r"(?<!/)/(?!/)", "//", e)                            # usize division
    e = _vec_macros(e)
    e = re.sub(r"\[\s*(\w+)\s*;\s*(\w+)\s*\]", r"[\1 for _ in range(\2)]", e)   # [0; 32]
    e = e.replace("Vec::new()", "[]").replace("String::new()", '""')
    e = e.replace("String::from(", "str(")
    e = e.replace("Scalar::from_bytes_mod_order(", "Scalar.from_bytes_mod_order(")
    e = e.replace("Scalar::zero()", "Scalar.zero()").replace("Scalar::one()", "Scalar.one()")
    e = e.replace("Scalar::from(", "Scalar.from_int(")
    e = re.sub(r"\b([a-z_]\w*)\.(\d+)\b", r"\1[\2]", e)              # tuple fields
    if "::" in e or "!" in e.replace("!=", ""):
        raise ValueError("untranslated expression: " + e)
    return e


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return [re.sub(r"//.*$", "", ln).strip() for ln in text.split("\n")]


def translate(lines, entry_skip_until=None):
    """lines: comment-free, stripped Rust lines of function bodies -> Python source"""
    out, depth = [], 0
    match_on = None      # (subject, first_arm_pending) while inside a `match x { ... }`
    skip_block = 0       # brace depth of a `let padded_* = { ... };` block being skipped
    started = entry_skip_until is None

    def emit(s):
        out.append("    " * depth + s)

    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        if not ln:
            continue
        if not started:
            if re.search(entry_skip_until, ln):
                started = True
            continue
        if skip_block:
            skip_block += ln.count("{") - ln.count("}")
            continue
        if re.search(STOP, ln):
            break
        if any(re.search(p, ln) for p in SKIP):
            continue
        if re.match(r"^let (mut )?padded_\w+ = \{$", ln):     # libspartan's VarsAssignment::pad
            skip_block = 1
            continue
        # ---- functions
        m = re.match(r"^fn (\w+)\((.*)\)\s*->\s*.*\{$", ln)
        if m:
            args = [a.split(":")[0].strip() for a in _split_top(m.group(2), ",")]
            emit("def %s(%s):" % (m.group(1), ", ".join(args)))
            depth += 1
            continue
        # ---- match on a char (process_bit_string)
        m = re.match(r"^match (\w+) \{$", ln)
        if m:
            match_on = [m.group(1), True]
            continue
        if match_on is not None:
            if ln == "}":
                match_on = None
                continue
            m = re.match(r"^(\S+) => (.*),$", ln)
            if not m:
                raise ValueError("match arm: " + ln)
            pat, body = m.group(1), m.group(2)
            if body.startswith("panic!("):
                body_py = "raise ValueError(%s)" % match_on[0]
            else:
                body_py = stmt(body + ";")
            if pat == "_":
                emit("else: " + body_py)
            else:
                emit(("if" if match_on[1] else "elif") + " %s == %s: %s" % (match_on[0], pat, body_py))
            match_on[1] = False
            continue
        # ---- control flow
        m = re.match(r"^\}\s*else if (.*)\{$", ln) or (re.match(r"^else if (.*)\{$", ln))
        if m:
            if ln.startswith("}"):
                depth -= 1
            emit("elif %s:" % expr(m.group(1)))
            depth += 1
            continue
        if re.match(r"^\}\s*else\s*\{$", ln) or re.match(r"^else\s*\{$", ln):
            if ln.startswith("}"):
                depth -= 1
            emit("else:")
            depth += 1
            continue
        m = re.match(r"^if (.*)\{$", ln)
        if m:
            emit("if %s:" % expr(m.group(1)))
            depth += 1
            continue
        m = re.match(r"^for \((\w+), &(\w+)\) in (.+)\.iter\(\)\.enumerate\(\)\s*\{$", ln)
        if m:
            emit("for %s, %s in enumerate(%s):" % (m.group(1), m.group(2), expr(m.group(3))))
            depth += 1
            continue
        m = re.match(r"^for (\w+) in \((.+)\.\.(.+)\)\.rev\(\)\s*\{$", ln)
        if m:
            emit("for %s in reversed(range(%s, %s)):" % (m.group(1), expr(m.group(2)), expr(m.group(3))))
            depth += 1
            continue
        m = re.match(r"^for (\w+) in (\w+)\.chars\(\)\.rev\(\)\s*\{$", ln)
        if m:
            emit("for %s in reversed(%s):" % (m.group(1), m.group(2)))
            depth += 1
            continue
        m = re.match(r"^for (\w+) in (.+?)\.\.(.+?)\s*\{$", ln)
        if m:
            emit("for %s in range(%s, %s):" % (m.group(1), expr(m.group(2)), expr(m.group(3))))
            depth += 1
            continue
        if ln in ("}", "};"):
            if out and out[-1].rstrip().endswith(":") and len(out[-1]) - len(out[-1].lstrip()) == 4 * (depth - 1):
                emit("pass")
            depth -= 1
            if depth < 0:
                raise ValueError("unbalanced braces")
            continue
        # ---- a bare identifier at the end of a function is its value
        if re.match(r"^\w+$", ln):
            emit("return " + ln)
            continue
        # ---- statements; a `(` left open continues on the next lines (the returned tuple is never reached)
        while ln.count("(") > ln.count(")") and i < len(lines):
            ln += " " + lines[i]
            i += 1
        emit(stmt(ln))
    if depth != 0:
        raise ValueError("translation ended at depth %d" % depth)
    return "\n".join(out) + "\n"


def stmt(ln):
    m = re.match(r"^(\w+)\.push_str\(&(\w+)\.to_string\(\)\);$", ln)
    if m:
        return "%s += str(%s)" % (m.group(1), m.group(2))
    m = re.match(r"^([\w\[\]]+)\.push\((.*)\);$", ln)
    if m:
        return "%s.append(%s)" % (m.group(1), expr(m.group(2)))
    m = re.match(r"^let (mut )?(\w+)\s*(:[^=]+)?=\s*(.+);$", ln)
    if m:
        return "%s = %s" % (m.group(2), expr(m.group(4)))
    m = re.match(r"^let \(([\w, ]+)\) = (.+);$", ln)
    if m:
        return "%s = %s" % (m.group(1), expr(m.group(2)))
    m = re.match(r"^([\w\[\]+\-*() ]+?)\s*=\s*(.+);$", ln)
    if m and "==" not in m.group(0).split("=")[0]:
        return "%s = %s" % (expr(m.group(1)), expr(m.group(2)))
    raise ValueError("no rule for line: " + ln)


# ---------------------------------------------------------------------------------------------------------

def load_n():
    """the bit width n is the literal load_data returns (load_data.rs:62)"""
    src = open(os.path.join(REF, "load_data.rs")).read()
    m = re.search(r"\(weights_len, weights, point_mult_x_byte, point_mult_y_byte, (\d+)\)", src)
    return int(m.group(1))


def program(fname, entry):
    lines = strip_comments(open(os.path.join(REF, fname)).read())
    return translate(lines, entry_skip_until=entry)


def split_setup(py):
    """(setup, rest): setup = everything before the first top-level loop (the sizes and the param chain)"""
    lines = py.split("\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith("for "))
    # the matrices are declared before the loop; keep declarations in the setup part
    return "\n".join(lines[:k]) + "\n", "\n".join(lines[k:]) + "\n"


def run_mult(py, weights, px, py_bytes, n, setup_only=False):
    env = dict(Scalar=Scalar, weights_len=len(weights), weight_list=list(weights),
               point_mult_x_byte=[list(b) for b in px], point_mult_y_byte=[list(b) for b in py_bytes], n=n)
    setup, rest = split_setup(py)
    exec(setup, env)
    if not setup_only:
        exec(rest, env)
    return env


def run_add(py, px, py_bytes, rx, ry, rz, setup_only=False):
    env = dict(Scalar=Scalar, len=len(px), point_add_px_byte=[list(b) for b in px],
               point_add_py_byte=[list(b) for b in py_bytes], point_add_rx_byte=[list(b) for b in rx],
               point_add_ry_byte=[list(b) for b in ry], point_add_rz_byte=list(rz))
    setup, rest = split_setup(py)
    exec(setup, env)
    if not setup_only:
        exec(rest, env)
    return env


def digest_triplets(trip):
    h = hashlib.sha256()
    for r, c, v in trip:
        h.update(int(r).to_bytes(8, "little") + int(c).to_bytes(8, "little") + bytes(v))
    return h.hexdigest()


def digest_vec(vec):
    h = hashlib.sha256()
    for v in vec:
        h.update(bytes(v))
    return h.hexdigest()


def record(env, names):
    A, B, C = (env[k] for k in names["abc"])
    out = dict(num_cons=env["num_cons_1"], num_vars=env["num_vars_1"], num_inputs=env["num_inputs_1"],
               num_non_zero_entries=env["num_non_zero_entries_1"],
               nnz=[len(A), len(B), len(C)],
               sha256=dict(A=digest_triplets(A), B=digest_triplets(B), C=digest_triplets(C),
                           vars_para=digest_vec(env["vars_para"]), vars_input=digest_vec(env["vars_input"]),
                           vars=digest_vec(env[names["vars"]]), inputs=digest_vec(env[names["inputs"]])))
    return out


# ---------------------------------------------------------------------------------------------------------
# witness inputs of the cases (explicit in the fixture; drawn here with Python's own SHA-256 counter stream
# so that the fixture does not depend on any other file of the repo)

E2_A = 3491403595575449084947959021303599933011749826127899762162894550148391771037
E2_GX = 4561981307020378385254256586024830594940985765081274686120783167106442831732
E2_GY = 684120277165286233470758410892647831027470652988879249692043589061244861334


def _e2_add(P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = (3 * x1 * x1 + E2_A) * pow(2 * y1, Q - 2, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, Q - 2, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def _e2_mul(k, P):
    acc = None
    while k:
        if k & 1:
            acc = _e2_add(acc, P)
        P = _e2_add(P, P)
        k >>= 1
    return acc


def _stream(tag):
    ctr = 0
    while True:
        yield int.from_bytes(hashlib.sha256(("%s/%d" % (tag, ctr)).encode()).digest(), "little")
        ctr += 1


def _points(tag, count):
    s = _stream(tag)
    return [_e2_mul(next(s) % (2**64) + 1, (E2_GX, E2_GY)) for _ in range(count)]


def b32(x):
    return list(int(x).to_bytes(32, "little"))


def mult_cases():
    cases = []
    for N in (1, 2, 18):
        pts = _points("mult%d" % N, N)
        s = _stream("w%d" % N)
        if N == 18:   # conv f=3's filter (src/convolution/Server.py:453-455) twice: zeros, ones, twos
            w = [1, 0, 1, 2, 0, 2, 1, 0, 1] * 2
        else:
            w = [next(s) % 2**128 for _ in range(N)]
        cases.append(dict(name="mult_N%d" % N, weights=[str(x) for x in w],
                          px=[b32(p[0]) for p in pts], py=[b32(p[1]) for p in pts]))
    # edge inputs: weight 0 and 2^128-1; coordinate bytes >= q (from_bytes_mod_order reduces); P with y = 0
    # (the doubling's inverse of zero), x = y = 0; the same point twice
    pts = _points("edge", 3)
    big = (Q + 5) % 2**256
    cases.append(dict(name="mult_edge", weights=[str(0), str(2**128 - 1), str(3), str(2**127), str(1)],
                      px=[b32(pts[0][0]), b32(pts[1][0]), [255] * 32, b32(pts[2][0]), b32(0)],
                      py=[b32(pts[0][1]), b32(0), b32(big), b32(pts[2][1]), b32(0)]))
    return cases


def add_cases():
    cases = []
    for N in (1, 2, 18):
        pts = _points("add%d" % N, 2 * N)
        rz = [1 if (N == 18 and i % 3 == 0) else 0 for i in range(N)]
        cases.append(dict(name="add_N%d" % N, px=[b32(pts[2 * i][0]) for i in range(N)],
                          py=[b32(pts[2 * i][1]) for i in range(N)],
                          rx=[b32(0) if rz[i] else b32(pts[2 * i + 1][0]) for i in range(N)],
                          ry=[b32(0) if rz[i] else b32(pts[2 * i + 1][1]) for i in range(N)], rz=rz))
    # edge inputs: R == P (inverse of zero -> 0), R = -P, bytes >= q, rz = 1 with junk in R, rz values
    # other than 0/1 (`== 0` else one: point_addition.rs:189-193)
    p = _points("addedge", 4)
    cases.append(dict(name="add_edge",
                      px=[b32(p[0][0]), b32(p[1][0]), [255] * 32, b32(p[2][0]), b32(p[3][0])],
                      py=[b32(p[0][1]), b32(p[1][1]), b32(Q + 7), b32(p[2][1]), b32(p[3][1])],
                      rx=[b32(p[0][0]), b32(p[1][0]), b32(p[0][0]), b32(p[3][0]), b32(0)],
                      ry=[b32(p[0][1]), b32(Q - p[1][1]), b32(p[0][1]), b32(p[3][1]), b32(0)],
                      rz=[0, 0, 0, 1, 7]))
    return cases


# operation counts whose sizes / declared nnz are pinned through the param chains alone (no triplet loop):
# every instance of BASELINE.json's configurations plus the branch boundaries of the two chains
MULT_COUNTS = [1, 2, 18, 50, 98, 168, 178, 210, 240, 300, 658, 659, 660, 800, 6000, 6001]
ADD_COUNTS = [1, 16, 96, 186, 288, 406, 499, 500, 768, 779, 780, 2130, 2131, 2144, 2149, 2150, 2336, 2400, 2449,
              2450, 5000, 5001, 5760, 7056, 7999, 8000]


def main():
    n = load_n()
    py_mult = program("point_mult.rs", r"= load_data\(network\);$")
    py_add = program("point_addition.rs", r"= load_data_add\(network\);$")
    # pa, pd and the bit helpers follow the body of point_mult in the same file: translate() stops the BODY at
    # the is_sat line, so the helpers are translated separately from the first `fn pa(`
    lines = strip_comments(open(os.path.join(REF, "point_mult.rs")).read())
    k = next(i for i, ln in enumerate(lines) if ln.startswith("fn pa("))
    py_helpers = translate(lines[k:])
    py_mult = py_helpers + py_mult

    out = dict(n=n, source="tests/golden/make_gadget_pins.py over point_mult.rs / point_addition.rs / load_data.rs",
               serialisation="triplets: row u64 LE || col u64 LE || 32 value bytes, push order; vectors: 32 B each, "
                             "unpadded (num_vars entries)", mult=[], add=[], mult_shapes=[], add_shapes=[])
    for c in mult_cases():
        env = run_mult(py_mult, [int(w) for w in c["weights"]], c["px"], c["py"], n)
        c.update(record(env, dict(abc="ABC", vars="vars", inputs="inputs")))
        out["mult"].append(c)
        print(c["name"], c["nnz"], c["num_non_zero_entries"], file=sys.stderr)
    for c in add_cases():
        env = run_add(py_add, c["px"], c["py"], c["rx"], c["ry"], c["rz"])
        c.update(record(env, dict(abc=("A1", "B1", "C1"), vars="vars_1", inputs="inputs_1")))
        out["add"].append(c)
        print(c["name"], c["nnz"], c["num_non_zero_entries"], file=sys.stderr)
    for N in MULT_COUNTS:
        env = run_mult(py_mult, [0] * N, [[0] * 32] * N, [[0] * 32] * N, n, setup_only=True)
        out["mult_shapes"].append(dict(ops=N, num_cons=env["num_cons_1"], num_vars=env["num_vars_1"],
                                       num_inputs=env["num_inputs_1"], num_non_zero_entries=env["num_non_zero_entries_1"]))
    for N in ADD_COUNTS:
        z = [[0] * 32] * N
        env = run_add(py_add, z, z, z, z, [0] * N, setup_only=True)
        out["add_shapes"].append(dict(ops=N, num_cons=env["num_cons_1"], num_vars=env["num_vars_1"],
                                      num_inputs=env["num_inputs_1"], num_non_zero_entries=env["num_non_zero_entries_1"]))
    path = os.path.join(HERE, "gadget_pins.json")
    with open(path, "w") as f:
        txt = json.dumps(out, separators=(",", ":"))   # compact, one case per line
        for key in ('{"name":', '"mult_shapes":', '"add_shapes":', '"add":['):
            txt = txt.replace(key, "\n" + key)
        f.write(txt + "\n")
    print("wrote", path, file=sys.stderr)
    if "--dump-py" in sys.argv:   # for inspection only; never committed
        sys.stdout.write(py_mult + "\n# ----\n" + py_add)


if __name__ == "__main__":
    main()
