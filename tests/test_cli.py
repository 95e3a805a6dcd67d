"""vpin_prove: the reference binary's CLI / witness-file contract (VP/main.rs, load_data*.rs)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import gadgets_model as GM
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vpin_amd", "bin", "vpin_prove")


def write_witness(root, label, add_ops, mult_ops):
    """the 8 JSON files exactly as src/convolution/Server.py:324-417 writes them"""
    b32 = lambda v: list(int(v).to_bytes(32, "little"))
    pa = os.path.join(root, "rust_files", label, "pointAdd")
    pm = os.path.join(root, "rust_files", label, "pointMult")
    os.makedirs(pa)
    os.makedirs(pm)
    for name, idx in (("px", 0), ("py", 1), ("rx", 2), ("ry", 3)):
        json.dump([b32(o[idx]) for o in add_ops], open(os.path.join(pa, f"point_add_{name}_byte.json"), "w"))
    json.dump([int(o[4]) for o in add_ops], open(os.path.join(pa, "point_add_rz_byte.json"), "w"))
    json.dump([str(o[0]) for o in mult_ops], open(os.path.join(pm, "weight.json"), "w"))
    json.dump([b32(o[1]) for o in mult_ops], open(os.path.join(pm, "point_mult_px_byte.json"), "w"))
    json.dump([b32(o[2]) for o in mult_ops], open(os.path.join(pm, "point_mult_py_byte.json"), "w"))


@pytest.fixture(scope="module")
def built():
    from vpin_amd import build as vbuild
    vbuild.build()
    assert os.path.exists(BIN)


def test_missing_witness_files_fail_like_the_reference(built, tmp_path):
    r = subprocess.run([BIN, "nope"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 101 and "Failed to open file" in r.stderr
    assert r.stdout.startswith("network: nope")


def test_malformed_json_is_rejected(built, tmp_path):
    ops = GM.synthetic_add_ops(3, 2)
    write_witness(tmp_path, "X", ops, GM.synthetic_mult_ops(4, 1))
    with open(tmp_path / "rust_files" / "X" / "pointAdd" / "point_add_py_byte.json", "w") as f:
        f.write("[[1,2,")
    r = subprocess.run([BIN, "X"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 101 and "Failed to parse JSON" in r.stderr


@pytest.mark.gpu
def test_cli_proof_matches_oracle(built, tmp_path):
    add_ops = GM.synthetic_add_ops(0x5650494E, 6, rz_one_every=3)
    mult_ops = GM.synthetic_mult_ops(0x5650494E + 2, 1, weights=[(1 << 127) + 12345])
    write_witness(tmp_path, "T", add_ops, mult_ops)
    os.makedirs(tmp_path / "out")
    master = bytes(range(64)) + bytes((7 * i + 3) % 256 for i in range(64))
    # every proof draws its own RandomTape seeds (random.rs:14-20 inside each proof_point_*): with --seed the
    # CLI derives them per proof, SHAKE256(master || domain) -> commit seed | proof seed
    seeds = {name: hashlib.shake_256(master + dom).digest(128) for name, dom in (("add", b"vPIN/point_add"), ("mult", b"vPIN/point_mult"))}
    assert seeds["add"] != seeds["mult"]
    r = subprocess.run([BIN, "T", "--seed", master.hex(), "--write-proof", "out"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "network: T"
    assert lines[1:3] == ["Point Addition Gadget...", "Number of Point Additions: 6"]
    assert "Point Multiplication Gadget..." in lines and "Number of Point Multiplications: 1" in lines
    assert "Generating Proof..." in lines and "Still working on..." in lines
    assert any(l.startswith("Total proof size: ") for l in lines)
    assert lines.count("Proof verification successful!") == 2 and any(l.startswith("Proof verification time: ") for l in lines)
    for name, g in (("add", GM.build_point_add(add_ops)), ("mult", GM.build_point_mult(mult_ops))):
        inst = GM.instance_new(g)
        seed_c, seed_p = seeds[name][:64], seeds[name][64:]
        exp = O.snark_prove(inst, seed_c, seed_p)  # the CLI proves the whole SNARK, like the reference binary
        got = open(tmp_path / "out" / f"T_{name}.proof", "rb").read()
        assert got == exp["proof"]
        assert open(tmp_path / "out" / f"T_{name}.comm", "rb").read() == exp["comm"]
        assert f"Proof size: {len(got)} bytes" in lines
        res = dict(exp, proof=got,
                   comm_para=np.frombuffer(open(tmp_path / "out" / f"T_{name}.comm_para", "rb").read(), dtype=np.uint8).reshape(-1, 32).copy(),
                   comm_input=np.frombuffer(open(tmp_path / "out" / f"T_{name}.comm_input", "rb").read(), dtype=np.uint8).reshape(-1, 32).copy())
        assert O.snark_verify(inst, res) == 1
    # the two proofs must not share Hyrax row blinds: the add gadget's vars_para is all zero, so its comm_para rows
    # are b_i*H; with one shared tape comm_para_mult[i] - comm_para_add[i] would be an unblinded commitment to the
    # model parameters.  Same tape <=> same first row of the all-zero polynomial's commitment.
    inst_add = GM.instance_new(GM.build_point_add(add_ops))
    cp_add = open(tmp_path / "out" / "T_add.comm_para", "rb").read()
    shared = O.sat_prove(inst_add, seeds["mult"][:64], seeds["mult"][64:])["comm_para"].tobytes()
    assert cp_add[:32] != shared[:32]
    # without --seed every proof draws from the OS: two runs differ, and so do the two proofs' tapes
    os.makedirs(tmp_path / "o1")
    os.makedirs(tmp_path / "o2")
    for d in ("o1", "o2"):
        r = subprocess.run([BIN, "T", "--write-proof", d, "--sat-only"], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    rd = lambda d, f: open(tmp_path / d / f, "rb").read()
    assert rd("o1", "T_add.comm_para") != rd("o2", "T_add.comm_para")
    # --sat-only: the R1CS satisfiability proof alone
    os.makedirs(tmp_path / "out2")
    r = subprocess.run([BIN, "T", "--seed", master.hex(), "--write-proof", "out2", "--sat-only"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(tmp_path / "out2" / "T_add.proof", "rb").read() == O.sat_prove(inst_add, seeds["add"][:64], seeds["add"][64:])["proof"]


@pytest.mark.gpu
def test_several_labels_in_one_process_give_the_one_label_outputs(built, tmp_path):
    """script.sh:205-211 runs the reference binary once per LeNet layer; `vpin_prove L1 L2 ...` proves them in ONE process
    (HIP context, generator derivations and window tables built once, sized for the largest label of the run).  Every
    label's stdout block and -- with --seed -- every proof file must be what the one-label runs give; a label without
    point multiplications (L2 / L4, main.rs:24-31) prints the reference's zero block."""
    specs = {"U": (5, 2), "L2": (9, 0), "V": (3, 1)}
    for k, (label, (na, nm)) in enumerate(specs.items()):
        write_witness(tmp_path, label, GM.synthetic_add_ops(100 + k, na, rz_one_every=4), GM.synthetic_mult_ops(200 + k, max(nm, 1)))
    master = bytes((3 * i + 1) % 256 for i in range(128))
    os.makedirs(tmp_path / "all")
    r = subprocess.run([BIN, "U", "L2", "V", "--seed", master.hex(), "--write-proof", "all"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    blocks = r.stdout.split("network: ")[1:]
    assert [b.split("\n")[0] for b in blocks] == ["U", "L2", "V"]
    assert "Number of Point Multiplications: 0" in blocks[1] and "Proof generation time: 0 ms" in blocks[1]
    strip = lambda text: [l for l in text.splitlines() if l and " time: " not in l]   # times differ run to run
    for label, block in zip(specs, blocks):
        os.makedirs(tmp_path / ("one_" + label))
        r1 = subprocess.run([BIN, label, "--seed", master.hex(), "--write-proof", "one_" + label], cwd=tmp_path, capture_output=True, text=True)
        assert r1.returncode == 0, r1.stderr
        assert strip("network: " + block) == strip(r1.stdout), label
        assert block.count("Total proof generation time: ") == 1 and block.count("Proof verification successful!") == (1 if label == "L2" else 2)
        for f in sorted(os.listdir(tmp_path / ("one_" + label))):
            assert open(tmp_path / "all" / f, "rb").read() == open(tmp_path / ("one_" + label) / f, "rb").read(), f
    assert subprocess.run([BIN, "U", "--bogus"], cwd=tmp_path, capture_output=True, text=True).returncode == 101
