"""Build the gfx950 HIP library (vpin_amd/lib/libvpin_hip.so) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU present, so this also runs in the CPU-only
build container.  The built .so is git-ignored but travels with the tree to the GPU box.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvpin_hip.so")
ARCH = "gfx950"


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no CPU fallback exists)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def deps():
    return sources() + [os.path.join(CSRC, "cli", "vpin_prove.cpp")] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "host", "*.h")) + [
        os.path.join(os.path.dirname(HERE), "include", "vpin_hip.h")]


def up_to_date():
    if not os.path.exists(LIB_PATH):
        return False
    t = os.path.getmtime(LIB_PATH)
    return all(os.path.getmtime(s) <= t for s in deps())


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        newest_dep = max(os.path.getmtime(d) for d in deps() if not d.endswith((".hip", ".cpp")) or d == src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest_dep:
            continue
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-c", src, "-o", obj,
               "-I", os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("VPIN_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fopenmp", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_cli(verbose)
    return LIB_PATH


def build_cli(verbose=False):
    """vpin_amd/bin/vpin_prove: the reference binary's CLI contract over the C ABI."""
    src = os.path.join(CSRC, "cli", "vpin_prove.cpp")
    out_dir = os.path.join(HERE, "bin")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "vpin_prove")
    cmd = [hipcc(), "-O2", "-std=c++17", "-pthread", src, "-o", out, "-L", LIB_DIR, "-lvpin_hip", "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
