"""Reference-span timing of the drop-in CLI: writes the synthetic witness files of one label
(vpin_amd.gadgets.write_witness_files), runs vpin_amd/bin/vpin_prove <label> in that directory the way
`cargo run -- <label>` is run (VP/main.rs:14-46) and prints its stdout plus the VPIN_CLI_TRACE spans."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SCRUB = r"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
blocks = []
while True:
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), C.c_size_t(8 << 30)) != 0:
        break
    blocks.append(p)
for p in blocks:
    hip.hipFree(p)
print(f"scrub: took and explicitly freed {8 * len(blocks)} GB", flush=True)
"""


def scrub():
    """Bring VRAM to the state of an idle GPU: memory a previous process left behind at exit is only wiped when it
    is allocated again; allocate everything once, free it explicitly and give the background wipe time to finish."""
    subprocess.run([sys.executable, "-c", SCRUB], check=False)
    time.sleep(12)


def main():
    labels = [a for a in sys.argv[1:] if not a.startswith("--")] or ["A"]
    from vpin_amd import gadgets as G
    binp = os.path.join(ROOT, "vpin_amd", "bin", "vpin_prove")
    if "--one-process" in sys.argv:
        # vpin_prove <label> <label> ...: every label in ONE process (script.sh:205-211 starts one per label)
        import re
        with tempfile.TemporaryDirectory() as d:
            for lab in labels:
                G.write_witness_files(d, lab)
            if "--scrub" in sys.argv:
                scrub()
            t0 = time.perf_counter()
            r = subprocess.run([binp, *labels, "--seed", "00112233445566778899aabbccddeeff"], cwd=d, capture_output=True, text=True,
                               env=dict(os.environ, VPIN_CLI_TRACE="1"))
            wall = time.perf_counter() - t0
            print(r.stdout)
            print(r.stderr)
            gen = [int(x) for x in re.findall(r"Total proof generation time: (\d+) ms", r.stdout)]
            ver = [int(x) for x in re.findall(r"Total proof verification time: (\d+) ms", r.stdout)]
            print(f"== one process, labels {labels}: exit {r.returncode}, process wall {wall:.2f} s; "
                  f"'Total proof generation time' summed over the labels {sum(gen)} ms {gen}, verification {sum(ver)} ms", flush=True)
        return
    with tempfile.TemporaryDirectory() as d:
        for lab in labels:
            # VRAM freed by an earlier process is wiped by the driver before it can be allocated again (~30 GB/s);
            # let that finish so the run starts from the state an idle GPU is in
            time.sleep(float(os.environ.get("TIME_CLI_IDLE_S", "0")))
            if "--scrub" in sys.argv:
                scrub()
            t0 = time.perf_counter()
            G.write_witness_files(d, lab)
            print(f"== {lab}: witness files written in {time.perf_counter() - t0:.2f} s", flush=True)
            t0 = time.perf_counter()
            r = subprocess.run([binp, lab, "--seed", "00112233445566778899aabbccddeeff"], cwd=d, capture_output=True, text=True,
                               env=dict(os.environ, VPIN_CLI_TRACE="1", **({"VPIN_SPARK_TRACE": "1"} if os.environ.get("TIME_CLI_SPARK") else {})))
            wall = time.perf_counter() - t0
            print(r.stdout)
            print(r.stderr)
            print(f"== {lab}: exit {r.returncode}, process wall {wall:.2f} s", flush=True)


if __name__ == "__main__":
    main()
