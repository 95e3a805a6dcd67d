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


def main():
    labels = sys.argv[1:] or ["A"]
    from vpin_amd import gadgets as G
    binp = os.path.join(ROOT, "vpin_amd", "bin", "vpin_prove")
    with tempfile.TemporaryDirectory() as d:
        for lab in labels:
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
