"""Static instruction counts of the hot loops, from the gfx950 ISA hipcc emits -> profiles/r03_isa_counts.json.

bench.py's roofline.secondary prices msm_rows_kernel against the VALU-issue ceiling (one wave-instruction per SIMD
per 4 cycles); this is where its "VALU instructions per affine table addition" comes from.  Run in the build container:
    python tools/isa_counts.py
The loop of interest in a kernel is picked by its memory signature: the innermost loop (closed by a backward branch)
with exactly the expected number of vector loads -- 6 x dwordx4 = one 96-byte affine table entry for the MSM; the
pair loop of the round kernels is the loop with the most VALU instructions.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa(src, *defines):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + "".join(defines) + ".s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fopenmp", "-I", os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", src, "-o", out] + list(defines), stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_body(lines, mangled_prefix):
    start = [i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(":") or
             (l.startswith(mangled_prefix) and ": " in l and l.split(":")[0].startswith(mangled_prefix))]
    start = [i for i, l in enumerate(lines) if re.match(r"^" + re.escape(mangled_prefix) + r"\w*:", l)][0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[start:end]


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i]
            out.append({"valu": sum(1 for x in seg if re.match(r"^\s+v_", x)),
                        "v_mad_u64_u32": sum(1 for x in seg if "v_mad_u64_u32" in x),
                        "vmem_loads": sum(1 for x in seg if re.match(r"^\s+(global_load|buffer_load)", x)),
                        "vmem_stores": sum(1 for x in seg if re.match(r"^\s+(global_store|buffer_store)", x)),
                        "lines": len(seg)})
    return out


def main():
    res = {"_how": "tools/isa_counts.py: hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S, instructions between a loop's "
                   "label and its backward branch"}
    m = isa(os.path.join(ROOT, "vpin_amd", "csrc", "msm.hip"))
    ls = loops(kernel_body(m, "_ZN4vpin15msm_rows_kernel"))
    add = min((l for l in ls if l["vmem_loads"] == 6 and l["v_mad_u64_u32"] >= 400), key=lambda l: l["valu"])
    res["msm_rows_kernel"] = {"valu_per_table_add": add["valu"], "v_mad_u64_u32_per_table_add": add["v_mad_u64_u32"],
                              "loads_per_table_add": add["vmem_loads"],
                              "note": "one iteration of table_mul_acc's window loop: digit extraction, one 96-byte table entry "
                                      "(6 x dwordx4), one ge_add_niels (7 products mod 2^255-19)"}
    sc = isa(os.path.join(ROOT, "vpin_amd", "csrc", "sumcheck.hip"))
    ls = loops(kernel_body(sc, "_ZN4vpin16sc_cubic3_kernelILb1ELb1EE"))
    big = max(ls, key=lambda l: l["valu"])
    res["sc_cubic3_kernel<true, true>"] = {"valu_per_pair": big["valu"], "v_mad_u64_u32_per_pair": big["v_mad_u64_u32"],
                                           "loads_per_pair": big["vmem_loads"], "stores_per_pair": big["vmem_stores"]}
    sp = isa(os.path.join(ROOT, "vpin_amd", "csrc", "spark.hip"))
    ls = loops(kernel_body(sp, "_ZN4vpin17prod_round_kernelILb1ELb1ELb1EE"))
    big = max(ls, key=lambda l: l["valu"])
    # the loop holds the every-seventh-pair flush of the lazily reduced sums (sc_dev.h LeadAcc): price body and flush apart
    spb = isa(os.path.join(ROOT, "vpin_amd", "csrc", "spark.hip"), "-DVPIN_LAZY_NO_FLUSH")
    body = max(loops(kernel_body(spb, "_ZN4vpin17prod_round_kernelILb1ELb1ELb1EE")), key=lambda l: l["valu"])
    eff = lambda k: round(body[k] + (big[k] - body[k]) / 7.0)
    res["prod_round_kernel<true, true>"] = {"valu_per_pair": eff("valu"), "v_mad_u64_u32_per_pair": eff("v_mad_u64_u32"),
                                            "loads_per_pair": big["vmem_loads"], "stores_per_pair": big["vmem_stores"],
                                            "loop_with_flush": {"valu": big["valu"], "v_mad_u64_u32": big["v_mad_u64_u32"]},
                                            "loop_without_flush": {"valu": body["valu"], "v_mad_u64_u32": body["v_mad_u64_u32"]},
                                            "note": "per pair = the loop without the flush + a seventh of the flush (two Montgomery "
                                                    "reductions every seven pairs)"}
    out = os.path.join(ROOT, "profiles", os.environ.get("VPIN_ISA_OUT", "r04_isa_counts.json"))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
