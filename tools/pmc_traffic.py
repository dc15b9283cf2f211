#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into profiles/pmc_traffic.json,
the per-launch HBM traffic that bench.py reports as roofline.traffic.

    python tools/pmc_traffic.py gpurun_out/profile_<tag> profiles/pmc_traffic.json

Units and corrections (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read, so it is
doubled.  Checked on this path against known byte counts: the W4 compress reads 134.2 MB of
weights + 1.6 MB of scales / zero points and the doubled counter says 136.5 MB."""
import json
import re
import sys

ALIAS = {
    "ct::w4_quant_pack_kernel<2, true, true>": "w4_quant_pack_kernel<bf16>",
    "ct::w4_quant_pack_lean_kernel<2, true>": "w4_quant_pack_lean_kernel<bf16>",
    "ct::w4_unpack_dequant_kernel<2, 2, false>": "w4_unpack_dequant_kernel<bf16>",
    "ct::w4_unpack_dequant_kernel<2, 2, false, false>": "w4_unpack_dequant_kernel<bf16>",
    "ct::w4_unpack_dequant_kernel<2, 2, false, 2>": "w4_unpack_dequant_kernel<bf16>",
}


def parse(path, counter):
    out = {}
    for line in open(path):
        # kernel, counter, dispatches, avg_value [, full_n, full_avg_value]: the figure at the kernel's largest grid when the summary has it
        m = re.match(r"^(ct::.*?)\s+" + counter + r"\s+(\d+)\s+([\d.]+)(?:\s+(\d+)\s+([\d.]+))?\s*$", line)
        if m:
            out[m.group(1).strip()] = float(m.group(5) if m.group(5) is not None else m.group(3))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    # every kernel from the full bench; the W4 headline kernels from the --no-extra run, where each of their
    # launches is the 8192x8192 workload (the full bench also launches them on small TinyLlama-shaped tensors)
    fetch = parse(f"{src}/extras_fetch.txt", "FETCH_SIZE")
    write = parse(f"{src}/extras_write.txt", "WRITE_SIZE")
    fetch.update(parse(f"{src}/headline_fetch.txt", "FETCH_SIZE"))
    write.update(parse(f"{src}/headline_write.txt", "WRITE_SIZE"))
    res = {"_source": f"{src}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py",
           "_formula": "bytes per launch = 2 * FETCH_SIZE * 1024 (gfx950 read under-count) + WRITE_SIZE * 1024; launches at the kernel's largest grid only"}
    try:
        res["_srchash"] = open(f"{src}/srchash").read().strip()  # the library the counters were collected on (bench.py flags a mismatch as stale)
    except OSError:
        pass
    res["_tag"] = src.rstrip("/").split("/")[-1]
    for k in sorted(set(fetch) & set(write)):
        b = int(2 * fetch[k] * 1024 + write[k] * 1024)
        res[ALIAS.get(k, k.replace("ct::", ""))] = b
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
