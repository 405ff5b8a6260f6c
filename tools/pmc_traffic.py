"""Aggregate rocprofv3 --pmc passes of bench.py into per-launch HBM traffic of the fused kernels.

    python tools/pmc_traffic.py <dir with the rocprofv3 output trees> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at
half their bytes (MI355X_MICROARCH.md, section HBM) -- the kernels here read 16 bytes per lane
(stream blocks, LDS-DMA pieces), so it is doubled.  The JSON carries the hash of the kernel
sources it was measured on; bench.py ignores it once they change.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(name):
    if "k_fused_ring<" in name:
        # k_fused_ring<D, Fn, HAS_GRAD, PS, LIN>(...): PS = 0 fp32 stream, 1 codebook, 2 byte index
        targs = name.split("k_fused_ring<", 1)[1].split(">(", 1)[0]
        flags = [t.strip() for t in targs.split(",")][-3:]
        if len(flags) == 3 and flags[0] == "true":
            return {"0": "ring_fp32", "1": "ring_codebook", "2": "ring_bytes"}.get(flags[1])
    if "k_ring_combine" in name:
        return "ring_combine"
    if "k_hub_rows" in name:
        return "hub_rows"
    if "k_hub_finish" in name:
        return "hub_finish"
    if "k_fused_wide4" in name:
        return "wide4"
    if "k_fused_small" in name:
        return "csr"
    return None


def main():
    src, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = classify(r["Kernel_Name"])
            if k:
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    import bench
    rec = {"source_sha": bench.source_sha(), "source_sha_wide": bench.source_sha(bench.WIDE_SOURCES),
           "counters_per_launch": {}, "bytes_per_launch": {}}
    for k, cs in acc.items():
        mean = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        rec["counters_per_launch"][k] = mean
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            rec["bytes_per_launch"][k] = 2.0 * mean["FETCH_SIZE"] * 1024.0 + mean["WRITE_SIZE"] * 1024.0
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec["bytes_per_launch"]), rec["source_sha"])


if __name__ == "__main__":
    main()
