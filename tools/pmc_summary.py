#!/usr/bin/env python3
"""Per-kernel-family summary of rocprofv3 --pmc passes of `bench.py` (one counter set per pass, as MI355X_MICROARCH.md
prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).

    python tools/pmc_summary.py OUT.json FETCH=dir/p_counter_collection.csv WRITE=... MFMA=...

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 B, so it
is doubled (MI355X_MICROARCH.md, 'HBM').  SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16);
MFMA pipe utilisation = busy cycles / (1024 SIMDs x kernel duration x clock), the clock taken from GRBM_GUI_ACTIVE / 8
per dispatch when present."""
import collections
import csv
import json
import re
import sys

FAMILIES = [("igemm_kernel", r"igemm(_s1|_pt)?_kernel<"), ("igemm3x3_kernel", r"igemm3x3_kernel<"),
            ("wgrad", r"wgrad_(split_)?kernel<"), ("wgrad_reduce_kernel", r"wgrad_reduce_kernel"),
            ("roi_align_fwd", r"roi_align_fwd"), ("roi_align_bwd_gather", r"roi_align_bwd_gather"),
            ("sgd_kernel", r"sgd_kernel")]


def family(name):
    for f, pat in FAMILIES:
        if re.search(pat, name):
            return f
    return None


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        a = agg[f][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg


def main():
    out_path = sys.argv[1]
    passes = dict(a.split("=", 1) for a in sys.argv[2:])
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_digest
    out = {"csrc_sha256": csrc_digest(), "units": "bytes per launch; FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-byte requests at 64 B), WRITE_SIZE "
                    "(KiB) x 1024; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x seconds x clock)", "kernels": {}}
    fam = collections.defaultdict(dict)
    if "FETCH" in passes:
        for f, c in load(passes["FETCH"]).items():
            n, v, _ = c["FETCH_SIZE"]
            fam[f]["launches_profiled"] = n
            fam[f]["fetch_bytes_per_launch"] = int(v * 1024 * 2 / max(n, 1))
    if "WRITE" in passes:
        for f, c in load(passes["WRITE"]).items():
            n, v, _ = c["WRITE_SIZE"]
            fam[f]["write_bytes_per_launch"] = int(v * 1024 / max(n, 1))
    if "MFMA" in passes:
        for f, c in load(passes["MFMA"]).items():
            n, busy, secs = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            gui = c.get("GRBM_GUI_ACTIVE")
            clock = (gui[1] / 8.0 / gui[2]) if gui and gui[2] > 0 else 2.0e9
            clock = min(clock, 2.4e9)          # the quotient reads high on dispatches shorter than ~0.3 ms (guide)
            fam[f]["mfma_busy_cycles_per_launch"] = int(busy / max(n, 1))
            fam[f]["avg_us_under_pmc"] = round(secs / max(n, 1) * 1e6, 2)
            fam[f]["clock_ghz_from_grbm"] = round(clock / 1e9, 3)
            fam[f]["mfma_util"] = round(busy / (1024.0 * secs * clock), 4) if secs > 0 else None
    for f, d in fam.items():
        if "fetch_bytes_per_launch" in d and "write_bytes_per_launch" in d:
            d["hbm_bytes_per_launch"] = d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]
        out["kernels"][f] = d
    with open(out_path, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
