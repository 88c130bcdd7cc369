#!/usr/bin/env python3
"""Per-kernel-family HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md's HBM section prescribes) -> profiles/*_traffic_pmc.json.
usage: summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [command string] [images per GPU]
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 reads half the bytes a wide coalesced stream fetches)."""
import csv, json, sys

FAMILIES = {"conv_igemm": ("conv_igemm", "conv_row_persist", "conv_row_pp128", "conv_row_lw", "conv_row_tall"), "dw_mfma_fwd": "dw_mfma_fwd",
            "dw_lw_fan3": "dw_lw_fan3_kernel", "dw_mfma_wgrad_multi": "dw_mfma_wgrad_multi", "dw_mfma_wgrad": "dw_mfma_wgrad_kernel",
            "dwconv_fwd": "dwconv_fwd_kernel", "pw_wgrad": "pw_wgrad", "conv_wgrad_lw": "conv_wgrad_lw_kernel", "conv_wgrad_pw_lw": "conv_wgrad_pw_lw_kernel", "conv_wgrad_row": "conv_wgrad_row_kernel",
            "conv_wgrad_wide": "conv_wgrad_wide_kernel"}


def collect(path, counter):
    tot, cnt = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, key in FAMILIES.items():
            if any(k in r["Kernel_Name"] for k in ((key,) if isinstance(key, str) else key)):
                tot[fam] = tot.get(fam, 0.0) + float(r["Counter_Value"])
                cnt[fam] = cnt.get(fam, 0) + 1
                break
    return tot, cnt


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `%s`; FETCH_SIZE is in KiB and doubled per the "
                 "gfx950 correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE in KiB, exact" %
                 (sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 3 --warmup 1 --no-cpu-baseline"),
       "config": {"batch": int(sys.argv[5]) if len(sys.argv) > 5 else 8}, "kernels": {}}
for fam in FAMILIES:
    if fam in fetch and fam in write:
        f = 2.0 * 1024.0 * fetch[fam] / nf[fam]
        w = 1024.0 * write[fam] / nw[fam]
        out["kernels"][fam] = {"launches_profiled": nf[fam], "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                               "hbm_bytes_per_launch": f + w}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
