#!/usr/bin/env python3
"""L2 hit rates per kernel family from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum pass -> profiles/*_l2_pmc.json.
usage: summarize_l2.py <counter_collection.csv> <out.json> [command string]"""
import collections
import csv
import json
import sys

FAMILIES = [("conv_row_lw", ("conv_row_lw_kernel",)), ("conv_row_tall (512 x 128)", ("conv_row_tall_kernel",)), ("conv_row_pp128", ("conv_row_pp128_kernel",)),
            ("conv_igemm_persist (1x1)", ("conv_igemm_persist_kernel",)), ("conv_igemm_row (rate 36)", ("conv_igemm_row_kernel",)),
            ("conv_igemm (one tile)", ("conv_igemm_kernel",)), ("conv_wgrad_wide", ("conv_wgrad_wide_kernel",)), ("conv_wgrad_pw_lw (1x1, lone wave)", ("conv_wgrad_pw_lw_kernel",)), ("conv_wgrad_lw (3x3, lone wave)", ("conv_wgrad_lw_kernel",)),
            ("dw_lw_fan3", ("dw_lw_fan3_kernel",)), ("dw_mfma_fwd", ("dw_mfma_fwd_kernel",)), ("dw_mfma_wgrad_multi", ("dw_mfma_wgrad_multi_kernel",)), ("dw_mfma_wgrad", ("dw_mfma_wgrad_kernel",)),
            ("dwconv_fwd", ("dwconv_fwd_kernel",))]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    for fam, keys in FAMILIES:
        if all(k in r["Kernel_Name"] for k in keys):
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[fam].add(r["Dispatch_Id"])
            break
out = {"source": "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum on `%s`" % (sys.argv[3] if len(sys.argv) > 3 else "?"),
       "formulae": "l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (reads and writes); requests are 64 B; ea_read_requests = L2 -> fabric read "
                   "requests (infinity cache / HBM)",
       "kernels": {}}
for fam, c in acc.items():
    n = len(disp[fam])
    out["kernels"][fam] = {"launches_profiled": n, "l2_hit_rate": c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0),
                           "requests_per_launch": c["TCC_REQ_sum"] / n, "ea_read_requests_per_launch": c["TCC_EA0_RDREQ_sum"] / n}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["l2_hit_rate"], 3) for k, v in out["kernels"].items()}))
