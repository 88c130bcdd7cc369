#!/usr/bin/env python3
"""Counter-reported MFMA utilisation per kernel family from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
(no trace domains besides --kernel-trace in the same run).  usage: summarize_mfma.py <counter_collection.csv> <out.json> [cmd]

Derivations (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is reported summed over the 8 XCDs -> shader cycles = value / 8;
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 256 CUs x 4 SIMDs -> utilisation = busy / (cycles x 1024);
SQ_INSTS_VALU_MFMA_MOPS_BF16 counts units of 512 FLOP (checked against the algorithmic FLOP count of the conv kernels:
516 algorithmic FLOP per unit, the difference being kernel rows skipped outside the image and the padded decoder K)."""
import collections
import csv
import json
import sys

FAMILIES = [("conv_row_lw<256x256, one wave per SIMD>", ("conv_row_lw_kernel",)), ("conv_igemm_row_persist<CfgRow 256x256>", ("conv_row_persist_kernel",)),
            ("conv_row_tall<512x128, one wave per SIMD>", ("conv_row_tall_kernel",)), ("conv_igemm_row_pp128<512x128>", ("conv_row_pp128_kernel",)),
            ("conv_igemm_persist<CfgWide 256x256 1x1>", ("conv_igemm_persist_kernel",)),
            ("conv_igemm_row<CfgRow 256x256>", ("conv_igemm_row_kernel", "CfgRowT<8, 2, 4, 128, 320")),
            ("conv_igemm_row<CfgRowX 256x256 rate-36>", ("conv_igemm_row_kernel", "CfgRowT<8, 2, 4, 128, 384")),
            ("conv_igemm_row<CfgRowN 256x128>", ("conv_igemm_row_kernel", "CfgRowT<4, 4, 2, 64")),
            ("conv_igemm<CfgWide 256x256 gathered>", ("conv_igemm_kernel", "Cfg<8, 2, 4, 2, 128")),
            ("conv_igemm<CfgNarrow2 256x128 gathered>", ("conv_igemm_kernel", "Cfg<4, 4, 2, 3, 64")),
            ("conv_wgrad_wide", ("conv_wgrad_wide_kernel",)), ("conv_wgrad_lw<3x3, one wave per SIMD>", ("conv_wgrad_lw_kernel",)), ("conv_wgrad_pw_lw<1x1, one wave per SIMD>", ("conv_wgrad_pw_lw_kernel",)), ("conv_wgrad_row", ("conv_wgrad_row_kernel",)), ("pw_wgrad_tr", ("pw_wgrad_tr_kernel",)),
            ("dw_lw_fan3 (lone-wave fan-out)", ("dw_lw_fan3_kernel",)), ("dw_mfma_fwd", ("dw_mfma_fwd_kernel",)), ("dw_mfma_wgrad", ("dw_mfma_wgrad_kernel",)),
            ("dw_mfma_wgrad_multi", ("dw_mfma_wgrad_multi_kernel",))]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    for fam, keys in FAMILIES:
        if all(k in r["Kernel_Name"] for k in keys):
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[fam] += 1
                acc[fam]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            break
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE on `%s`"
                 % (sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch-sweep"),
       "formulae": "cycles = GRBM_GUI_ACTIVE / 8; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); "
                   "counter_tflops = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / duration; frac_of_peak = counter_tflops / 2500",
       "kernels": {}}
tot = collections.defaultdict(float)
for fam, _ in FAMILIES:
    if not n[fam]:
        continue
    a = acc[fam]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    tf = a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / (a["ns"] * 1e-9) / 1e12
    out["kernels"][fam] = {"launches_profiled": n[fam], "avg_us": a["ns"] / n[fam] / 1e3, "mfma_util": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                           "counter_tflops": tf, "frac_of_2500": tf / 2500.0, "clock_ghz_from_grbm": cyc / a["ns"]}
    if fam.startswith("conv_") and not fam.startswith("conv_wgrad"):      # every forward / input-gradient conv family (lone-wave kernels included)
        for k, v in a.items():
            tot[k] += v
if tot["ns"]:
    cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
    tf = tot["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / (tot["ns"] * 1e-9) / 1e12
    out["conv_igemm_all"] = {"mfma_util": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), "counter_tflops": tf, "frac_of_2500": tf / 2500.0,
                             "clock_ghz_from_grbm": cyc / tot["ns"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
