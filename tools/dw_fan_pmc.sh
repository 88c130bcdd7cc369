#!/bin/bash
# HBM-side counters of the ASPP fan-out launch (tools/ubench/dw_fan_ab.py), lone-wave kernel and (KDCC_DW_LW=0) the 8-wave kernel:
# separate rocprofv3 --pmc passes with --kernel-trace only.  Output: gpurun_out/dwpmc/<kernel>_<pass>/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/dwpmc; mkdir -p $out
for lw in 1 0; do
  export KDCC_DW_LW=$lw
  i=0
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/lw${lw}_p$i -o p -- python3 $R/tools/ubench/dw_fan_ab.py > $out/lw${lw}_p$i.log 2>&1 || echo "pass $i ($ctr) failed" >> $out/fail.log
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/dwpmc"
for d in sorted(glob.glob(out + "/lw*_p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dw_" not in k: continue
            key = (k[:60], r["Counter_Name"])
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        for (k, c), (n, v) in sorted(agg.items()):
            print(os.path.basename(d), k, c, "launches", n, "mean", v / n)
PY
