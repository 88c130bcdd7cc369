#!/bin/bash
# Counters of the dense 3x3 weight gradient (tools/wgrad_lw_check.py --child, one layer): conv_wgrad_lw_kernel and (KDCC_WGRAD_LW=0)
# conv_wgrad_row_kernel; separate rocprofv3 --pmc passes with --kernel-trace only.  Output: gpurun_out/wgpmc/<arm>_p<pass>/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/wgpmc; rm -rf $out; mkdir -p $out
only="${1:-mod7}"
for lw in 1 0; do
  export KDCC_WGRAD_LW=$lw
  i=0
  for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/lw${lw}_p$i -o p -- python3 $R/tools/wgrad_lw_check.py --child --only "$only" --iters 2 > $out/lw${lw}_p$i.log 2>&1 || echo "pass $i ($ctr) failed" >> $out/fail.log
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/wgpmc"
for d in sorted(glob.glob(out + "/lw*_p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_wgrad" not in k: continue
            key = (k[:40], r["Counter_Name"])
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        for (k, c), (n, v) in sorted(agg.items()):
            print(os.path.basename(d), k, c, "launches", n, "mean", v / n)
PY
cat $out/fail.log 2>/dev/null
