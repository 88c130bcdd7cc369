#!/bin/bash
# Mode-B passes only (after a change to the weight-gradient kernels): kernel stats, FETCH_SIZE / WRITE_SIZE and MFMA counters, each its own
# rocprofv3 run with --kernel-trace only.   tools/profile_modeb.sh <tag>  ->  gpurun_out/<tag>/{profB,fetchB,writeB,mfmaB}
set -o pipefail
tag=${1:-profB}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --mode B"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profB -o modeB -- $B --steps 5 --warmup 2 > $out/b.log 2>&1 || exit 1
echo "[profile] mode B stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetchB -o f -- $B --steps 2 --warmup 1 > $out/fb.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE pass (mode B) done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/writeB -o w -- $B --steps 2 --warmup 1 > $out/wb.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass (mode B) done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfmaB -o m -- $B --steps 2 --warmup 1 > $out/m.log 2>&1 || exit 1
echo "[profile] MFMA pass (mode B) done"
find $out -name "*_kernel_trace.csv" -size +8M -delete
ls $out/*
