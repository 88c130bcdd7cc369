#!/bin/bash
# Profile passes of one round on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{profA,profB,profG,fetch,write,fetchB,writeB,fetchG,writeG,tcc,mfma}
# Kernel stats (mode A headline, mode B, GSCNN) and the three counter passes, each its own rocprofv3 run with
# --kernel-trace only (no other trace domain next to --pmc).  Summaries for profiles/ are made afterwards with
# tools/summarize_profile.py / summarize_pmc.py / summarize_mfma.py.
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profA -o modeA -- $B --steps 5 --warmup 2 > $out/a.log 2>&1 || exit 1
echo "[profile] mode A stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profB -o modeB -- $B --steps 5 --warmup 2 --mode B > $out/b.log 2>&1 || exit 1
echo "[profile] mode B stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/profG -o gscnn -- $B --steps 5 --warmup 2 --arch gscnn --plan P86 > $out/g.log 2>&1 || exit 1
echo "[profile] gscnn stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B --steps 3 --warmup 1 > $out/f.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B --steps 3 --warmup 1 > $out/w.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetchB -o f -- $B --steps 2 --warmup 1 --mode B > $out/fb.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE pass (mode B) done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/writeB -o w -- $B --steps 2 --warmup 1 --mode B > $out/wb.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass (mode B) done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetchG -o f -- $B --steps 2 --warmup 1 --arch gscnn --plan P86 > $out/fg.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE pass (gscnn) done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/writeG -o w -- $B --steps 2 --warmup 1 --arch gscnn --plan P86 > $out/wg.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass (gscnn) done"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/tcc -o t -- $B --steps 2 --warmup 1 > $out/t.log 2>&1 || exit 1
echo "[profile] L2 hit / miss pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o m -- $B --steps 3 --warmup 1 > $out/m.log 2>&1 || exit 1
echo "[profile] MFMA pass done"
# keep only what the summarizers read (the traces are tens of MB)
find $out -name "*_kernel_trace.csv" -size +8M -delete
ls -la $out/*
