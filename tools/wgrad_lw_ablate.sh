#!/bin/bash
# Phase ablations of conv_wgrad_lw_kernel (GPU box; TIMING only, the results are wrong by construction): the generated stage loop is
# rebuilt without one kind of instruction at a time (KDCC_GEN_WGRAD_ABL bits: 1 LDS-DMA, 2 fragment reads, 4 MFMAs, 8 zero fill,
# 16 address steps, 32 the pieces' memory requests into registers instead of LDS, 96 = 32 + an ordinary ds_write_b128 per piece) and one large layer is timed.  -> gpurun_out/wgrad_lw_ablate.txt; the shipped .inc / .so are restored at the end.
set -e
out=gpurun_out/wgrad_lw_ablate.txt; : > $out
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for abl in ${ABLS:-0 1 2 4 8 16 9 3 7 27 31 32 96}; do
  KDCC_GEN_WGRAD_ABL=$abl python tools/gen_wgrad_lw.py > /dev/null
  make -s -C $csrc > /dev/null 2>&1
  ms=$(KDCC_WGRAD_LW=1 python tools/wgrad_lw_check.py --child --only "${1:-mod7}" --iters 5 2>&1 >/dev/null | grep -o "[0-9.]* ms" | tr '\n' ' ')
  echo "abl $abl: $ms" | tee -a $out
done
python tools/gen_wgrad_lw.py > /dev/null
make -s -C $csrc > /dev/null 2>&1
