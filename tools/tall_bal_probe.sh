#!/bin/bash
# TIMING probe (GPU box, results wrong by construction): conv_row_tall_kernel's loop regenerated with three row-buffer pieces in EVERY k-step
# instead of nine in the period's last one (KDCC_GEN_TALL_BAL=1) = what a third row buffer (36 KiB, fits) and dynamic buffer addressing
# could buy at most.  Restores the shipped loop.
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for nb in 0 1 0 1; do
  KDCC_GEN_TALL_BAL=$nb python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
  echo "== TALL_BAL=$nb"; KDCC_BENCH_BATCH=8 python tools/bench_conv.py --only "mod2 3x3" --iters 10 2>/dev/null | grep -v weighted | head -4
done
python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
