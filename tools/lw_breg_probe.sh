#!/bin/bash
# TIMING probe (GPU box, results wrong by construction): conv_row_lw_kernel's loop regenerated with the weight operand loaded straight into
# registers (KDCC_GEN_LW_BREG=1: per k-step and wave eight global_load_dwordx4 in place of four LDS-DMA pieces and eight ds_read_b128) = what
# taking B off the LDS could buy at most.  Restores the shipped loop.
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for nb in 0 1 0 1; do
  KDCC_GEN_LW_BREG=$nb python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
  echo "== BREG=$nb"; KDCC_BENCH_BATCH=8 python tools/bench_conv.py --only "mod4 3x3 512,mod5 3x3,mod3 3x3 256,mod7 3x3" --iters 10 2>/dev/null | grep -v weighted | head -8
done
python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
