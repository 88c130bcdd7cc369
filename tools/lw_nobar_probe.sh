#!/bin/bash
# TIMING probe (GPU box, results wrong by construction): conv_row_lw_kernel's loop regenerated without the barrier of every even k-step
# (KDCC_GEN_LW_NOBAR=1) = the most a "one barrier per TWO k-steps" restructuring (B slots in pairs) could buy.  Measured, round 6: <= 0.7 %
# (mod7 6.09 -> 6.04 ms, the others within noise): the per-k-step barrier is not what the loop waits on.  Restores the shipped loop.
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for nb in 0 1 0 1; do
  KDCC_GEN_LW_NOBAR=$nb python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
  echo "== NOBAR=$nb"; KDCC_BENCH_BATCH=8 python tools/bench_conv.py --only "mod4 3x3 512,mod5 3x3,mod3 3x3 256,mod7 3x3" --iters 10 2>/dev/null | grep -v weighted | head -8
done
python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
