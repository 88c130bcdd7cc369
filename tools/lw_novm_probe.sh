#!/bin/bash
# TIMING probe (GPU box, results wrong by construction): conv_row_lw_kernel with k-step waits that ignore vmcnt (KDCC_GEN_LW_NOVM=1): does the loop ever wait for a piece to land?
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for nb in 0 1 0 1; do
  KDCC_GEN_LW_NOVM=$nb python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
  echo "== NOVM=$nb"; KDCC_BENCH_BATCH=8 python tools/bench_conv.py --only "mod4 3x3 512,mod5 3x3,mod3 3x3 256,mod7 3x3" --iters 10 2>/dev/null | grep -v weighted | head -8
done
python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
