#!/bin/bash
# Placement experiment of conv_wgrad_lw_kernel (GPU box): gap ranges of the fragment reads and of the LDS-DMA pieces / iterator groups in the
# two k-steps (KDCC_GEN_WGRAD_PHASES of tools/gen_wgrad_lw.py); one large layer timed per placement.  Restores the shipped loop at the end.
set -e
out=gpurun_out/wgrad_lw_phases.txt; : > $out
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
for ph in "" "0,23,24,47;0,23,24,47" "0,31,32,47;0,31,32,47" "16,47,0,15;16,47,0,15" "0,23,24,47;16,47,0,15" "0,15,16,47;0,15,16,47"; do
  KDCC_GEN_WGRAD_PHASES="$ph" python tools/gen_wgrad_lw.py > /dev/null
  make -s -C $csrc > /dev/null 2>&1
  ms=$(KDCC_WGRAD_LW=1 python tools/wgrad_lw_check.py --child --only "${1:-mod7}" --iters 5 2>&1 >/dev/null | grep -o "[0-9.]* ms" | tr '\n' ' ')
  echo "phases '$ph': $ms" | tee -a $out
done
python tools/gen_wgrad_lw.py > /dev/null
make -s -C $csrc > /dev/null 2>&1
