#!/bin/bash
# Same-box A/B of the generator switch KDCC_GEN_LW_SPREAD (a row-buffer piece's instructions in four MFMA gaps / in one): the headline step
# with each generated loop, rebuilt on the GPU box, alternating, 2 rounds.  Restores the shipped loop.  -> gpurun_out/r6_spread_ab.txt
out=gpurun_out/r6_spread_ab.txt; : > $out
csrc=knowledge-distillation-by-replacing-cheap-conv_amd/csrc
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --steps 12 --warmup 3"
for i in 1 2; do
  for sp in 1 0; do
    KDCC_GEN_LW_SPREAD=$sp python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
    $B 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); c=r['roofline']['classes']; print('spread $sp', r['ms_per_step'], r['value'], {k: v for k, v in c.items() if k.startswith('conv3x3')})" | tee -a $out
  done
done
python tools/gen_conv_lw.py > /dev/null && make -s -C $csrc > /dev/null 2>&1
