#!/bin/bash
# Same-box A/B of the one-wave-per-SIMD dense weight gradients (3x3 and 1x1) in the mode-B step (all 92 M student parameters trainable): new = default,
# old = KDCC_WGRAD_LW=0 KDCC_WGRAD_PW_LW=0 (the 8-wave kernels); alternating, fresh process each, 2 rounds.  -> gpurun_out/r6_modeb_ab.jsonl
out=gpurun_out/r6_modeb_ab.jsonl; rm -f $out
B="python bench.py --mode B --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --steps 8 --warmup 2"
P='import sys,json; r=json.loads(sys.stdin.read()); c=r["roofline"]["classes"]; print(json.dumps({"arm":sys.argv[1],"ms":r["ms_per_step"],"img_s":r["value"],"conv_wgrad":c.get("conv_wgrad"),"frac":r["roofline"]["frac"]}))'
for i in 1 2; do
  $B 2>/dev/null | tail -n 1 | python -c "$P" new >> $out
  KDCC_WGRAD_LW=0 KDCC_WGRAD_PW_LW=0 $B 2>/dev/null | tail -n 1 | python -c "$P" old >> $out
done
cat $out
