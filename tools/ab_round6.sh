#!/bin/bash
# Same-box A/B of round 6's kernel changes: the headline step with (new) the lone-wave fan-out + the fused image-pooling sums, and (old)
# KDCC_DW_LW=0 KDCC_FUSE_GAP=0; alternating, fresh process each, 3 rounds.  -> gpurun_out/r6_ab.jsonl
out=gpurun_out/r6_ab.jsonl; rm -f $out
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --steps 12 --warmup 3"
for i in 1 2 3; do
  $B 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); c=r['roofline']['classes']; print(json.dumps({'arm':'new','ms':r['ms_per_step'],'img_s':r['value'],'fanout':c.get('depthwise_fwd_fanout'),'sum':c.get('depthwise_dgrad_sum'),'losses':c.get('losses')}))" >> $out
  KDCC_DW_LW=0 KDCC_FUSE_GAP=0 $B 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); c=r['roofline']['classes']; print(json.dumps({'arm':'old','ms':r['ms_per_step'],'img_s':r['value'],'fanout':c.get('depthwise_fwd_fanout'),'sum':c.get('depthwise_dgrad_sum'),'losses':c.get('losses')}))" >> $out
done
cat $out
