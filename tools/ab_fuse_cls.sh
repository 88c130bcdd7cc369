#!/bin/bash
# Same-box A/B of the classifier epilogue (KDCC_FUSE_CLS=0: final[3]'s activation stored and the 1x1 classifier as its own launch), headline step,
# alternating fresh processes.
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --steps 10 --warmup 3"
for v in 1 0 1 0; do
  KDCC_FUSE_CLS=$v $B 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); c=r['roofline']['classes']; print('FUSE_CLS=$v', r['ms_per_step'], r['value'], c.get('conv_other_tiles'), c.get('conv3x3_row_lone_wave_256x256'), r['losses'])"
done
