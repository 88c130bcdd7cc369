#!/bin/bash
# Same-box A/B of the opt-in lone-wave 1x1 forward kernel (KDCC_CONV_LW_PW=1: conv_pw_lw_kernel) against the shipped ping-pong kernel in the
# headline step, alternating fresh processes.  Round 6, final tree: 172.25 ms (ping-pong) vs 174.15 ms; per layer (tools/bench_conv.py --only
# "1x1,pw") the lone-wave form is 1-7 % slower on every 1x1 shape, so it stays opt-in.
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler-ab --steps 10 --warmup 3"
for v in 0 1 0 1; do
  KDCC_CONV_LW_PW=$v $B 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); c=r['roofline']['classes']; print('LW_PW=$v', r['ms_per_step'], r['value'], {k: v for k, v in c.items() if k.startswith('conv')})"
done
