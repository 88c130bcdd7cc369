#!/bin/bash
# The frozen teacher (next batch) on the side stream under the student's step, the chip shared by CUs: persistent conv grids of the
# teacher's launches capped at T workgroups, the student's at S (0 = all).  -> gpurun_out/r6_cushare.jsonl
out=gpurun_out/r6_cushare.jsonl; rm -f $out
B="python bench.py --no-cpu-baseline --no-batch-sweep --no-sub-records --no-profiler --steps 12 --warmup 4"
run() { # name, extra args, env...
  local name=$1; shift; local args=$1; shift
  env "$@" $B $args 2>/dev/null | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'arm':'$name','ms':r['ms_per_step'],'img_s':r['value']}))" >> $out
}
run main "" A=1
run backward "--teacher-stream backward" A=1
for ts in "128 128" "112 144" "96 160" "128 0" "64 192" "160 96"; do set -- $ts
  run "backward_T$1_S$2" "--teacher-stream backward" KDCC_TEACHER_CUS=$1 KDCC_STUDENT_CUS=$2
done
run main "" A=1
cat $out
