#!/bin/bash
# fan-out / sum anatomy: KDCC_DW_DBG bits on the tuning library (timing only: results are wrong by construction)
for dbg in ${DBGS:-0 1 2 4 8 16 3 24 27 32 64 96 123 127}; do
  echo "== KDCC_DW_DBG=$dbg"
  KDCC_LIB=tuning KDCC_DW_DBG=$dbg ONLY=fs python tools/ubench/aspp_dw_time.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print(' '.join(f'{k}={min(v[\"ms\"]):.3f}' for k,v in d.items() if isinstance(v,dict)))"
done
