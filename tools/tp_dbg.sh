#!/bin/bash
# diagnostics: fused-tail timing with parts of tconv_planes switched off (ODIN_TP_DBG bit mask)
for d in ${DBGS:-0 1 2 4 7}; do
  echo "== ODIN_TP_DBG=$d"
  ODIN_TP_DBG=$d timeout 200 python tools/stamps_t.py 2>&1 | grep -E "^--- fused|stamps|${PAT:-stamps}"
done
