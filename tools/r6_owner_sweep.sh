#!/bin/bash
# launch parameters of k_owner_rows_v on 96^3 trilinear elasticity (tools/vector_probe.py): ms per re-assembly
for cfg in ${OWNER_CFGS:-"A=0" "NH_OWNER_PERSIST=0" "NH_OWNER_PERSIST=2" "NH_OWNER_NT=384" "NH_OWNER_ROWS=24,NH_OWNER_LDS=160" "NH_OWNER_ROWS=8" "NH_OWNER_XLDS=0"}; do
  echo "$cfg: $(env ${cfg//,/ } python tools/vector_probe.py ${OWNER_N:-96} 10 2>&1 | grep -o 'owner kernel [0-9.]* ms\|rows_per_block.: [0-9]*\|bit-identical: [A-Za-z]*\|difference [0-9.e-]*' | tr '\n' ' ')"
done
