#!/bin/bash
# launch parameters of the scalar owner blocks (k_fused_p1hex) on 128^3 trilinear Laplace through the any-mesh entry (tools/generic_probe.py)
for cfg in ${FUSED_CFGS:-"A=0" "NH_FUSED_NT=384" "NH_FUSED_NT=448" "NH_FUSED_NT=512" "NH_FUSED_ROWS=216,NH_FUSED_NT=384" "NH_FUSED_ROWS=125,NH_FUSED_NT=256" "NH_FUSED_ROWS=343,NH_FUSED_NT=512" "NH_FUSED_ROWS=180,NH_FUSED_NT=320"}; do
  echo "$cfg: $(env ${cfg//,/ } python tools/generic_probe.py '3D P1 128' 2>&1 | tail -1)"
done
