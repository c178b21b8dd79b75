for r in 256 192 160 128 96 64; do
  echo -n "R=$r: "; NH_FUSED_ROWS=$r python tools/generic_probe.py "3D P1 128" 2>&1 | grep -i "128" | tail -1 | cut -c1-200
done
