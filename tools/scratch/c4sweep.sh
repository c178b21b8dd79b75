for w in 512 256 128 96 64 48 32 512 64; do
  echo -n "WGS=$w: "; C4_STEPS=12 NH_INDEX_COPY_WGS=$w python tools/c4_step.py 512 2>&1 | grep "STEPS"
done
