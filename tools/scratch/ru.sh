for v in 0_256 1_256 2_256 3_256 2_512 3_512; do
  echo -n "$v: "; NUTILS_AMD_LIB=$PWD/nutils_amd/csrc/var/lib_$v.so python tools/c3_bench.py 64 20 uniform 2>&1 | grep "kernel ms" | cut -c1-90
done
