for v in a1 a2 a3 a4; do
echo -n "$v: "; NUTILS_AMD_LIB=$PWD/nutils_amd/csrc/var/lib_$v.so python tools/generic_probe.py "3D P1 128" 2>&1 | grep -i "128" | tail -1 | cut -c60-130
done
