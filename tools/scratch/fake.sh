echo -n "base: "; python tools/generic_probe.py "3D P1 128" 2>&1 | grep -i "128" | tail -1 | cut -c60-200
echo -n "fake verts: "; NUTILS_AMD_LIB=$PWD/nutils_amd/csrc/var/lib_fake.so python tools/generic_probe.py "3D P1 128" 2>&1 | grep -i "128" | tail -1 | cut -c60-200
