#!/bin/bash
# usage: tools/kstat.sh file.hip kernel-name-substring : register / scratch / spill statistics of matching kernels (gfx950 ISA)
f=$1; pat=$2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-jump-tables ${KSTAT_FLAGS} -S --cuda-device-only -o /tmp/kstat.s $f 2>/dev/null
awk -v pat="$pat" '
/^_Z.*:$/ || /^[A-Za-z_].*:$/ {name=$1}
/\.amdhsa_next_free_vgpr|\.amdhsa_next_free_sgpr|\.amdhsa_private_segment_fixed_size|\.amdhsa_accum_offset/ { if (name ~ pat) print name, $1, $2 }
/; ScratchSize|; NumVgprs|; NumAgprs|; TotalNumVgprs|; Occupancy|; codeLenInByte|; NumSgprs|spill/i { if (lastk ~ pat) print "   ", $0 }
/^\s*\.amdhsa_kernel/ {lastk=$2}
' /tmp/kstat.s | sort | uniq | head -60
