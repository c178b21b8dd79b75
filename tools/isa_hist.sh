#!/bin/bash
# usage: tools/isa_hist.sh <file.hip> <kernel-name-substring>   -> instruction histogram of one kernel (gfx950)
set -e
D=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -save-temps=obj -c "$1" -o $D/x.o 2>/dev/null
S=$(ls $D/*gfx950.s)
a=$(grep -n "^_Z.*$2.*:" $S | head -1 | cut -d: -f1)
b=$(grep -n "s_endpgm" $S | awk -F: -v a=$a '$1>a{print $1; exit}')
sed -n "${a},${b}p" $S > $D/k.s
echo "lines: $(wc -l < $D/k.s)  ($D/k.s)"
grep -oE "^\s+[a-z_0-9]+" $D/k.s | sort | uniq -c | sort -rn | head -${3:-40}
