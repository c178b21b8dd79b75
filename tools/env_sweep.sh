#!/bin/bash
# usage: tools/env_sweep.sh VAR v1 v2 ... : bench.py (c2, no cpu leg) once per value of the environment variable VAR
cd "$(dirname "$0")/.."
var=$1; shift
for v in "$@"; do
  echo -n "$var=$v: "
  env $var=$v timeout 300 python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f kernel_ms %.4f frac %.3f check %.1e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['checks']['owned_row_sums_rel']))"
done
