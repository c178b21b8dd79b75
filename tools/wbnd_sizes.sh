cd $GRAFT_REPO_ROOT
for n in 64 96 192 256; do for w in 16 20; do echo -n "n=$n wbnd=$w: "; NH_P1HEX_WBND=$w timeout 300 python bench.py --no-cpu --elements-per-axis $n --steps 100 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"; done; done
