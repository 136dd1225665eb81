#!/bin/bash
# Scaling experiment: march time vs sample count N and batch B (and optional gcfr_options knob settings (bench.py --tune)).
cd ${GRAFT_REPO_ROOT:-/root/repo}
TUNE=${1:-}
for n in 40 160 320 640; do for f in 8 64; do python bench.py --no-cpu-baseline --steps 50 --faces $f --samples $n --mask ${2:-ellipse} --size 256 --lights 1 ${TUNE:+--tune $TUNE} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=$TUNE B=$f N=$n', 'march %.4f ms' % d['roofline']['avg_launch_ms'], 'per (8 faces x 160 steps): %.1f us' % (d['roofline']['avg_launch_ms']*1e3*8/$f*160/$n))"; done; done
