#!/bin/bash
# round 5, GPU session N (library 42970108...): the audit over every march unit and the soak matrix over every tile shape / group size /
# variant on the final source; the normals stage as its own kernel against the fused epilogue at config 5's shape (18 lights per face
# recompute the light-independent stencil 18 times), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; mkdir -p $O
for round in 1 2 3; do for ns in fused kernel; do
  python bench.py --no-cpu-baseline --no-worst-case --no-parity-check --no-train-leg --size 512 --lights 18 --samples 320 --faces 1 --steps 300 --normals-stage $ns 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_stream') or {}; print('normals-stage $ns'.ljust(24), 'step %.1f G/s' % (d['value']/1e9), ' %.4f ms/step' % d['ms_per_step'], ' 1-stream %.1f G/s' % (s.get('ray_steps_per_sec',0)/1e9))"
done; done > $O/config5_normals_stage_ab.txt 2>&1; cat $O/config5_normals_stage_ab.txt
bash tools/r05_audit_matrix.sh 2>&1 | tail -1
bash tools/r05_soak_matrix.sh 2>&1 | tail -1
