cd $GRAFT_REPO_ROOT
for n in 2 8 20; do for f in 8 64; do for m in "" "--unfused"; do python bench.py --no-cpu-baseline --steps 100 --faces $f --samples $n --size 256 --lights 1 $m 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$f N=$n $m', 'march %.4f ms' % d['roofline']['avg_launch_ms'], 'step %.4f ms' % d['ms_per_step'])"; done; done; done
