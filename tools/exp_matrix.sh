cd $GRAFT_REPO_ROOT
for cfg in "--faces 8 --streams 1" "--faces 8 --streams 2" "--faces 8 --streams 4" "--faces 8 --streams 6" "--faces 8 --streams 8" "--faces 16 --streams 4" "--faces 32 --streams 1" "--faces 32 --streams 2" "--faces 64 --streams 1" "--faces 128 --streams 1" "--faces 128 --streams 2"; do
python bench.py --no-cpu-baseline --steps 600 $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_stream') or {}; print('$cfg'.ljust(28), 'step %.1f G/s' % (d['value']/1e9), ' march %.4f ms' % (d['roofline']['avg_launch_ms']), ' ms/step %.4f' % d['ms_per_step'])"
done
