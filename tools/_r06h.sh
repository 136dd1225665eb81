cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
ls -d ~/.cache/miopen ~/.config/miopen 2>&1 | head
bash tools/session.sh r06h bench
du -sh ~/.cache/miopen ~/.config/miopen 2>&1 | head
for mode in default FAST; do
  rm -rf ~/.cache/miopen ~/.config/miopen
  if [ $mode = default ]; then ( time python bench.py --workload train --steps 12 --warmup 6 ) > $O/train_cold_$mode.json 2> $O/train_cold_$mode.err
  else ( time MIOPEN_FIND_MODE=$mode python bench.py --workload train --steps 12 --warmup 6 ) > $O/train_cold_$mode.json 2> $O/train_cold_$mode.err; fi
  grep real $O/train_cold_$mode.err; python -c "
import json; d=json.loads(open('$O/train_cold_$mode.json').read().strip().splitlines()[-1]); print('$mode', 'ms_per_step', d['ms_per_step'])"
done
