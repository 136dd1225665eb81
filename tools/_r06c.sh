cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
bash tools/session.sh r06c suite smoke
for m in 13 14; do
  GCFR_HIP_LIB=$PWD/geomconsistentfr_amd/lib/mut_$m.so timeout 300 python -m pytest tests/test_gpu_margins.py -m gpu -q -p no:cacheprovider -k "wrap_column" --tb=line 2>&1 | tail -4 > $O/mut_${m}_wrap_column_e2e.log; cat $O/mut_${m}_wrap_column_e2e.log
  GCFR_HIP_LIB=$PWD/geomconsistentfr_amd/lib/mut_${m}_audit.so timeout 600 python tools/audit.py --random 0 --families wrap_edge,wrap_column --family-seeds 8 --more "" --out $O/audit_mut_${m}_wrap.json | tail -c 700; echo
done
GCFR_HIP_LIB=$PWD/geomconsistentfr_amd/lib/audit.so timeout 600 python tools/audit.py --random 0 --families wrap_edge,wrap_column --family-seeds 8 --more "" --out $O/audit_product_wrap.json | tail -c 700; echo
bash tools/session.sh r06c train_breakdown
( time python bench.py --workload train --steps 12 --warmup 6 ) > $O/train_default.json 2> $O/train_default.err; tail -3 $O/train_default.err; tail -c 300 $O/train_default.json; echo
( time MIOPEN_FIND_MODE=2 python bench.py --workload train --steps 12 --warmup 6 ) > $O/train_fast.json 2> $O/train_fast.err; tail -3 $O/train_fast.err; tail -c 300 $O/train_fast.json; echo
bash tools/session.sh r06c bench
