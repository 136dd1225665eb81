cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f
for w in "--depth-noise 400" "" "--data ffhq" "--mask ones"; do
  echo "== $w"
  AB_STEPS=2000 AB_EXTRA="--no-worst-case --no-train-leg --no-parity-check $w" timeout 900 tools/ab.sh "lib:exp_ctl.so" "lib:exp_rough3.so" "lib:exp_rough4.so"
done > gpurun_out/r05f/ab_rough.txt 2>&1
cat gpurun_out/r05f/ab_rough.txt
