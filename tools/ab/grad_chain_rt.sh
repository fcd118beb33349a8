cd $GRAFT_REPO_ROOT
for rt in 4 2; do
  echo "== CNR_CHAIN_GRAD=1 RT=$rt"; CNR_CHAIN_GRAD=1 CNR_CHAIN_FWD_RT=$rt python tools/step_records.py 4096 0.5 2>&1 | grep -E "chain|total"
done
echo "== default"; python tools/step_records.py 4096 0.2 2>&1 | grep -E "chain|layer_gemm_ws|narrow_dx|total"
