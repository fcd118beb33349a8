cd $GRAFT_REPO_ROOT
for sw in NONE CNR_NO_SWEEP0 CNR_NO_NARROW_BWD CNR_NO_NARROW_DX CNR_NO_CHAIN_FWD CNR_NO_CHAIN_SDF CNR_NO_FUSED; do
  echo "== $sw"; env $sw=1 python -m pytest tests/test_edge_batches.py -m gpu -q -k "hip" 2>&1 | grep -E "^E  .*Assertion|passed|failed" | cut -c1-200
done
