#!/bin/bash
# backward pass (80 % layer_dw) with parts of layer_dw switched off (tuning build, WRONG results): what each part costs in time and joules
cd $GRAFT_REPO_ROOT
export CNR_LIB=$GRAFT_REPO_ROOT/tools/_build/libcolorneus_hip_tuning.so
for d in ${FDW_LIST:-0 2 1 3 4 5 0}; do
  CNR_FDW_DBG=$d python tools/energy_ledger.py --family backward --seconds 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('LEDGER '):
        r = json.loads(l[7:]); k = r['kernels']['layer_dw']
        print('dbg $d: %.3f ms/pass  %.0f W  %.0f MHz  %.2f J/pass  layer_dw %.3f ms' % (r['ms_per_pass'], r['W'], r['sclk'], r['W'] * r['ms_per_pass'] * 1e-3, k[1]))"
done
