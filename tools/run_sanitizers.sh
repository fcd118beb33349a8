#!/bin/bash
# Host orchestration (cnr_plan.cpp: arena layout, descriptors, slot bookkeeping) + the CPU twin of the kernels under AddressSanitizer and
# UndefinedBehaviorSanitizer.  CPU only (GPU sanitizers are not available on the pool).  Writes profiles/<tag>_sanitizer_log.txt.
#   bash tools/run_sanitizers.sh r03
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-rXX}
make -C "$R/color-neus_amd/csrc" -j4 emu-asan > /dev/null || exit 1
LOG="$R/profiles/${TAG}_sanitizer_log.txt"
export CNR_EMU_LIB="$R/tests/_build/libcolorneus_emu_asan.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
# leaks: CPython itself never frees its interned objects; what matters here is out-of-bounds / use-after-free / UB in the library
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export OMP_NUM_THREADS=4
TESTS="tests/test_host_emu.py tests/test_edge_batches.py tests/test_sharded_gloo.py tests/test_sampler_functions.py tests/test_marching_cubes.py tests/test_fused_loss.py tests/test_linear_op.py tests/test_background_ops.py tests/test_forward_only.py tests/test_prune.py tests/test_rays.py tests/test_mesh_reference_semantics.py"
{
  echo "# build $(git -C "$R" rev-parse --short HEAD); g++ $(g++ -dumpversion); -fsanitize=address,undefined on cnr_plan.cpp + cnr_kernels_emu.cpp"
  echo "# python -m pytest $TESTS -m 'not gpu'"
  cd "$R" && python -m pytest $TESTS -q -x -m "not gpu" -p no:cacheprovider 2>&1 | grep -v "Warning\|warn" | tail -15
  ST=${PIPESTATUS[0]}
  echo "# exit status of pytest: $ST"
  echo "$ST" > "$LOG.status"
} > "$LOG" 2>&1
cat "$LOG"
ST=$(cat "$LOG.status"); rm -f "$LOG.status"
exit "$ST"
