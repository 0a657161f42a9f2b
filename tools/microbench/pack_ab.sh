#!/bin/bash
# INVESTIGATION (GPU box): harris kernels on the 1/f^1.3 8K frame: tiles per wave (-DEFX_PACK_TPW=4|8|16), and with the Harris
# arithmetic and its pixel loads taken out (debug build, EFX_DEBUG=4: results invalid)
cd "$GRAFT_REPO_ROOT/cuda-efficient-features_amd/csrc"
rm -f detect_kernels.o efx_api.o && make -s -j8 EXTRA="-DEFX_DEBUG_BUILD" 2>&1 | grep -E "error" | head
cd "$GRAFT_REPO_ROOT"
for pack in 1 0; do for dbg in 0 4; do
  echo "== EFX_PACK=$pack EFX_DEBUG=$dbg"
  EFX_PACK=$pack EFX_DEBUG=$dbg timeout 200 tools/microbench/natural_prof.sh 1.3 2>&1 | grep -E "harris"
done; done
