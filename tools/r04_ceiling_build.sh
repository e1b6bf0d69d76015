#!/bin/bash
# Probe builds for profiles/r04_ceiling.md (cross-compiled here; build/ is git-ignored but travels to the GPU box).
#   nobins   K1_DBG_EPI=16   K1 without its bin-index stores
#   ldsatom  K1_DBG_EPI=48   ... plus 16 LDS atomics per spectrum on a dummy counter image (counting inside K1)
#   epi15    K1_DBG_EPI=15   K1 without its epilogue arithmetic (no v_log, ambiguity, live / max, bin byte)
#   k2noatom K2_DBG=1        the count kernel without its LDS atomics (loads and address arithmetic kept)
#   k2store  K2_DBG=2        ... with plain LDS stores in their place
# Every probe library is built with -DFOSPHOR_AMD_PROBES: the FOSPHOR_AMD_DBG_* environment switches exist only there.
# Results of these builds are wrong by construction: timing only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/ab"
CSRC=$ROOT/gr-fosphor_amd/csrc
build() {
	hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -pthread -Wno-unused-function -DFOSPHOR_AMD_PROBES $2 -x hip -shared \
		-o "$ROOT/build/ab/lib_$1.so" $CSRC/fosphor_kernels.hip $CSRC/fosphor_cmap.hip $CSRC/fosphor_api.cpp $CSRC/fosphor_render.cpp \
		$CSRC/fosphor_sink.cpp $CSRC/fosphor_exchange.cpp -ldl &
}
build probes  ""			# the product kernels + the FOSPHOR_AMD_DBG_* switches (the product library has none of them)
build nobins  "-DK1_DBG_EPI=16"
build ldsatom "-DK1_DBG_EPI=48"
build epi15   "-DK1_DBG_EPI=15"
build k2noatom "-DK2_DBG=1"
build k2store  "-DK2_DBG=2"
wait
for u in read_skew; do
	hipcc --offload-arch=gfx950 -O3 -o "$ROOT/tools/ubench/$u" "$ROOT/tools/ubench/$u.hip"
done
ls -la "$ROOT/build/ab/"
