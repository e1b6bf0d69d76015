#!/bin/bash
# Build A/B variants of the product library into build/ab/ (cross-compiled here; build/ travels to the GPU box):
#   tools/ab_build.sh "name1:-DK1H_SPLIT=1" "name2:-DK1H_LOAD_ORDER=0 -DK1H_SPLIT=2"      then   gpurun -- 'bash tools/ab_libs.sh C5 3 name1 name2 cur'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/ab"
CSRC=$ROOT/gr-fosphor_amd/csrc
for spec in "$@"; do
	name=${spec%%:*}; flags=${spec#*:}
	hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -pthread -Wno-unused-function $flags -x hip -shared \
		-o "$ROOT/build/ab/lib_$name.so" $CSRC/fosphor_kernels.hip $CSRC/fosphor_cmap.hip $CSRC/fosphor_api.cpp $CSRC/fosphor_render.cpp \
		$CSRC/fosphor_sink.cpp $CSRC/fosphor_exchange.cpp -ldl 2>&1 | grep -E "error|spill" || true &
done
wait
ls -la "$ROOT/build/ab/"
