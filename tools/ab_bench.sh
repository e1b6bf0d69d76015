#!/bin/bash
# A/B harness: build K1 variants (here, cross-compiled) and bench them in ONE gpurun call.
#   tools/ab_bench.sh "name1:-DK1_X=1 -DK1_Y=2" "name2:..."     (run from the repo root)
# Variant libraries go to gpurun_out is NOT sent to the box, so they live in build/ab/ (git-ignored).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build/ab"
CSRC=$ROOT/gr-fosphor_amd/csrc
names=()
for spec in "$@"; do
	name=${spec%%:*}; flags=${spec#*:}
	if [ "$flags" = "PREBUILT" ]; then names+=("$name"); continue; fi	# build/ab/lib_$name.so already there
	hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wno-unused-function $flags -x hip -shared \
		-o "$ROOT/build/ab/lib_$name.so" $CSRC/fosphor_kernels.hip $CSRC/fosphor_api.cpp $CSRC/fosphor_render.cpp $CSRC/fosphor_sink.cpp \
		-Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "${AB_KERNEL:-k1v2_fft_binILb0}" | grep -E "VGPRs:|Scratch|Occupancy" | tr '\n' ' ' | sed "s/^/[$name] /"; echo
	names+=("$name")
done
cmd='for n in '"${names[*]}"'; do for rep in 1 2; do FOSPHOR_AMD_LIB=$PWD/build/ab/lib_$n.so python bench.py --steps ${AB_STEPS:-640} --warmup 64 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j[\"roofline\"]; print(\"'"'"'$n'"'"' value %.0f MS/s  k1 %.1f us (%.1f%%)  k2 %.1f us  k3 %.1f us\" % (j[\"value\"], r[\"k1_ms_per_launch\"]*1e3, 100*r[\"frac\"], r[\"k2_ms_per_launch\"]*1e3, r[\"k3_ms_per_launch\"]*1e3))"; done; done'
/usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] sending"
