#!/bin/bash
# Run ON the GPU box: bench each prebuilt build/ab/lib_<name>.so, optionally with env settings.
#   gpurun -- 'bash tools/ab_run.sh "name[,ENV=V[,ENV=V]]" ...'
for spec in "$@"; do
	IFS=, read -r name e1 e2 e3 <<< "$spec"
	for rep in 1 2; do
		env FOSPHOR_AMD_LIB=$PWD/build/ab/lib_$name.so $e1 $e2 $e3 python3 bench.py --steps ${AB_STEPS:-640} --warmup ${AB_WARMUP:-64} --no-cpu-baseline 2>/dev/null | tail -1 | \
		python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; i=r['isolated'] or {}; print('%-28s value %.0f MS/s  k1 %.1f us (%.1f%%)  k2 %.1f us  k3 %.1f us  k1 alone %.1f us' % ('$spec', j['value'], r['k1_ms_per_launch']*1e3, 100*r['frac'], r['k2_ms_per_launch']*1e3, r['k3_ms_per_launch']*1e3, i.get('k1_ms_per_launch',0)*1e3))"
	done
done
