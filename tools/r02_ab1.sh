#!/bin/bash
# GPU box: parity suite, then the bench under pipeline knobs (env), compact lines.
out=gpurun_out/ab1; mkdir -p $out
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
b default X=1
b default_again X=1
b alt0 FOSPHOR_AMD_ALT=0
b tile16 FOSPHOR_AMD_TILE=16
b tile32 FOSPHOR_AMD_TILE=32
b tile16_alt0 FOSPHOR_AMD_TILE=16 FOSPHOR_AMD_ALT=0
b sub32 FOSPHOR_AMD_SUB_LOG2=25
b sub128 FOSPHOR_AMD_SUB_LOG2=27
b sub16_tile16 FOSPHOR_AMD_SUB_LOG2=24 FOSPHOR_AMD_TILE=16
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --strict-ordering 2>$out/strict.err | python3 tools/bline.py strict
python3 bench.py --steps 20 --warmup 5 > $out/driver_style.json 2>$out/driver_style.err; python3 tools/bline.py driver_style < $out/driver_style.json
tail -3 $out/default.err
