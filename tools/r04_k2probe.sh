for spec in "full:" "k2noatom:k2noatom" "k2store:k2store" "full2:" ; do
  label=${spec%%:*}; lib=${spec#*:}
  for rep in 1 2; do
    if [ -n "$lib" ]; then export FOSPHOR_AMD_LIB=$PWD/build/ab/lib_$lib.so; else unset FOSPHOR_AMD_LIB; fi
    python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r04b_${label}_$rep.json 2>/dev/null
    python3 tools/bline.py $label gpurun_out/r04b_${label}_$rep.json
  done
done
