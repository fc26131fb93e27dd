#!/bin/bash
# A/B of library builds in the driver's form: tools/sweep_lib_variants.sh build_x build_y ...  (each simulst_amd/csrc/<dir>/libsimulst_hip.so,
# made with `make -C simulst_amd/csrc BUILD=<dir> EXTRA=-D...`); "shipped" = the product library.  Two rounds each, interleaved.
for round in 1 2; do
  for v in shipped "$@"; do
    if [ "$v" = shipped ]; then unset SIMULST_LIB_PATH; else export SIMULST_LIB_PATH=$PWD/simulst_amd/csrc/$v/libsimulst_hip.so; fi
    printf "%-12s round %d: " $v $round
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline 2>&1 | grep "timed passes\|rror" | cut -c20-200
  done
done
