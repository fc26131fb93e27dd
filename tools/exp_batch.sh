#!/bin/bash
# sweep of the bench schedule: co-scheduled batches per launch sequence x HIP streams (override with SWEEP="24,3 48,3")
for cfg in ${SWEEP:-24,3 48,3 64,3 96,2}; do g=${cfg%,*}; s=${cfg#*,}
r=$(timeout 300 python bench.py --group $g --concurrency $s --steps $(( g * s * 2 )) --warmup $(( g * s )) --no-cpu-baseline --timed-only 2>&1 | grep "timed region" | sed 's/.*-> //')
echo "group=$g streams=$s -> $r"; done
