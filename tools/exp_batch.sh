#!/bin/bash
# sweep of the bench schedule: co-scheduled batches per launch sequence x HIP streams
for cfg in "24 3" "48 3" "64 3" "96 2" "36 4"; do set -- $cfg
r=$(timeout 300 python bench.py --group $1 --concurrency $2 --steps $(( $1 * $2 * 2 )) --warmup $(( $1 * $2 )) --no-cpu-baseline --timed-only 2>&1 | grep "timed region" | sed 's/.*-> //')
echo "group=$1 streams=$2 -> $r"; done
