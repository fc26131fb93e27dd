for cfg in "16 3" "24 3" "32 3" "32 2" "16 4"; do set -- $cfg; 
r=$(timeout 300 python bench.py --group $1 --concurrency $2 --steps $(( $1 * $2 * 2 )) --warmup $(( $1 * $2 )) --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['serial_one_sequence_in_flight']['tokens_per_s'], d['roofline']['kernel'], d['roofline']['frac'])")
echo "group=$1 streams=$2 -> $r"; done
