cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_decoder.py tests/test_hip_cif_decode.py tests/test_hip_streaming.py tests/test_hip_edges.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>&1 | grep -E "timed passes"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_sequence_alone'], d['roofline']['class_ms_per_sequence']['argmax'])"
