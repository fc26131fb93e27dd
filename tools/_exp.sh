cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_streaming.py tests/test_hip_cif_decode.py -x -q 2>&1 | tail -3
for c in 1 2 3; do timeout 300 python3 tools/profile_streaming.py --config $c --rows 448 --self-paced --encoder offline 2>&1 | tail -1; done
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming 2>&1 | tail -1 | cut -c1-330
