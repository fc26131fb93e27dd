cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_streaming.py -x -q 2>&1 | tail -2
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming 2>&1 | tail -1 | cut -c1-330
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming --policy hard 2>&1 | tail -1 | cut -c1-330
