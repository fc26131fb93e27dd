cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_streaming.py -x -q -k "concurrent" 2>&1 | tail -5
