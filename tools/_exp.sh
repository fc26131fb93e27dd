cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_streaming.py -x -q 2>&1 | tail -8
for c in 1 2; do echo "config $c lockstep"; timeout 300 python3 tools/profile_streaming.py --config $c --rows 448 2>&1 | tail -1;  echo "config $c self-paced"; timeout 300 python3 tools/profile_streaming.py --config $c --rows 448 --self-paced 2>&1 | tail -1; done
