cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_cif_decode.py tests/test_hip_streaming.py tests/test_hip_properties.py -x -q 2>&1 | tail -8
