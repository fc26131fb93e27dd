cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_cif_decode.py tests/test_hip_streaming.py -x -q 2>&1 | tail -8
echo "config 3 lockstep"; timeout 300 python3 tools/profile_streaming.py --config 3 --rows 448 2>&1 | tail -1
echo "config 3 self-paced"; timeout 300 python3 tools/profile_streaming.py --config 3 --rows 448 --self-paced 2>&1 | tail -1
echo "config 3 self-paced offline"; timeout 300 python3 tools/profile_streaming.py --config 3 --rows 448 --self-paced --encoder offline 2>&1 | tail -1
