cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming 2>&1 | tail -1 | cut -c1-500
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming --policy hard 2>&1 | tail -1 | cut -c1-500
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming --batch 512 2>&1 | tail -1 | cut -c1-500
