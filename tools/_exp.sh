cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for plan in rows work; do
timeout 600 python tools/eval_sharded.py --utterances 5000 --plan $plan 2>&1 | tail -1 | cut -c1-420
timeout 600 python tools/eval_sharded.py --utterances 5000 --plan $plan --streaming 2>&1 | tail -1 | cut -c1-420
done
timeout 600 python tools/eval_sharded.py --utterances 5000 --plan work --streaming --policy hard 2>&1 | tail -1 | cut -c1-420
