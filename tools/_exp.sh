cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_cif_decode.py tests/test_hip_streaming.py -x -q 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s4/bench_k20b.json 2> gpurun_out/s4/bench_k20b.err; echo "bench rc=$?"
grep -E "configs|timed passes" gpurun_out/s4/bench_k20b.err | cut -c1-400
timeout 600 python tools/eval_sharded.py --utterances 5000 --streaming 2>&1 | tail -1 | cut -c1-300
