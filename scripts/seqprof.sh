cd /tmp; export TMPDIR=/tmp
python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_fused_learner.py $GRAFT_REPO_ROOT/tests/test_gpu_trainer.py -x -q -m gpu -k "meta or golden or data_parallel or two_ranks" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python $GRAFT_REPO_ROOT/scripts/top_kernels.py /tmp/tr/t_results.db 8 | head -12
tail -1 /tmp/b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases']['meta_ms'])"
python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases']['meta_ms'])"
