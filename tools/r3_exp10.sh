cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e10; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_montecarlo.py tests/test_gpu_parity.py -x -q -k "montecarlo or recorded or snapshots or every_kernel or dealt or pooled" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/mc_log_rate.py 2>&1 | tail -4
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('drv', d['value'])"
timeout -k 10 200 python bench.py --config c3 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'])"
