# full GPU suite + smoke + quick bench lines of the current tree (outputs: gpurun_out/suite/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err
for c in c2 c3 c4 c5; do timeout -k 10 300 python bench.py --config $c --cpu-seconds 1.0 > $O/$c.json 2> $O/$c.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/suite/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['cpu_baseline']
        print(f, '%.4g'%d['value'], d['parity']['bit_identical'], d['parity']['arenas_checked'], 'cpu', c['value'], c['one_thread_value'], c['sample_seconds'], c['cpu_model'])
    except Exception as e: print(f, 'ERR', e)
PY
