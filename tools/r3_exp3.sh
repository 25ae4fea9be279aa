cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e3; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "dealt_evenly or pooled_kernels or sticky" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample-arenas 65536 --cpu-seconds 0.6 > $O/drv_$rep.json 2> $O/drv.err
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --variant 4 > $O/drv_pool_$rep.json 2>> $O/drv.err
done
timeout -k 10 200 python bench.py --config c2 --cpu-sample-arenas 1024 --cpu-seconds 0.6 > $O/c2.json 2> $O/c2.err
timeout -k 10 200 python bench.py --config c4 --cpu-sample-arenas 1024 --cpu-seconds 0.6 > $O/c4.json 2> $O/c4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], (d.get('parity') or {}).get('bit_identical'), (d.get('parity') or {}).get('arenas_checked'), d['events'])
    except Exception as e: print(f, 'ERR', e)
PY
