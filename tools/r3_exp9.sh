cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e9; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -k "wide or (every_kernel and 2) or (every_kernel and 1) or (every_kernel and 3) or pooled_kernels or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $O/drv_v0_$rep.json 2> $O/drv.err
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --variant 6 --cpu-sample-arenas 65536 --cpu-seconds 0.6 > $O/drv_v6_$rep.json 2>> $O/drv.err
done
timeout -k 10 200 python bench.py --config c2 --no-cpu > $O/c2_v0.json 2> $O/c2.err
timeout -k 10 200 python bench.py --config c2 --variant 6 --cpu-sample-arenas 2048 --cpu-seconds 0.6 > $O/c2_v6.json 2>> $O/c2.err
timeout -k 10 200 python bench.py --config c4 --variant 6 --cpu-sample-arenas 2048 --cpu-seconds 0.6 > $O/c4_v6.json 2> $O/c4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e9/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['events']
        print(f, '%.4g'%d['value'], d['roofline']['kernel_ms_per_launch'], (d.get('parity') or {}).get('bit_identical'), (d.get('parity') or {}).get('arenas_checked'), 'dealt', e.get('hot_arenas_dealt_to_other_workgroups'), 'flagged', e['flagged'], 'ckpt', e.get('filled_from_checkpoints'))
    except Exception as ex: print(f, 'ERR', ex)
PY
