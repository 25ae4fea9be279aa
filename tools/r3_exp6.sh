cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e6; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -k "dealt_evenly or pooled_kernels or sticky or every_kernel or snapshots or golden or pathological" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample-arenas 65536 --cpu-seconds 0.6 > $O/drv_$rep.json 2> $O/drv.err
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-rebalance > $O/drv_noreb_$rep.json 2>> $O/drv.err
done
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --no-cpu > $O/st.json 2>> $O/drv.err
timeout -k 10 200 python bench.py --config c2 --cpu-sample-arenas 2048 --cpu-seconds 0.6 > $O/c2.json 2> $O/c2.err
timeout -k 10 200 python bench.py --config c4 --cpu-sample-arenas 2048 --cpu-seconds 0.6 > $O/c4.json 2> $O/c4.err
timeout -k 10 200 python bench.py --config c4 --variant 5 --cpu-sample-arenas 2048 --cpu-seconds 0.6 > $O/c4_v5.json 2> $O/c4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e6/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['events']; st=e.get('stamps')
        print(f, '%.4g'%d['value'], d['roofline']['kernel_ms_per_launch'], (d.get('parity') or {}).get('bit_identical'), (d.get('parity') or {}).get('arenas_checked'), 'dealt', e.get('hot_arenas_dealt_to_other_workgroups'), 'flagged', e['flagged'])
        if st: print('    avg total', st['total_cycles']/1024, 'max', st['max_total_cycles'], 'ratio', st['max_total_cycles']/(st['total_cycles']/1024), 'lane', st['lane_cycles']/1024, 'wave', st['wave_cycles']/1024, 'rounds', st['rounds'], 'max lane passes', st['max_rounds'])
    except Exception as ex: print(f, 'ERR', ex)
PY
