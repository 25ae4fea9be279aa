# round 3, experiment 1: split grid at 5v5 (diagnostic build _s5) on the instantiation test, then baseline numbers of the tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e1; rm -rf $O; mkdir -p $O
RGS_LIB_SUFFIX=_s5 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "every_kernel_instantiation" > $O/s5_inst.log 2>&1; echo "s5 inst rc=$?" 
tail -15 $O/s5_inst.log
RGS_LIB_SUFFIX=_s5 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "c2_circles and 5-" > $O/s5_c2.log 2>&1; echo "s5 c2 rc=$?"
tail -5 $O/s5_c2.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err
timeout -k 10 200 python bench.py --config c2 --cpu-sample-arenas 1024 > $O/c2.json 2> $O/c2.err
timeout -k 10 200 python bench.py --config c3 --cpu-sample-arenas 256 > $O/c3.json 2> $O/c3.err
timeout -k 10 200 python bench.py --config c4 --cpu-sample-arenas 1024 > $O/c4.json 2> $O/c4.err
timeout -k 10 200 python bench.py --config c5 --no-streamed --cpu-sample-arenas 512 > $O/c5.json 2> $O/c5.err
RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --cpu-sample-arenas 256 > $O/drv_stamps.json 2> $O/drv_stamps.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['parity']['bit_identical'], d['events'])
    except Exception as e: print(f, 'ERR', e)
PY
