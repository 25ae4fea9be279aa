# Round gate on the GPU box: GPU tests, smoke, every bench config (the driver's own 20-step command first; C2 with ALL
# arenas checked against the oracle), the self-started 2-rank form of bench.py on the one GPU.  Outputs: gpurun_out/gate/.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gate
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gate/pytest.log 2>&1 || { tail -40 gpurun_out/gate/pytest.log; exit 1; }
tail -3 gpurun_out/gate/pytest.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/gate/smoke.log 2>&1 || { tail -20 gpurun_out/gate/smoke.log; exit 1; }
tail -1 gpurun_out/gate/smoke.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/gate/bench_c2_driver_20steps.json
for c in c2 c3 c4 c5; do timeout -k 10 400 python bench.py --config $c > gpurun_out/gate/bench_${c}_default.json; done
timeout -k 10 600 python bench.py --cpu-sample-arenas 65536 > gpurun_out/gate/bench_c2_all_arenas_checked.json
timeout -k 10 300 python bench.py --gpus 2 --steps 200 --backend gloo --force-device 0 > gpurun_out/gate/bench_2rank_rehearsal.json 2> gpurun_out/gate/bench_2rank_rehearsal.err
if [ -f robo-game-sim_amd/csrc/librgs_hip_stamps.so ]; then   # in-kernel phase stamps of the driver's command (make SUFFIX=_stamps STAMPS=1)
  RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps > gpurun_out/gate/bench_c2_driver_20steps_stamps.json
  RGS_LIB_SUFFIX=_stamps timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --stamps --variant 2 > gpurun_out/gate/bench_c2_driver_20steps_stamps_hybrid.json
fi
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/gate/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d.get('parity') or {}
    print(f, '%.4g'%d['value'], 'gpus', d['n_gpus'], 'steps', d['steps'], 'kernel ms/launch', d['roofline']['kernel_ms_per_launch'], 'frac', d['roofline']['frac'],
          'traffic', d['roofline']['traffic'], 'parity', p.get('bit_identical'), p.get('arenas_checked'), p.get('host_streamed_bit_identical'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
