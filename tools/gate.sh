set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gate
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gate/pytest.log 2>&1 || { tail -40 gpurun_out/gate/pytest.log; exit 1; }
tail -3 gpurun_out/gate/pytest.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/gate/smoke.log 2>&1 || { tail -20 gpurun_out/gate/smoke.log; exit 1; }
tail -1 gpurun_out/gate/smoke.log
for c in c2 c3 c4 c5; do timeout -k 10 400 python bench.py --config $c > gpurun_out/gate/bench_$c.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/gate/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['events'].get('flagged'), d.get('parity'), d['cpu_baseline']['value'])
PY
