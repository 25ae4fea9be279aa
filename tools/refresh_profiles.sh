# Round-end evidence on the GPU box (after tools/gate.sh): rocprofv3 passes of every bench shape (tools/prof.sh) and the
# whole-ensemble parity runs.  Outputs: gpurun_out/prof/<tag>/ and gpurun_out/whole/; tools/pmc_merge.py then
# installs the summaries into profiles/r03/.
set -e
cd $GRAFT_REPO_ROOT
bash tools/prof.sh c2_default --config c2 > /dev/null
bash tools/prof.sh c2_driver --steps 20 --warmup 5 > /dev/null
bash tools/prof.sh c4_default --config c4 > /dev/null
bash tools/prof.sh c5_default --config c5 > /dev/null
bash tools/prof.sh c3_default --config c3 > /dev/null
mkdir -p gpurun_out/whole
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sample-arenas 65536 > gpurun_out/whole/bench_c2_driver_all_arenas_checked.json
timeout -k 10 300 python bench.py --config c4 --cpu-sample-arenas 131072 > gpurun_out/whole/bench_c4_all_arenas_checked.json
timeout -k 10 300 python bench.py --config c5 --cpu-sample-arenas 32768 > gpurun_out/whole/bench_c5_all_arenas_checked.json
timeout -k 10 400 python bench.py --config c3 --arenas 16384 --cpu-sample-arenas 16384 > gpurun_out/whole/bench_c3_16384_arenas_checked.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/whole/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['parity']
    print(f, '%.4g'%d['value'], p['bit_identical'], p['arenas_checked'], p.get('host_streamed_bit_identical'))
for f in sorted(glob.glob('gpurun_out/prof/*/summary.json')):
    s=json.load(open(f)); print(s['key'], s.get('kernel'), s.get('timed_launch_ms'), s.get('FETCH_SIZE_KB'), s.get('WRITE_SIZE_KB'), s.get('SQ_INSTS_VALU'))
PY
