# A/B of library builds on the same box.  usage: ab.sh "<lib suffixes>" "<bench arg sets, comma-joined>" reps
set -e
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ab; mkdir -p gpurun_out/ab
C=robo-game-sim_amd/csrc
cp $C/librgs_hip.so /tmp/keep.so
for r in $(seq 1 ${3:-2}); do
  for w in $1; do
    if [ $w = cur ]; then cp /tmp/keep.so $C/librgs_hip.so; else cp $C/librgs_hip_$w.so $C/librgs_hip.so; fi
    for cfg in $2; do
      timeout -k 10 300 python bench.py $(echo $cfg | tr ',' ' ') > gpurun_out/ab/${w}_${cfg}_$r.json
    done
  done
done
cp /tmp/keep.so $C/librgs_hip.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.4g'%d['value'], d['events'].get('flagged'), (d.get('parity') or {}).get('bit_identical'))
PY
