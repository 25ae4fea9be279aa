# C2-shaped workload at batch sizes from 1024 to 1 048 576 arenas (AUTO kernel choice), one fused launch each.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_batch; rm -rf $O; mkdir -p $O
for b in 1024 4096 16384 32768 65536 131072 262144 1048576; do
  k=500; if [ $b -ge 262144 ]; then k=200; fi
  timeout -k 10 400 python bench.py --no-cpu --arenas $b --steps $k > $O/b$b.json
done
python - <<'PY'
import json,glob
rows=[]
for b in (1024,4096,16384,32768,65536,131072,262144,1048576):
    d=json.loads(open('gpurun_out/sweep_batch/b%d.json'%b).read().strip().splitlines()[-1])
    rows.append(dict(arenas=b, steps=d["steps"], value=d["value"], ms_per_step=d["ms_per_step"], flagged=d["events"]["flagged"], auto_prefers_wave=d["events"]["auto_prefers_wave"]))
    print(rows[-1])
json.dump({"note":"tools/sweep_batch.sh: bench.py --no-cpu --arenas B --steps K (C2 shape, 2v2, dt 0.01, AUTO), one GPU","rows":rows},open('gpurun_out/sweep_batch/sweep_batch_size.json','w'),indent=1)
PY
