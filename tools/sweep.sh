# Team-size sweep: one-lane vs two-lane hybrid at 65 536 and 32 768 arenas, then AUTO with the CPU oracle beside it.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep; rm -rf $O; mkdir -p $O
for n in 1 2 3 4 5 6 7; do
  for v in 2 3; do
    for b in 65536 32768; do
      timeout -k 10 300 python bench.py --no-cpu --n-per-team $n --arenas $b --variant $v --steps 500 > $O/n${n}_v${v}_b$b.json
    done
  done
  timeout -k 10 400 python bench.py --n-per-team $n --steps 500 --cpu-sample-arenas 1024 > $O/n${n}_auto.json
done
python - <<'PY'
import json,glob
rows=[]
for n in range(1,8):
    r={"n_per_team":n}
    for v in (2,3):
        for b in (65536,32768):
            d=json.loads(open('gpurun_out/sweep/n%d_v%d_b%d.json'%(n,v,b)).read().strip().splitlines()[-1])
            r["%s_%d"%("one_lane" if v==2 else "two_lane",b)]=round(d["value"]/1e6,1)
    d=json.loads(open('gpurun_out/sweep/n%d_auto.json'%n).read().strip().splitlines()[-1])
    r.update(arenas=65536, steps=500, value=d["value"], ms_per_step=d["ms_per_step"], hbm_gbs=d["roofline"]["achieved"], hbm_frac=d["roofline"]["frac"],
             cpu_16_threads=d["cpu_baseline"]["value"], cpu_16_threads_reference_rng=d["cpu_baseline"]["reference_faithful_rng_value"],
             parity_bit_identical=d["parity"]["bit_identical"], flagged=d["events"]["flagged"])
    rows.append(r); print(r)
json.dump({"note":"tools/sweep.sh: bench.py --n-per-team n --steps 500 (C2-shaped workload, dt 0.01; the first 500 steps are more eventful than the 1000-step default); one_lane_/two_lane_ columns: forced variants at 65 536 / 32 768 arenas in 1e6 arena-steps/s; value: AUTO at 65 536 arenas with the CPU oracle on a 1024-arena sample","rows":rows},open('gpurun_out/sweep/sweep_team_size.json','w'),indent=1)
PY
