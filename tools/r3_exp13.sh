cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for spl in 20 10 5; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --steps-per-launch $spl | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spl', $spl, d['value'], d['events']['hot_arenas_dealt_to_other_workgroups'])"
done; done
