import sys, numpy as np, time
sys.path.insert(0,'.'); sys.path.insert(0,'robo-game-sim_amd')
import robo_game_py as rgp
from robo_game_py import workloads as w
from oracle import oracle as orc
for (n,B,K,dt,var) in [(2,256,40,0.05,5),(2,4096,60,0.01,5),(3,1000,50,0.05,5),(2,4096,60,0.01,4),(7,600,30,0.05,5)]:
    x0,amp,ph=w.c2_inputs(B,n)
    table=w.circle_controls(amp*2,ph,w.step_times(K,dt))
    sim=rgp.BatchedGameSim(B,n); sim.set_variant(var); sim.reset(dt,1000000,x0)
    t=time.time(); got=sim.run_table(table,chunk_steps=K); el=time.time()-t
    c=sim.counters(); sim.close()
    ref=orc.batch_run(n,x0,table,K,dt=dt,winning_score=1000000,n_threads=8)
    print(n,B,K,var,'equal',np.array_equal(got,ref['state']),'maxdiff',np.max(np.abs(got-ref['state'])),c['resolved'],ref['stats']['resolved'],c['flagged'],'%.3fs'%el, flush=True)
