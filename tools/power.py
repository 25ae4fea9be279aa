import sys, os, time, subprocess, threading, json
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "robo-game-sim_amd"))
import robo_game_py as rgp
from robo_game_py import workloads as w
B, n, K, dt = 65536, 2, 200, 0.01
calm = len(sys.argv) > 1 and sys.argv[1] == "calm"
x0, amp, ph = w.c2_inputs(B, n)
if calm: amp = amp * 1e-3; x0 = x0 * np.array([[1e-3],[1e-3],[1e-3],[1e-3]])
tab = w.circle_controls(amp, ph, w.step_times(K, dt))
sim = rgp.BatchedGameSim(B, n); sim.reset(dt, 1000000, x0)
buf = sim.alloc(tab.size); buf.upload(tab)
stride = tab.shape[1] * 2 * B
samples = []; stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o))
        except Exception as e:
            samples.append((time.time(), "ERR %s" % e))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
t0 = time.time()
for r in range(150):
    sim.step_device(buf.ptr, stride, K)
sim.sync()
t1 = time.time()
time.sleep(0.5)
stop = True; th.join()
print("run: %.3f s for %d arena-steps -> %.4g /s" % (t1 - t0, 150 * K * B, 150 * K * B / (t1 - t0)))
for ts, o in samples:
    try:
        d = json.loads(o); c = d[sorted(d)[0]]
        keys = [k for k in c if "sclk" in k.lower() or "power" in k.lower() or "Power" in k]
        print("%.2f" % (ts - t0), {k: c[k] for k in keys})
    except Exception:
        print("%.2f" % (ts - t0), o[:200].replace("\n", " "))
