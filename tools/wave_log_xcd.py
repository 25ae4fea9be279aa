#!/usr/bin/env python3
"""Per-XCD view of the asynchronous pool's per-wavefront log (gpurun_out/wave_log.npy, written by
`RGS_LIB_SUFFIX=_stamps python bench.py --stamps ...`): workgroups are dispatched round-robin over the 8 XCDs, so
workgroup w runs on XCD w % 8.  Prints kernel time and lane-pass time per XCD."""
import sys
import numpy as np

wl = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wave_log.npy").astype(float)
wg = wl.reshape(-1, 4, 8)
tot = wg[:, :, 0].max(1)
lane = wg[:, :, 1].sum(1) / np.maximum(1.0, wg[:, :, 3].sum(1))
rounds = wg[:, :, 4].sum(1)
print("workgroups %d; kernel ticks per workgroup: mean %.4g max %.4g (max/mean %.3f)" % (len(tot), tot.mean(), tot.max(), tot.max() / tot.mean()))
print("XCD  kernel ticks (mean)  ticks per lane pass  rounds per workgroup")
for x in range(8):
    print("%3d  %18.4g  %19.0f  %20.1f" % (x, tot[x::8].mean(), lane[x::8].mean(), rounds[x::8].mean()))
print("corr(kernel ticks, ticks per lane pass) = %.2f; corr(kernel ticks, rounds) = %.2f" % (np.corrcoef(tot, lane)[0, 1], np.corrcoef(tot, rounds)[0, 1]))
