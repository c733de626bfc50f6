"""Items per level of one wavefront frame (rt_debug_wavefront_items): count, lanes, cycles; tile costs of level 0 and of the tile walk.  Usage (GPU box): python tools/wf_items.py [4k|1080p]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import raytracing_amd as rt
wl = sys.argv[1] if len(sys.argv) > 1 else "4k"
W, H = (3840, 2160) if wl == "4k" else (1920, 1080)
s = rt.default_scene()
ctx = rt.Context(0); ctx.upload(s)
import torch
out = torch.zeros((H, W), dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
f = s.frame(W, H)
ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
for _ in range(5):
    ctx.render_device(f, out.data_ptr())
ctx.sync()
for lvl in (1, 2, 3):
    it = ctx.wavefront_items(lvl)
    cyc = it[:, 0].astype(np.int64)
    pop = np.array([bin(int(a) | (int(b) << 32)).count("1") for a, b in zip(it[:, 2], it[:, 3])])
    kind = it[:, 1] & 1
    print(f"level {lvl}: items {len(it)} lanes avg {pop.mean():.1f} refl {int((kind==0).sum())} refr {int((kind==1).sum())} cycles avg {cyc.mean():.0f} p50 {np.percentile(cyc,50):.0f} p90 {np.percentile(cyc,90):.0f} p99 {np.percentile(cyc,99):.0f} max {cyc.max()} children {int(it[:,6].sum())}")
    top = np.argsort(-cyc)[:5]
    for t in top:
        print("   ", cyc[t], "lanes", pop[t], "kind", kind[t], "tile", it[t,4] & 0x7fff, it[t,4] >> 16, "children", it[t,6])
    h, e = np.histogram(pop, bins=[0,2,4,8,16,32,48,64,65]); print("   lanes hist", h.tolist())
costs = ctx.tile_costs()
c = costs[costs > 0].astype(np.int64)
print("L0 tiles", len(c), "avg", c.mean(), "p99", np.percentile(c, 99), "max", c.max(), "sum", c.sum())
ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 0)
for _ in range(5):
    ctx.render_device(f, out.data_ptr())
ctx.sync()
costs = ctx.tile_costs()
c = costs[costs > 0].astype(np.int64)
print("classic tiles", len(c), "avg", c.mean(), "p99", np.percentile(c, 99), "max", c.max(), "sum", c.sum())
