import sys, numpy as np
sys.path.insert(0, '/root/repo')
import raytracing_amd as rt, torch
W, H = 3840, 2160
s = rt.synthetic_scene()
ctx = rt.Context(0); ctx.upload(s)
out = torch.zeros((H, W), dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
f = s.frame(W, H)
for _ in range(20):
    ctx.render_device(f, out.data_ptr())
ctx.sync()
c = ctx.tile_costs().astype(np.int64)
tx = (W + 7) // 8
nz = c[c > 0]
print(f"traced tiles {len(nz)} avg {nz.mean():.0f} p50 {np.percentile(nz,50):.0f} p90 {np.percentile(nz,90):.0f} p99 {np.percentile(nz,99):.0f} p99.9 {np.percentile(nz,99.9):.0f} max {nz.max()} sum {nz.sum()/1e6:.1f}M")
top = np.argsort(-c)[:16]
print("   top:", [(int(t % tx), int(t // tx), int(c[t])) for t in top])
h, e = np.histogram(nz, bins=[0, 8e3, 16e3, 32e3, 64e3, 128e3, 256e3, 512e3, 1e6, 2e6, 4e6, 1e9]); print("   hist", h.tolist())
