"""Phase split of the slowest tiles from a diagnostic build's dump (-DRTX_PROFILE, RTX950_PROF_DUMP=file read by the diagnostic build only).  Usage: python tools/prof_dump.py <dump>"""
import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 40)
gx = 480
names = ["raygen","pre","cull","cand","contested","resolve","shade","shadow_post","childsetup","other"]
tot = d[:, :10].astype(np.int64).sum(1)
def show(t):
    r = d[t]
    print(f"tile ({t%gx},{t//gx}) total {tot[t]}: " + " ".join(f"{n}={int(v)}" for n, v in zip(names, r[:10])) + f" | cand evals {r[16]} exact-waves {r[17]} live {r[21]} queries {r[22]} rays {r[23]} | q by kind {r[24:32].tolist()} cand by kind {r[32:40].tolist()}")
top = np.argsort(-tot)[:6]
for t in top: show(t)
print("portal region:")
reg = [t for t in range(len(tot)) if 215 <= t % gx <= 250 and 95 <= t // gx <= 130 and tot[t] > 0]
reg.sort(key=lambda t: -tot[t])
for t in reg[:8]: show(t)
