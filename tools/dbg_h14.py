import sys, types, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_model_gpu as TM
from oracle import tvts_oracle as O
from tvts_amd import arch as A
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m, oarch, P = TM.build(arch=A.small_arch_h(num_frames=16), seed=6)
batch = O.synth_batch(oarch, B=3, T=T, seed=8, n_trans=nt, caption_len=9)
r1, r2, rte, rve, rpred, grads = TM.oracle_step(P, batch, oarch)
l1, l2, te, ve, pred, store = TM.engine_step(m, batch)
print("loss", l1, r1, "ve rel", TM.rel(ve, rve), "te rel", TM.rel(te, rte))
rows = []
for k, g in grads.items():
    mine = store.g(k).detach().cpu()
    rows.append((float(mine.norm()) / (float(g.norm()) + 1e-30), float(g.norm()), k))
rows.sort()
for r in rows[:12] + rows[-6:]:
    print("%.4f %.4f %s" % r)
tot = sum(r[1] ** 2 for r in rows) ** .5
print("tot ref", tot, "mine", float(store.grad.double().norm()))
for pre in ("text", "video_model", "pred"):
    a = sum(float(store.g(k).double().norm()) ** 2 for k in grads if k.startswith(pre)) ** .5
    b = sum(float(grads[k].double().norm()) ** 2 for k in grads if k.startswith(pre)) ** .5
    print(pre, a, b, a / (b + 1e-30))
