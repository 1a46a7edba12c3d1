"""Per-shape timing of one canonical step (HIP events per launch), sorted by time."""
import os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dep_gan_im_amd as dg
from bench import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
netG = dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=1)
netD1 = dg.Dis_C2D_FCN1((256, 256, 1), seed=2)
netD2 = dg.Dis_C2D_FCN1((256, 256, 1), seed=3)
tr = dg.build_trainers(netG, netD1, netD2, batchSize=B, delta=10.0, lrD=1e-4, lrG=1e-4, IM_TRSH=0.5, device=dev)
x, y2, z, ep = [torch.from_numpy(a).to(dev) for a in synth(1000, B)]
def step():
    tr.netD_y2_train([y2, x, z, ep]); tr.netD_dem_train([y2, x, z, ep]); tr.netG_train([x, y2, z])
for _ in range(2): step()
eng = tr.engine
eng.profile(True); eng.profile_reset()
N = 3
for _ in range(N): step()
path = "gpurun_out/labels.csv"
eng.profile_dump(path)
agg = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    k = (r["class"], r["label"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
tot = sum(a[1] for a in agg.values())
print("total %.2f ms/step" % (tot / N))
for (k, l), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("c%s %-34s n/step %5.1f  %7.3f ms/step  %6.1f TF/s  %4.1f%%" % (k, l, a[0] / N, a[1] / N, a[2] / a[1] if a[1] else 0, 100 * a[1] / tot))
