import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
N = 4096
args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
d = make_batch(2, N, first_plot=9)
d["fps_start"] = torch.tensor([[5, 77], [3, 1]])
m = PointNet2(args); m.load_state_dict(network.init_state_dict(2)); m.eval()
with torch.no_grad():
    xyz = d["xyz"].cuda(); fs = d["fps_start"].to(torch.int32).cuda()
    g1 = m._geometry(xyz, fs); g2 = m._geometry(xyz, fs)
    for k in ("idx1", "cnt1", "ord1", "ord2", "idx2", "cnt2"):
        print(k, torch.equal(getattr(g1, k), getattr(g2, k)))
    print("knn1", torch.equal(g1.knn1[0], g2.knn1[0]), torch.equal(g1.knn1[1], g2.knn1[1]))
    outs = []
    for rep in range(3):
        cov, proba, s = m._forward_impl(xyz, d["cloud"].cuda(), fs, False)
        outs.append((cov.clone(), s.ext1.clone(), s.arg1.clone(), s.x1.clone(), s.ext2.clone(), s.x2.clone(), s.h2.clone(), s.h1.clone()))
    names = ["cov", "ext1", "arg1", "x1", "ext2", "x2", "h2", "h1"]
    for i, n in enumerate(names):
        print(n, torch.equal(outs[0][i], outs[1][i]), torch.equal(outs[0][i], outs[2][i]), (outs[0][i].float() - outs[1][i].float()).abs().max().item())
    print("order head", g1.ord1[-4:].tolist(), "M1", g1.M1, "nsolo-ish", int((g1.cnt1 > 64).sum()))
print("---- prefetch path")
with torch.no_grad():
    cov_a, proba_a = m(d)
    geo = m.prefetch_geometry(d)
    torch.cuda.synchronize()
    g0 = m._geometry(xyz, fs)
    for k in ("idx1", "cnt1", "ord1", "ord2", "idx2", "cnt2", "nbr1", "pos1_aos", "pos2_aos"):
        a, b = getattr(g0, k), getattr(geo, k)
        if k.startswith("nbr"):
            mask = torch.arange(a.shape[1], device=a.device)[None] < g0.cnt1[:, None]
            print(k, torch.equal(a[mask], b[mask]))
        else:
            print(k, a.shape == b.shape and torch.equal(a, b))
    print("knn1", torch.equal(g0.knn1[0], geo.knn1[0]), torch.equal(g0.knn1[1], geo.knn1[1]))
    print("knn2", torch.equal(g0.knn2[0], geo.knn2[0]), torch.equal(g0.knn2[1], geo.knn2[1]))
    print("knn3", torch.equal(g0.knn3[0], geo.knn3[0]), torch.equal(g0.knn3[1], geo.knn3[1]))
    cov_b, proba_b = m({"cloud": d["cloud"], "xyz": d["xyz"], "geometry": geo})
    print("cov equal", torch.equal(cov_a, cov_b), (cov_a - cov_b).abs().max().item())
