import sys
import torch
sys.path.insert(0, ".")
from oracle import network
from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
from stratanet2_vegetation_coverage_maps_amd import hip_ops as ops
from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
N, B = 4096, 2
args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
model = PointNet2(args)
model.load_state_dict(network.init_state_dict(5))
model = model.cuda().train()
d = make_batch(B, N, first_plot=40)
d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
d["fps_start"] = torch.zeros(2, B, dtype=torch.int32, device="cuda")
orig = ops.head_bn_sums
MODE = {"m": "normal"}
def head_bn_sums(hd, gamma, beta, dgamma, dbeta, ok):
    if MODE["m"] == "normal":
        return orig(hd, gamma, beta, dgamma, dbeta, ok)
    if MODE["m"] == "skip":          # FP1 backward sees dgamma = dbeta = 0 and believes the sums are done
        ok.fill_(1)
        return None
    if MODE["m"] == "declined":      # the shortcut declines: the ordinary row pass computes the sums
        ok.fill_(0)
        return None
ops.head_bn_sums = head_bn_sums
def grads():
    model.zero_grad()
    cov, proba = model(d)
    pred = project_to_plotwise_coverages(cov, d["cloud"], args)
    loss, _ = losses.total_loss(pred, proba, d["coverages"], d["pdf_all"], args.m, args.e)
    loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
keys = ["sa1_module.conv.local_nn.0.0.weight", "fp1_module.nn.0.0.weight", "fp1_module.nn.0.2.weight", "lin1.weight"]
MODE["m"] = "declined"
ref = grads()
for mode in ["normal"] * 6 + ["skip", "declined"]:
    MODE["m"] = mode
    g = grads()
    print(mode.ljust(9), "  ".join("%s %.3e" % (k.split(".")[0] + "." + k.split(".")[-2] + "." + k.split(".")[-1], float((g[k] - ref[k]).abs().max())) for k in keys))
