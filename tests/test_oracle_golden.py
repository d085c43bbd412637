"""The oracle's restatement of the reference glue (oracle/network.py, projection.py, losses.py) against goldens
produced by running the reference's own code (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, golden_args, golden_state_dict, load_golden
from oracle import losses, network, projection


def _inputs(g):
    cloud, xyz = torch.from_numpy(g["in/cloud"]), torch.from_numpy(g["in/xyz"])
    st = torch.from_numpy(g["in/fps_start"])
    return cloud, xyz, (st[0], st[1])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_eval_forward_and_projections(name):
    g, args = load_golden(name), golden_args(name)
    sd = golden_state_dict(g)
    cloud, xyz, st = _inputs(g)
    with torch.no_grad():
        cov, proba, _ = network.forward(sd, cloud, xyz, args, training=False, fps_start=st)
        pred = projection.project_to_plotwise_coverages(cov, cloud, args)
    np.testing.assert_allclose(cov.numpy(), g["eval/coverages_pointwise"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(proba.numpy(), g["eval/proba_pointwise"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(pred.numpy(), g["eval/pred_coverages"], atol=2e-6, rtol=0)
    B, N = cloud.shape[0], cloud.shape[2]
    cov_b = cov.view(B, N, 4).permute(0, 2, 1)
    for b in range(B):
        r = projection.project_to_2d_rasters(cloud[b], cov_b[b], args)
        ref = g["eval/rasters"][b]
        assert r.dtype == np.float64 and r.shape == ref.shape
        assert np.array_equal(np.isnan(r), np.isnan(ref)), "occupied-pixel mask must be bit-exact"
        np.testing.assert_allclose(np.nan_to_num(r), np.nan_to_num(ref), atol=2e-6, rtol=0)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_train_forward_loss_backward(name):
    g, args = load_golden(name), golden_args(name)
    sd = golden_state_dict(g)
    keys = network.param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    cloud, xyz, st = _inputs(g)
    cov, proba, ex = network.forward(sd, cloud, xyz, args, training=True, fps_start=st)
    pred = projection.project_to_plotwise_coverages(cov, cloud, args)
    loss, (l_abs, l_log, l_e) = losses.total_loss(pred, proba, torch.from_numpy(g["in/coverages"]),
                                                  torch.from_numpy(g["in/pdf_all"]), args.m, args.e)
    loss.backward()
    np.testing.assert_allclose(cov.detach().numpy(), g["train/coverages_pointwise"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(pred.detach().numpy(), g["train/pred_coverages"], atol=2e-6, rtol=0)
    np.testing.assert_allclose([loss.item(), l_abs.item(), l_log.item(), l_e.item()], g["train/losses"],
                               atol=1e-6, rtol=1e-6)
    for k in keys:
        ref = g[f"grad/{k}"]
        np.testing.assert_allclose(sd[k].grad.numpy(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), rtol=0,
                                   err_msg=k)
    for k, v in ex["new_stats"].items():
        np.testing.assert_allclose(v.numpy(), g[f"sd_after/{k}"], atol=1e-6, rtol=1e-5, err_msg=k)


def test_init_state_dict_matches_reference_constructor():
    """Default init under manual_seed(0), key names and shapes; only the Linear weights and the
    lin2 bias are compared, the goldens' BN tensors were perturbed on purpose."""
    g = load_golden("c1_ref_defaults")
    sd = network.init_state_dict(0)
    ref = golden_state_dict(g)
    assert list(sd.keys()) == list(ref.keys())
    for k in sd:
        assert sd[k].shape == ref[k].shape, k
        if ".0.weight" in k or ".0.bias" in k or k.startswith("lin"):
            assert torch.equal(sd[k], ref[k]), k
    assert sum(sd[k].numel() for k in network.param_keys(sd)) == 14997
