"""Rank bodies of tests/test_gpu_distributed.py.  They run in processes forked from a forkserver that was started before
pytest touched the GPU (tests/conftest.py), so each rank is a clean process that initialises the GPU itself."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, port, backend, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)                   # one-GPU box: both ranks share the card (gloo; RCCL needs one GPU each)
        dist.init_process_group(backend, rank=rank, world_size=world)
        from stratanet2_vegetation_coverage_maps_amd import PointNet2, losses, project_to_plotwise_coverages
        from stratanet2_vegetation_coverage_maps_amd.optim import FlatAdam, allreduce_flat_grad, flatten_parameters, shard_of_rank
        from stratanet2_vegetation_coverage_maps_amd.synthetic import make_args, make_batch
        N, per_rank = 2048, 2
        args = make_args(cuda=0, subsample_size=N, ratio1=0.125, r1=1.0, ratio2=0.25, r2=2.0)
        torch.manual_seed(0)                       # identical initial weights on every rank, as bench.py does
        model = PointNet2(args).train()
        flatten_parameters(model)
        opt = FlatAdam(model, lr=1e-3, weight_decay=1e-3, world_size=world)

        def shard_grad(r):
            first, n = shard_of_rank(r, per_rank)
            d = make_batch(n, N, first_plot=first)
            d["fps_start"] = torch.zeros(2, n, dtype=torch.long)
            opt.zero_grad()
            cov, proba = model(d)
            pred = project_to_plotwise_coverages(cov, d["cloud"], args, model=model)
            loss, _ = losses.total_loss(pred, proba, d["coverages"].cuda(), d["pdf_all"].cuda(), args.m, args.e)
            loss.backward()
            return model._last_flat_grad

        singles = [shard_grad(r).clone() for r in range(world)]          # what G independent single-GPU runs produce
        g = shard_grad(rank)
        scale = allreduce_flat_grad(g, world)                            # the ONE exchange of the data-parallel path
        mean = torch.stack(singles).mean(0)
        err = float(((g * scale) - mean).abs().max() / mean.abs().max())
        for _ in range(2):                                               # two exchanged steps on fresh gradients
            shard_grad(rank)                                             # (opt.step() does the exchange itself)
            opt.step()
        params = model._flat_params.clone()
        gathered = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(gathered, params)
        drift = max(float((gathered[0] - t).abs().max()) for t in gathered[1:])
        q.put((rank, "ok", err, drift, int(g.numel())))
        dist.destroy_process_group()
    except Exception as exc:                                             # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), 0.0, 0))


def run_command(cmd, env, q):
    """Run a command from a clean (never touched a GPU) process and hand back (rc, stdout, stderr)."""
    import subprocess
    e = dict(os.environ)
    e.update(env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, cwd=ROOT)
    q.put((r.returncode, r.stdout, r.stderr[-4000:]))
