"""Worker of tests/test_gpu_parallel.py::test_sync_bn_two_ranks_equal_the_global_batch: one of 2 ranks sharing ONE MI355X (gloo).

--sync_bn (the reference: apex.parallel.convert_syncbn_model, train.py:65-68) on the TRAINABLE ResNet extractor: each rank holds half
of a batch of images; with sync_bn the train-mode BatchNorm statistics, the running statistics, the features and — summed over the ranks —
the parameter gradients must equal ONE process running the whole batch; without it they must not (negative control)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svol_amd import synthetic as syn  # noqa: E402
from svol_amd.modeling.resnet import ResNetExtractor  # noqa: E402

DEPTHS, WIDTHS = (1, 1, 1), (64, 128, 256)


def run(imgs, probe, sync):
    net = ResNetExtractor(DEPTHS, WIDTHS, compute_dtype='bf16', trainable=True, sync_bn=sync)
    net.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes(DEPTHS, WIDTHS), seed=3))
    net = net.cuda().train()
    feat = net(imgs.cuda())
    (feat.float() * probe.cuda()).sum().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().clone() for k, p in net.named_parameters()}
    stats = {k: v.detach().float().clone() for k, v in net.state_dict().items() if 'running_' in k}
    return feat.detach().float(), grads, stats


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert world == 2
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n = 4
    imgs = syn.synth_images(n, syn.vit_config(image_size=64), seed=9)
    g = torch.Generator().manual_seed(5)
    full_feat, full_grads, full_stats = run(imgs, torch.randn(n, 16, 256, generator=g), False)   # ONE process, the whole batch
    g = torch.Generator().manual_seed(5)
    probe = torch.randn(n, 16, 256, generator=g)
    lo, hi = rank * n // 2, (rank + 1) * n // 2
    out = {}
    for sync in (True, False):
        feat, grads, stats = run(imgs[lo:hi], probe[lo:hi], sync)
        e_feat = float((feat - full_feat[lo:hi]).abs().max() / full_feat.abs().max())
        e_stat = max(float((stats[k] - full_stats[k]).abs().max() / full_stats[k].abs().max()) for k in stats)
        num = den = 0.0
        worst = 0.0
        for k, gl in grads.items():
            gs = gl.cpu()
            dist.all_reduce(gs)                       # sum over ranks of d(sum of the per-rank losses) = the whole batch's gradient
            ref = full_grads[k].cpu()
            num += float((gs - ref).double().pow(2).sum())
            den += float(ref.double().pow(2).sum())
            worst = max(worst, float((gs - ref).norm() / ref.norm().clamp_min(1e-12)))
        out[sync] = (e_feat, e_stat, (num / den) ** 0.5, worst)
    s, ns = out[True], out[False]
    print(f'rank {rank}: sync_bn: features {s[0]:.2e} running stats {s[1]:.2e} gradient L2 {s[2]:.2e} (worst parameter {s[3]:.2e}); '
          f'without: features {ns[0]:.2e} running stats {ns[1]:.2e} gradient L2 {ns[2]:.2e}', flush=True)
    # running statistics are fp32 sums of the same bf16 values in another order: ~1e-6; features / gradients carry bf16 activations whose
    # roundings can flip with the 7th digit of a statistic (a few bf16 ulps on single elements), never more
    assert s[1] <= 2e-5 and s[0] <= 2e-2 and s[2] <= 2e-2, s
    # negative control: per-rank statistics of half the batch are somewhere else
    assert ns[1] >= 1e-3 and ns[2] >= 5 * s[2], (s, ns)
    print(f'rank {rank}: sync_bn == one process on the global batch', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
