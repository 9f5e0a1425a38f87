"""Worker of tests/test_gpu_parallel.py::test_sync_bn_two_ranks_equal_the_global_batch: one of 2 ranks sharing ONE MI355X (gloo).

--sync_bn (the reference: apex.parallel.convert_syncbn_model, train.py:65-68) on the TRAINABLE ResNet extractor: each rank holds half
of a batch of images; with sync_bn the train-mode BatchNorm statistics, the running statistics, the features and — summed over the ranks —
the parameter gradients must equal ONE process running the whole batch; without it they must not (negative control).  "Equal": see
the comment at the assertions — exact where no bf16 rounding sits between the statistic and its inputs, to the resolution of a bf16
network's run-to-run reproducibility elsewhere."""
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
    probe = torch.randn(n, 16, 256, generator=g)
    lo, hi = rank * n // 2, (rank + 1) * n // 2
    fulls = [run(imgs, probe, False) for _ in range(2)]                      # ONE process, the whole batch (twice: see below)

    def compare(mine, full):
        feat, grads, stats = mine
        full_feat, full_grads, full_stats = full
        rel = lambda k: float((stats[k] - full_stats[k]).abs().max() / full_stats[k].abs().max())
        e_feat = float((feat - full_feat[lo:hi]).abs().max() / full_feat.abs().max())
        e_stem = max(rel(k) for k in stats if k.startswith('1.'))           # the stem's BatchNorm: its input is the pixels themselves
        e_stat = max(rel(k) for k in stats)
        num = den = 0.0
        for k, gl in grads.items():
            ref = full_grads[k].cpu()
            num += float((gl - ref).double().pow(2).sum())
            den += float(ref.double().pow(2).sum())
        return (e_feat, e_stem, e_stat, (num / den) ** 0.5)

    def ranks_summed(grads):
        out = {}
        for k, gl in grads.items():
            gs = gl.cpu()
            dist.all_reduce(gs)                       # sum over ranks of d(sum of the per-rank losses) = the whole batch's gradient
            out[k] = gs
        return out

    res = {}
    for sync, times in ((True, 3), (False, 1)):
        res[sync] = []
        for _ in range(times):
            feat, grads, stats = run(imgs[lo:hi], probe[lo:hi], sync)
            mine = (feat, ranks_summed(grads), stats)
            res[sync] += [compare(mine, f) for f in fulls]
    worst = tuple(max(r[i] for r in res[True]) for i in range(4))
    best = min(res[True], key=lambda r: r[2] + r[3])
    ns = min(res[False], key=lambda r: r[2] + r[3])
    print(f'rank {rank}: sync_bn, best of {len(res[True])} pairings: features {best[0]:.2e} stem statistics {best[1]:.2e} all running '
          f'statistics {best[2]:.2e} gradient L2 {best[3]:.2e}; worst: {worst[0]:.2e} {worst[1]:.2e} {worst[2]:.2e} {worst[3]:.2e}; '
          f'without: features {ns[0]:.2e} stem statistics {ns[1]:.2e} all {ns[2]:.2e} gradient L2 {ns[3]:.2e}', flush=True)
    # Two bf16 networks are compared.  Usually they agree bit for bit in the features (statistics 1e-6: fp32 sums of the same values
    # in another order; gradients 5e-5) — but a statistic that differs in its 7th digit (float atomics land in run-to-run order even
    # in ONE process) can flip a bf16 rounding behind the stem, and through seven convolutions that flip becomes a one-ulp perturbation
    # of everything: features 6e-3, deep statistics 8e-4, gradients 7e-2.  Measured between two IDENTICAL single-process runs (one
    # run in eight with two processes on the card); it failed this test's first form, which allowed "a few bf16 ulps on single
    # elements, never more".  Hence: (1) the stem's statistics — pixels in, nothing to amplify — must match in EVERY pairing;
    # (2) three sync_bn runs against two whole-batch runs: the BEST pairing must meet the strict bars (six unlucky draws in a row:
    # < 1e-4), (3) every pairing stays 10 x inside the unsynchronised network's distance (negative control).
    assert worst[1] <= 5e-6, worst
    assert best[2] <= 2e-5 and best[0] <= 2e-2 and best[3] <= 2e-2, best
    assert worst[0] <= 3e-2 and worst[2] <= 5e-3 and worst[3] <= 0.15, worst
    assert ns[1] >= 1e-3 and ns[2] >= 5e-2 and ns[3] >= 0.3, (best, ns)
    print(f'rank {rank}: sync_bn == one process on the global batch', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
