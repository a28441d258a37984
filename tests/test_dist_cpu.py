"""world_size-2 gloo test of the multi-GPU path's host logic (sharding, barrier, max-over-ranks
timing, detection gather).  The data path itself has no collective (images shard by rank)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from ctdet import dist as cdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, l, w = cdist.init('gloo')
    assert (r, w) == (rank, world)
    b, e = cdist.shard(7, rank, world)
    cdist.barrier('cpu')
    t = cdist.max_over_ranks(0.010 * (rank + 1))
    n = cdist.sum_over_ranks(e - b)
    dets = [[np.full((rank + 1, 5), i, np.float32)] for i in range(b, e)]
    allb = cdist.gather_detections(dets)
    q.put((rank, (b, e), t, n, None if allb is None else [int(a[0][0, 0]) for a in allb]))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 4) and res[1][1] == (4, 7)            # contiguous, sizes differ by <= 1
    assert abs(res[0][2] - 0.020) < 1e-12 and abs(res[1][2] - 0.020) < 1e-12   # max over ranks
    assert res[0][3] == 7 and res[1][3] == 7
    assert res[0][4] == list(range(7)) and res[1][4] is None        # global image order on rank 0


def test_shard_covers_everything():
    for total in (1, 5, 32, 33, 257):
        for world in (1, 2, 3, 8):
            spans = [cdist.shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    cdist.init('gloo')
    numels = [5, 1000, 3, 70000, 12, 40000, 7]
    # the training engine's mode: the bucketer reduces slices of the caller's own flat buffer (gradient arena) and
    # calls before_launch() exactly once per bucket, right before its all-reduce (stream join point)
    arena = torch.full((sum(numels) + 9,), float('nan'))
    bk = cdist.GradBucketer(numels, 'cpu', bucket_bytes=200000, flat=arena)   # 50k floats per bucket -> several buckets
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(n, generator=g) for n in numels]
    joins = []
    bk.begin()
    for i, t in enumerate(grads):
        bk.view(i).copy_(t)
        bk.ready(i, lambda i=i: joins.append(i))
    bk.finish()
    assert bk.flat.data_ptr() == arena.data_ptr() and torch.isnan(arena[sum(numels):]).all()
    assert joins == sorted(bk.last_in_bucket.values()), (joins, bk.last_in_bucket)
    n = cdist.global_normalizer(torch.tensor(10 + 5 * rank), 'cpu')       # integer count, odd global sum
    q.put((rank, [bk.view(i).numpy().copy() for i in range(len(numels))], len(bk.bucket_span), float(n)))
    torch.distributed.destroy_process_group()


def test_grad_bucketer_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    numels = [5, 1000, 3, 70000, 12, 40000, 7]
    want = []
    for i, n in enumerate(numels):
        gs = []
        for rank in range(world):
            g = torch.Generator().manual_seed(100 + rank)
            gs.append([torch.randn(m, generator=g) for m in numels][i])
        want.append(sum(gs) / world)
    for rank in range(world):
        assert res[rank][2] >= 2                                 # really bucketed
        assert res[rank][3] == 12.5                              # (10 + 15) / 2, not truncated
        for a, b in zip(res[rank][1], want):
            assert np.allclose(a, b.numpy(), atol=1e-6)


class _FakeNet:
    """Stands in for RFBNet in the 2-rank init_reweight test: returns prepared conf tensors for init=True."""

    def __init__(self, confs, C, T):
        self.confs, self.i = confs, 0
        self.OBJ_Target = torch.nn.Linear(C, T, bias=False)

    def _device(self):
        return torch.device('cpu')

    def __call__(self, x, init=False):
        assert init
        self.i += 1
        return self.confs[self.i - 1]


def _reweight_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    cdist.init('gloo')
    from ctdet import ops, reweight
    g = torch.Generator().manual_seed(5)
    P, C, ncls = 50, 12, 7
    confs = [torch.randn(2, P, C, generator=g) for _ in range(4)]
    labels = [torch.randint(0, ncls, (2, P), generator=g).float() for _ in range(4)]
    mine = list(range(rank, 4, world))                     # each rank sees its own batches
    it = iter([labels[i] for i in mine])
    real = ops.match_batched
    ops.match_batched = lambda *a, **k: (None, torch.stack([next(it), torch.ones(2, P)], 2), None)   # no HIP here
    try:
        net = _FakeNet([confs[i] for i in mine], C, ncls - 1)
        batches = [(torch.zeros(2, 3, 4, 4), [torch.zeros(1, 6)] * 2) for _ in mine]
        w = reweight.init_reweight(net, torch.zeros(P, 4), batches, ncls)
    finally:
        ops.match_batched = real
    s = c = 0
    for cf, lb in zip(confs, labels):                      # single-process answer over ALL batches
        a, b = reweight.class_feature_sums(cf, lb, ncls)
        s, c = s + a, c + b
    want = reweight.weights_from_sums(s, c)
    q.put((rank, float((w - want).abs().max())))
    torch.distributed.destroy_process_group()


def test_init_reweight_reduces_over_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_reweight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(err < 1e-6 for _, err in res), res
