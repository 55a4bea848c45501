"""CPU, world_size=2 (gloo): the bucketed gradient reducer.  N-rank step on a sharded batch must equal the
1-rank step on the concatenated batch, with gradients living in the flat buffer."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(24, 64), nn.SiLU(), nn.Linear(64, 64), nn.SiLU(), nn.Linear(64, 8))


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adm_amd.optim import BucketedGradReducer, FlatParams
    m = _model()
    flat = FlatParams(m)
    red = BucketedGradReducer(flat, bucket_bytes=2 * 1024)       # several buckets
    assert len(red.buckets) >= 3
    assert sum(hi - lo for lo, hi, _ in red.buckets) == flat.numel
    torch.manual_seed(1)
    x, y = torch.randn(16, 24), torch.randn(16, 8)
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    for _ in range(2):                                           # two steps: the hook state must reset
        flat.zero_grad()
        loss = ((m(xs) - ys) ** 2).sum() / 8                     # per-rank mean over its shard
        loss.backward()
        red.finish()
    g = flat.grad / world                                        # the optimiser's grad_scale
    for p, o in zip(flat.params, flat.offsets):                  # grads still are views of the flat buffer
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    if rank == 0:
        torch.save(g.clone(), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_equals_single_rank(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g2 = torch.load(out, weights_only=True)
    from adm_amd.optim import FlatParams
    m = _model()
    flat = FlatParams(m)
    torch.manual_seed(1)
    x, y = torch.randn(16, 24), torch.randn(16, 8)
    loss = ((m(x) - y) ** 2).sum() / 16
    loss.backward()
    torch.testing.assert_close(g2, flat.grad, rtol=1e-5, atol=1e-6)
