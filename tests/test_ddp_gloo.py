"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (bucket all-reduce + 1/world averaging,
parameter broadcast, batch sharding).  The encoder kernels are not involved: the reducer is exercised with the
same flat fp32 buckets the encoder's backward hands to it."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as g
    pkg = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        red = pkg.GradBucketReducer()
        # three "block" buckets, reduced one by one as the backward would, then finish()
        buckets = [torch.randn(1000 + 17 * i) for i in range(3)]
        local = [b.clone() for b in buckets]
        for b in buckets:
            red.reduce_bucket(b)
        red.finish()
        gathered = []
        for i, lb in enumerate(local):
            allb = [torch.zeros_like(lb) for _ in range(world)]
            dist.all_gather(allb, lb)
            gathered.append(torch.stack(allb).mean(0))
        ok_buckets = all(torch.allclose(b, gexp, atol=1e-6) for b, gexp in zip(buckets, gathered))
        # head parameters outside the encoder
        lin = torch.nn.Linear(8, 3)
        lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
        lin.bias.grad = torch.full_like(lin.bias, float(10 * (rank + 1)))
        red.reduce_grads(lin.parameters())
        ok_head = torch.allclose(lin.weight.grad, torch.full_like(lin.weight, 1.5)) and \
            torch.allclose(lin.bias.grad, torch.full_like(lin.bias, 15.0))
        # broadcast
        m = torch.nn.Linear(5, 5)
        pkg.ddp.broadcast_parameters(m, src=0)
        ws = [torch.zeros_like(m.weight) for _ in range(world)]
        dist.all_gather(ws, m.weight.data)
        ok_bcast = torch.equal(ws[0], ws[1])
        q.put((rank, ok_buckets, ok_head, ok_bcast, pkg.ddp.shard_batch(11, rank, world)))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1:4] == (True, True, True) and res[1][1:4] == (True, True, True)
    assert res[0][4] == (0, 6) and res[1][4] == (6, 11)      # whole images, contiguous, covering the batch


def test_shard_batch_covers_everything():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    pkg = g.load_package()
    for gb in (1, 7, 64, 513):
        for world in (1, 2, 4, 8):
            spans = [pkg.ddp.shard_batch(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------------------
# GPU: the encoder's backward really calls the reducer per bucket; two ranks share the one GPU of the test box and
# exchange over gloo (RCCL refuses two ranks on one device), which exercises exactly the hook plumbing bench.py uses.
import pytest  # noqa: E402


def _gpu_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as g
    pkg = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pkg.get_config("hostile")
        sd = pkg.weights.seeded_state_dict(cfg, 4)
        model = pkg.SiglipVisionModelHIP(cfg, compute_dtype="fp32")
        model.load_state_dict(sd)
        model = model.to("cuda")
        pkg.GradBucketReducer().attach(model)
        xs = [pkg.weights.seeded_pixels(2, 42, 42, seed=50 + r).cuda() for r in range(world)]
        out = model(pixel_values=xs[rank], output_hidden_states=True, interpolate_pos_encoding=True)
        (out.pooler_output.square().sum() + out.hidden_states[1].mean()).backward()
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        # single-process reference: mean over the two ranks' losses
        model._grad_reducer = None
        for p in model.parameters():
            p.grad = None
        total = 0
        for r in range(world):
            o = model(pixel_values=xs[r], output_hidden_states=True, interpolate_pos_encoding=True)
            total = total + (o.pooler_output.square().sum() + o.hidden_states[1].mean()) / world
        total.backward()
        torch.cuda.synchronize()
        worst = 0.0
        for n, p in model.named_parameters():
            ref = p.grad.detach().cpu()
            worst = max(worst, ((got[n] - ref).abs().max() / (ref.abs().max() + 1e-12)).item())
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_encoder_backward_all_reduces_every_bucket_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst in res:
        assert worst < 2e-5, f"rank {rank}: averaged gradients differ from the single-process mean by {worst}"
