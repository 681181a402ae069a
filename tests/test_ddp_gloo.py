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
        # bf16 wire with fp32 accumulation (all-to-all of shards + all-gather), deferred averaging, no_sync, async heads
        red2 = pkg.GradBucketReducer(wire="bf16", average="defer")
        b2 = [torch.randn(1001), torch.randn(64)]
        l2 = [b.clone() for b in b2]
        with red2.no_sync():
            red2.reduce_bucket(b2[0])
            red2.finish()
        ok_nosync = torch.equal(b2[0], l2[0]) and red2.collectives_issued == 0
        for b in b2:
            red2.reduce_bucket(b)
        lin2 = torch.nn.Linear(4, 2)
        lin2.weight.grad = torch.full_like(lin2.weight, float(rank + 1))
        lin2.bias.grad = torch.full_like(lin2.bias, 1.0)
        red2.reduce_grads(lin2.parameters(), async_op=True)
        red2.finish()
        ok_wire = red2.grad_scale == 0.5 and red2.collectives_issued == 3
        for b, lb in zip(b2, l2):
            allb = [torch.zeros_like(lb) for _ in range(world)]
            dist.all_gather(allb, lb)
            want = torch.stack([a.bfloat16().float() for a in allb]).sum(0)      # SUM (average deferred), bf16 inputs
            ok_wire = ok_wire and torch.allclose(b, want.bfloat16().float(), atol=1e-6)
        ok_wire = ok_wire and torch.allclose(lin2.weight.grad, torch.full_like(lin2.weight, 3.0))
        # gradient accumulation: .grad views inside one flat (what AccumulateGrad adopts from the encoder's backward) are
        # exchanged in place, one collective; scattered .grad tensors go through the packed path; both end as the mean
        red3 = pkg.GradBucketReducer()
        flat = torch.arange(40, dtype=torch.float32) * (rank + 1)
        pa, pb, pc = (torch.nn.Parameter(torch.zeros(n)) for n in (10, 7, 20))
        pa.grad, pb.grad, pc.grad = flat[0:10], flat[12:19], flat[20:40]
        ps = torch.nn.Parameter(torch.zeros(5))
        ps.grad = torch.full((5,), float(rank + 1))
        before = red3.collectives_issued
        red3.reduce_accumulated([(40, [(pa, 0, 10), (pb, 12, 7), (pc, 20, 20)]), (8, [(ps, 0, 5)])])
        ok_acc = red3.collectives_issued - before == 2 and pa.grad.data_ptr() == flat.data_ptr()
        ok_acc = ok_acc and torch.allclose(pb.grad, torch.arange(12, 19, dtype=torch.float32) * 1.5)
        ok_acc = ok_acc and torch.allclose(pc.grad, torch.arange(20, 40, dtype=torch.float32) * 1.5)
        ok_acc = ok_acc and torch.allclose(ps.grad, torch.full((5,), 1.5))
        pd = torch.nn.Parameter(torch.zeros(4))
        pd.grad = torch.full((4,), float(rank))          # a gradient that is NOT at its offset in pa's allocation
        red3.reduce_accumulated([(16, [(pa, 0, 10), (pd, 12, 4)])])
        ok_acc = ok_acc and torch.allclose(pd.grad, torch.full((4,), 0.5)) and \
            torch.allclose(pa.grad, torch.arange(10, dtype=torch.float32) * 1.5)   # already equal on both ranks
        # eval-side all-gather of per-sample logits / labels, uneven shards
        n_mine = 3 + rank
        lg = torch.arange(n_mine * 2, dtype=torch.float32).view(n_mine, 2) + 100 * rank
        lb = torch.arange(n_mine) + 10 * rank
        all_lg, all_lb = pkg.ddp.all_gather_eval(lg, lb)
        ok_gather = all_lg.shape == (7, 2) and all_lb.tolist() == [0, 1, 2, 10, 11, 12, 13] and \
            torch.equal(all_lg[3:], torch.arange(8, dtype=torch.float32).view(4, 2) + 100)
        q.put((rank, ok_buckets, ok_head and ok_nosync and ok_wire and ok_acc and ok_gather, ok_bcast,
               pkg.ddp.shard_batch(11, rank, world)))
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


def _sid_worker(rank, world, port, q):
    """Frozen-prefix SID multi-task model (Siglip2sidafrozen.py:750-803): encoder gradient chunks exchanged from inside
    the backward, decoder / classifier gradients through the asynchronous reduce_grads, 1/world folded into FusedAdamW's
    clip coefficient (average='defer'), two optimizer steps; every rank must end with identical parameters, equal to a
    single-process run over the concatenated batch."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as g
    pkg = g.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = pkg.heads
        cfg = pkg.get_config("hostile")

        def make():
            enc = pkg.SiglipVisionModelHIP(cfg, compute_dtype="fp32")
            enc.load_state_dict(pkg.weights.seeded_state_dict(cfg, 4))
            torch.manual_seed(0)
            return H.SigLIP2MTL(enc, seg_layers=(0, 1, -1), embed_dim=32, freeze_below=1).cuda()

        xs = [pkg.weights.seeded_pixels(2, 42, 42, seed=70 + r).cuda() for r in range(world)]
        ys = [torch.tensor([r, 2 - r]).cuda() for r in range(world)]
        ms = [(pkg.weights.seeded_tensor(f"mask{r}", (2, 1, 42, 42), 1.0) > 0.1).float().cuda() for r in range(world)]
        has = torch.tensor([True, True]).cuda()

        def loss_of(model, r):
            cls, seg = model(xs[r])
            return H.mtl_loss(cls, seg, ys[r], ms[r], has)

        # distributed run: this rank sees only its shard
        model = make()
        pkg.ddp.broadcast_parameters(model, src=0)
        red = pkg.GradBucketReducer(average="defer", max_buckets=2).attach(model.encoder)
        heads = [p for n, p in model.named_parameters() if not n.startswith("encoder.") and p.requires_grad]
        trainable = [p for p in model.parameters() if p.requires_grad]
        opt = pkg.FusedAdamW(trainable, lr=2e-3, weight_decay=0.01, max_grad_norm=0.5, grad_scale=red.grad_scale)
        opt.attach_encoder(model.encoder)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss_of(model, rank).backward()
            red.reduce_grads(heads, async_op=True)
            red.finish()
            opt.step()
        torch.cuda.synchronize()
        n_coll = red.collectives_issued
        mine = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}

        # single-process reference over both shards (mean of the per-rank losses = what averaging gradients computes)
        ref = make()
        ropt = pkg.FusedAdamW([p for p in ref.parameters() if p.requires_grad], lr=2e-3, weight_decay=0.01,
                              max_grad_norm=0.5)
        for _ in range(2):
            ropt.zero_grad(set_to_none=True)
            (sum(loss_of(ref, r) for r in range(world)) / world).backward()
            ropt.step()
        torch.cuda.synchronize()
        worst, who = 0.0, None
        for n, p in ref.named_parameters():
            if n.endswith("k_proj.bias") or n.endswith("in_proj_bias"):
                continue   # d loss / d key-bias is exactly zero (softmax shift invariance): Adam turns the rounding noise
                #            of either run into +-lr steps of arbitrary sign (see tests/test_train_loop_gpu.py)
            d = (mine[n] - p.detach().cpu()).abs().max().item() / (p.detach().abs().max().item() + 1e-12)
            if d > worst:
                worst, who = d, n
        # identical across ranks, bit for bit
        flat = torch.cat([v.reshape(-1) for v in mine.values()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        q.put((rank, worst, who, bool(torch.equal(both[0], both[1])), n_coll))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sid_frozen_prefix_two_ranks_identical_parameters_after_two_steps():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst, who, same, n_coll in res:
        assert same, "ranks ended with different parameters"
        assert worst < 5e-5, f"rank {rank}: {who} differs from the single-process run by {worst}"
        # per step: 2 encoder chunks (max_buckets=2: head+block 1) + 1 flat message for decoder/cls head
        assert n_coll == 2 * 3, n_coll


def _accum_worker(rank, world, port, q):
    """Gradient accumulation across ranks: micro-step 1 under no_sync(), micro-step 2 syncing.  Every
    rank must end with the SAME gradients, equal to one single-process pass over all four micro-batches."""
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
        model = pkg.SiglipVisionModelHIP(cfg, compute_dtype="fp32")
        model.load_state_dict(pkg.weights.seeded_state_dict(cfg, 4))
        model = model.to("cuda")
        head = torch.nn.Linear(cfg.hidden_size, 1).cuda()
        torch.manual_seed(0)
        with torch.no_grad():
            head.weight.copy_(pkg.weights.seeded_tensor("acc_head", tuple(head.weight.shape), 0.1).cuda())
            head.bias.zero_()
        red = pkg.GradBucketReducer(max_buckets=2).attach(model)
        xs = [[pkg.weights.seeded_pixels(2, 42, 42, seed=90 + 10 * r + m).cuda() for m in range(2)] for r in range(world)]

        def loss_of(x):
            out = model(pixel_values=x, output_hidden_states=True, interpolate_pos_encoding=True)
            return (head(out.pooler_output).square().sum() + out.hidden_states[1].mean()) / 2      # / accumulation steps

        for step in range(2):                                       # two optimizer steps' worth, to reuse cached buckets
            for p in list(model.parameters()) + list(head.parameters()):
                p.grad = None
            with red.no_sync():
                loss_of(xs[rank][0]).backward()
                red.reduce_grads(head.parameters())                 # no-op under no_sync
            issued = red.collectives_issued
            loss_of(xs[rank][1]).backward()
            red.reduce_grads(head.parameters())
            n_coll = red.collectives_issued - issued
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().clone() for n, p in list(model.named_parameters()) + list(head.named_parameters())}
        flat = torch.cat([v.reshape(-1) for v in got.values()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        model._grad_reducer = None
        for p in list(model.parameters()) + list(head.parameters()):
            p.grad = None
        total = 0
        for r in range(world):
            for m in range(2):
                total = total + loss_of(xs[r][m]) / world
        total.backward()
        torch.cuda.synchronize()
        worst = 0.0
        for n, p in list(model.named_parameters()) + list(head.named_parameters()):
            ref = p.grad.detach().cpu()
            worst = max(worst, ((got[n] - ref).abs().max() / (ref.abs().max() + 1e-12)).item())
        q.put((rank, worst, bool(torch.equal(both[0], both[1])), n_coll))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gradient_accumulation_with_no_sync_exchanges_the_accumulated_sum():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst, same, n_coll in res:
        assert same, "ranks hold different gradients after the syncing micro-step"
        assert worst < 2e-5, f"rank {rank}: accumulated + averaged gradients differ from the joined batch by {worst}"
        assert n_coll == 3, n_coll         # 2 encoder chunks (in place, from the engine callback) + the head message
