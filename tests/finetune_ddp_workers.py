"""Workers of the data-parallel fine-tuning tests (tests/test_finetune_ddp_cpu.py on the torch checker backend, tests/test_finetune_ddp_gpu.py
on the HIP path with every rank on cuda:0): ct_lipro_train.py:75-107 and ct_vocabfine_train.py:62-123 as one process per GPU.

Parity oracle for LiPro at W ranks = the REAL reference's single-process step on the concatenated global batch (tests/golden/finetune_tiny.pt,
B = 2): rank r gets sample r, the head's averaged gradients, the mean of the rank losses and the post-step VQ buffers must equal the golden ones.
VocabFine steps on ONE volume per rank: the checks are the data-parallel invariants (reduced gradient == mean of the ranks' local gradients,
every element reduced exactly once, identical codebooks on every rank equal to a one-process emulation on the gathered tokens)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port, device_kind):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    if device_kind == "cpu":
        from ct_clip_amd import backend
        from tests.ref_backend import RefBackend
        backend.use(RefBackend())
        return torch.device("cpu")
    dev = torch.device("cuda", 0)          # every rank on the one device of the GPU box, talking through gloo
    torch.cuda.set_device(dev)
    return dev


def _load():
    g = torch.load(os.path.join(ROOT, "tests", "golden", "tiny.pt"), weights_only=False)
    f = torch.load(os.path.join(ROOT, "tests", "golden", "finetune_tiny.pt"), weights_only=False)
    return g, f


def _gather(t):
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous())
    return out


def lipro_worker(rank, world, port, device_kind, out, comm):
    dev = _init(rank, world, port, device_kind)
    from ct_clip_amd import finetune as FT, functional as Fn
    from tests.helpers import TextBatch, build_model
    g, f = _load()
    L = f["lipro"]
    clip = build_model(g["config"], g["state_dict"], dev, torch.float32)
    head = FT.ImageLatentsClassifier(clip, g["config"]["dim_latent"], 18, dropout_prob=0.0).to(dev)
    with torch.no_grad():
        head.classifier.weight.copy_(L["W"]); head.classifier.bias.copy_(L["b"])
    tr = FT.LiProTrainer(head, lr=1e-3, wd=0.1, warmup_length=2, total_steps=10, pos_weight=L["pos_weight"].tolist(),
                         grad_comm_dtype=torch.bfloat16 if comm == "bf16" else torch.float32)
    assert tr.reducer is not None and tr.reducer.op == "mean" and tr.reducer.flat.numel() == 18 * g["config"]["dim_latent"] + 20
    per = g["video"].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    blank = TextBatch(g["input_ids"][:1].to(dev), g["attention_mask"][:1].to(dev))
    loss, logits = tr.forward_backward(blank, g["video"][sl].to(dev), L["labels"][sl])
    assert len(tr.reducer.log) == 1, "grad all-reduce only: the head is ONE collective"
    assert tr.reducer.vq_sync.calls == 1 and not tr.reducer.vq_sync.pending
    losses = _gather(loss.detach().float().cpu().reshape(1))
    logits_all = _gather(logits.detach().float().cpu())
    dW, db = head.classifier.weight.grad.detach().float().cpu().clone(), head.classifier.bias.grad.detach().float().cpu().clone()
    vq = {k: v.detach().float().cpu().clone() for k, v in clip.state_dict().items() if "vq._codebook" in k}
    # one optimisation step: clip + AdamW + cosine schedule on the averaged gradient -> identical heads on every rank
    tr.optim.step(tr.max_grad_norm)
    heads = _gather(tr.optim.flat_param.detach().float().cpu())
    assert all(torch.equal(heads[0], h) for h in heads[1:]), "ranks must hold the same head after the step"
    tr.close()
    assert Fn.VqFn.stat_sync is None
    if rank == 0:
        torch.save(dict(loss=torch.cat(losses).mean(), logits=torch.cat(logits_all), dW=dW, db=db, vq=vq), out)
    dist.barrier()
    dist.destroy_process_group()


def vocabfine_worker(rank, world, port, device_kind, out, bucket_bytes, fused=True):
    dev = _init(rank, world, port, device_kind)
    from ct_clip_amd import finetune as FT, functional as Fn
    from ct_clip_amd.trainer import hot_path_parameters
    from tests.helpers import TextBatch, build_model
    g, f = _load()
    V = f["vocabfine"]
    clip = build_model(g["config"], g["state_dict"], dev, torch.float32)
    tr = FT.VocabFineTrainer(clip, tokenize=None, lr=1e-5, wd=0.1, warmup_length=2, total_steps=10, pathologies=["a", "b", "c", "d"],
                             group_size=V["group"], grad_bucket_bytes=bucket_bytes)
    P = V["prompt_ids"].shape[0]
    pairs = [TextBatch(V["prompt_ids"][i].to(dev), V["prompt_mask"][i].to(dev)) for i in range(P)]
    volume = g["video"][rank % g["video"].shape[0]][None].to(dev)           # a different volume on every rank
    vq = clip.visual_transformer.vq
    embed0, cluster0 = vq._codebook.embed.clone(), vq._codebook.cluster_size.clone()

    # (1) the ranks' LOCAL gradients, without any reduction (and with the codebook put back afterwards)
    red, tr.reducer = tr.reducer, None
    sync, Fn.VqFn.stat_sync = Fn.VqFn.stat_sync, None
    prev_hook = Fn.set_grad_ready_hook(None)
    # the local pass must read the same codebook sequence as the data-parallel one: feed it the GLOBAL statistics through an immediate hook
    from ct_clip_amd import distributed as D
    Fn.VqFn.stat_sync = staticmethod(D.sync_vq_stats)
    tr.forward_backward(volume, pairs, fused=fused)
    local = tr.optim.flat_grad.detach().clone()
    embed_seq, cluster_seq = vq._codebook.embed.clone(), vq._codebook.cluster_size.clone()
    with torch.no_grad():
        vq._codebook.embed.copy_(embed0); vq._codebook.cluster_size.copy_(cluster0)
    tr.reducer, Fn.VqFn.stat_sync = red, sync
    Fn.set_grad_ready_hook(prev_hook)

    # (2) the data-parallel step: deferred statistics on the communication stream, buckets from inside backward, mean over the ranks
    tr.reducer.log.clear()
    launched_in_backward = []
    orig_finish = tr.reducer.finish
    tr.reducer.finish = lambda: (launched_in_backward.append(len(tr.reducer.log)), orig_finish())[1]
    losses, sims = tr.forward_backward(volume, pairs, fused=fused)
    # fused: ONE backward, buckets leave from inside it; literal loop: one backward per group into the same gradients -> everything at finish()
    assert (launched_in_backward[0] > 0) == (fused and bucket_bytes == 1), launched_in_backward
    cover = sorted(tr.reducer.log)
    assert cover[0][0] == 0 and cover[-1][1] == tr.optim.flat_grad.numel() and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), \
        "every element of the flat gradient buffer must be reduced exactly once"
    assert tr.reducer.vq_sync.calls == P and not tr.reducer.vq_sync.pending            # one fused statistics all-reduce per quantiser call
    reduced = tr.optim.flat_grad.detach().float().cpu()
    locs = _gather(local.float().cpu())
    mean_local = sum(locs) / world
    scale = float(mean_local.abs().max())
    torch.testing.assert_close(reduced, mean_local, rtol=1e-4, atol=1e-6 * scale)
    assert float((locs[0] - locs[-1]).abs().max()) > 1e-3 * scale, "the ranks stepped on different volumes"
    # codebooks: identical on every rank, equal to the immediate-hook sequence of pass (1)
    embeds = _gather(vq._codebook.embed.detach().float().cpu())
    assert all(torch.equal(embeds[0], e) for e in embeds[1:])
    torch.testing.assert_close(vq._codebook.embed, embed_seq, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(vq._codebook.cluster_size, cluster_seq, rtol=1e-5, atol=1e-6)
    assert not torch.equal(vq._codebook.cluster_size, cluster0)
    tr.optim.step(None)
    params = _gather(tr.optim.flat_param.detach().float().cpu())
    assert all(torch.equal(params[0], p) for p in params[1:]), "ranks must hold the same parameters after the step"
    tr.close()
    if rank == 0:
        torch.save(dict(launches=len(cover), n=int(reduced.numel()), names=len(hot_path_parameters(clip))), out)
    dist.barrier()
    dist.destroy_process_group()
