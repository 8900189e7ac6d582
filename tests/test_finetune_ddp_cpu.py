"""Data-parallel fine-tuning (BASELINE.json configs[3] VocabFine "8 GPUs", configs[4] LiPro "grad all-reduce only") on CPU through gloo:
the trainers of ct_clip_amd/finetune.py as one process per rank (the reference: nn.DataParallel, ct_lipro_train.py:75, ct_vocabfine_train.py:62).
Workers and the statement of the oracle: tests/finetune_ddp_workers.py."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests import finetune_ddp_workers as W


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_lipro(golden, res, tol, gtol):
    L = golden("finetune_tiny")["lipro"]
    torch.testing.assert_close(res["logits"], L["logits"], rtol=tol, atol=tol)
    torch.testing.assert_close(res["loss"], L["loss"], rtol=tol, atol=tol)            # mean of the rank losses = the global-batch loss
    torch.testing.assert_close(res["dW"], L["dW"], rtol=gtol, atol=gtol * float(L["dW"].abs().max()))
    torch.testing.assert_close(res["db"], L["db"], rtol=gtol, atol=gtol * float(L["db"].abs().max()))
    for k, v in L["vq_after"].items():          # the codebook moved by the GLOBAL batch's statistics, as in the one-process reference run
        torch.testing.assert_close(res["vq"][k], v, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("comm", ["f32", "bf16"])
def test_lipro_two_ranks_match_the_reference_global_batch(golden, tmp_path, comm):
    out = str(tmp_path / "lipro.pt")
    mp.spawn(W.lipro_worker, args=(2, _free_port(), "cpu", out, comm), nprocs=2, join=True)
    check_lipro(golden, torch.load(out, weights_only=False), 2e-4, 2e-3 if comm == "f32" else 1e-2)


@pytest.mark.parametrize("bucket_bytes,fused", [(1, True), (16 << 20, True), (1, False)])
def test_vocabfine_two_ranks_average_every_gradient(tmp_path, bucket_bytes, fused):
    out = str(tmp_path / "vocab.pt")
    mp.spawn(W.vocabfine_worker, args=(2, _free_port(), "cpu", out, bucket_bytes, fused), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    assert res["n"] > 100000 and res["names"] > 40
    assert res["launches"] >= (8 if (bucket_bytes == 1 and fused) else 1)


def test_data_parallel_loader_shards_disjointly(monkeypatch):
    """Without a process group: the reference's plain loader.  With one (patched world): rank shards are disjoint and cover the epoch."""
    from ct_clip_amd import distributed as D, finetune as FT
    ds = list(range(16))
    dl, sampler = FT.data_parallel_loader(ds, batch_size=4, shuffle=False)
    assert sampler is None and [b.tolist() for b in dl] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    seen = []
    for r in range(4):
        monkeypatch.setattr(D, "world_size", lambda: 4)
        monkeypatch.setattr(D, "rank", lambda r=r: r)
        dl, sampler = FT.data_parallel_loader(ds, batch_size=2, shuffle=True, seed=3)
        sampler.set_epoch(1)
        got = [int(x) for b in dl for x in b]
        assert len(got) == 4
        seen += got
    assert sorted(seen) == ds
