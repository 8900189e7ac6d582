"""Data-parallel fine-tuning on the HIP path (BASELINE.json configs[3] / configs[4]): two ranks on the ONE device of the GPU box, talking through
gloo -- every CUDA branch of the reducer (communication stream, per-segment events, the deferred statistics all-reduce) under the fine-tuning
trainers.  Same workers and oracle as tests/test_finetune_ddp_cpu.py (tests/finetune_ddp_workers.py): LiPro against the REAL reference's
global-batch step (tests/golden/finetune_tiny.pt), VocabFine against the data-parallel invariants."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests import finetune_ddp_workers as W
from tests.test_finetune_ddp_cpu import check_lipro

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_lipro_two_ranks_one_device_match_the_reference_global_batch(golden, tmp_path):
    out = str(tmp_path / "lipro.pt")
    mp.spawn(W.lipro_worker, args=(2, _free_port(), "cuda", out, "f32"), nprocs=2, join=True)
    check_lipro(golden, torch.load(out, weights_only=False), 5e-4, 5e-3)


@pytest.mark.parametrize("bucket_bytes", [1, 16 << 20])
def test_vocabfine_two_ranks_one_device_average_every_gradient(tmp_path, bucket_bytes):
    out = str(tmp_path / "vocab.pt")
    mp.spawn(W.vocabfine_worker, args=(2, _free_port(), "cuda", out, bucket_bytes, True), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    assert res["n"] > 100000 and res["launches"] >= (8 if bucket_bytes == 1 else 1)
