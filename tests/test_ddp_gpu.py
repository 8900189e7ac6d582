"""The N > 1 path on real devices (reference DDP: scripts/CTCLIPTrainer.py:138-140,183-193).

Parity oracle for W ranks = the single-process REAL reference on the concatenated global batch, i.e. the golden fixture `tiny`:
rank r gets sample r; the gathered-negatives loss, the SUMMED gradients and the post-step VQ buffers must equal the golden ones.
  * two ranks on ONE device through gloo (runs on the 1-GPU box): every CUDA branch of GradReducer -- per-segment events on the
    announcing stream and the weight-gradient stream, buckets on the communication stream -- with buckets of one block each;
  * two ranks on TWO devices through nccl (= RCCL over xGMI; skipped unless the box has >= 2 GPUs): f32 and bf16 buckets;
  * `python bench.py --gpus 2` with NO launcher around it (the driver's invocation): bench.py re-executes itself under
    torch.distributed.run and rank 0 prints exactly one JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend_name, single_device, name, out, comm, dtype_name):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    import torch.distributed as dist
    dev = torch.device("cuda", 0 if single_device else rank)
    torch.cuda.set_device(dev)
    if backend_name == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend_name, rank=rank, world_size=world)
    import ct_clip_amd
    from ct_clip_amd.trainer import hot_path_parameters
    from tests.helpers import TextBatch, build_model
    g = torch.load(os.path.join(ROOT, "tests", "golden", f"{name}.pt"), weights_only=False)
    per = g["video"].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    clip = build_model(g["config"], g["state_dict"], dev, dtype)
    clip.train()
    trainer = ct_clip_amd.CTClipTrainer(clip, num_train_steps=1, batch_size=per, tokenizer=object(), lr=1e-3, train_dataset=list(range(world)), evaluate=False,
                                        checkpoint=False, results_folder=os.path.join(os.path.dirname(out), f"r{rank}"), num_workers=0, device=dev,
                                        grad_comm_dtype=torch.float32 if comm == "f32" else torch.bfloat16, grad_bucket_bytes=1)
    text = TextBatch(g["input_ids"][sl].to(dev), g["attention_mask"][sl].to(dev))
    loss = trainer.forward_backward(g["video"][sl].to(dev), text)
    torch.cuda.synchronize()
    log = list(trainer.reducer.log)
    cover = sorted(log)
    assert cover[0][0] == 0 and cover[-1][1] == trainer.optim.flat_grad.numel() and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), \
        "every element of the flat gradient buffer must be reduced exactly once"
    if rank == 0:
        grads = {n: p.grad.detach().float().cpu() for n, p in hot_path_parameters(clip)}
        vq = {k: v.cpu() for k, v in clip.state_dict().items() if "vq._codebook" in k}
        torch.save(dict(loss=loss.detach().cpu(), grads=grads, vq=vq, launches=len(log), backend=dist.get_backend(),
                        vq_collectives=trainer.reducer.vq_sync.calls), out)
    dist.barrier()
    dist.destroy_process_group()


def _check(golden, res, name, comm):
    from tests.helpers import check_grad
    g = golden(name)
    torch.testing.assert_close(res["loss"].float(), g["loss"], rtol=1e-4, atol=1e-5)          # north_star bar: 1e-3
    assert res["launches"] >= 8, "buckets of one block each: several collectives per step"
    n = 0
    for k, rec in g["grads"].items():
        if rec["value"].numel() == 0 or k not in res["grads"]:
            continue
        if comm == "bf16":
            check_grad(rec, res["grads"][k], rtol=2e-2, atol_rel=1e-2, floor=1e-8 * float(g["grad_norm"]))
        else:
            check_grad(rec, res["grads"][k], rtol=5e-3, atol_rel=1e-3, floor=1e-8 * float(g["grad_norm"]))
        n += 1
    assert n > 40
    for k, v in g["vq_after"].items():
        torch.testing.assert_close(res["vq"][k], v, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("comm", ["f32", "bf16"])
def test_two_ranks_one_device_gloo_match_golden(golden, tmp_path, comm):
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), "gloo", True, "tiny", out, comm, "f32"), nprocs=2, join=True)
    _check(golden, torch.load(out, weights_only=False), "tiny", comm)


def test_four_ranks_one_device_gloo_match_golden(golden, tmp_path):
    """world_size 4 (round 5): one sample of tests/golden/tiny4.pt (the real reference at B = 4) per rank, all four on cuda:0 through gloo --
    rank slices of the gathered latents, 4-way gradient sums, the deferred VQ-statistics all-reduce on the communication stream."""
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(4, _free_port(), "gloo", True, "tiny4", out, "f32", "f32"), nprocs=4, join=True)
    _check(golden, torch.load(out, weights_only=False), "tiny4", "f32")


@pytest.mark.parametrize("comm", ["f32", "bf16"])
def test_one_rank_nccl_issues_every_collective_and_matches_golden(golden, tmp_path, monkeypatch, comm):
    """The RCCL branch on the 1-GPU box: a process group of ONE rank through backend "nccl" with CTCLIP_DIST_SINGLE_RANK=1, under which the
    data-parallel branches are taken although every collective is the identity -- RCCL's all_reduce kernels run on the communication stream
    (stream-ordered wait, record_stream, per-segment events, the staged bf16 wire format), the latent all-gather and its backward slice run
    on the compute stream, the quantiser's statistics go through the deferred fused all-reduce -- and the step must still be the single-process
    golden step.  (What it cannot show is xGMI traffic: that needs the driver's multi-GPU run.)"""
    monkeypatch.setenv("CTCLIP_DIST_SINGLE_RANK", "1")
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(1, _free_port(), "nccl", True, "tiny", out, comm, "f32"), nprocs=1, join=True)
    res = torch.load(out, weights_only=False)
    assert res["backend"] == "nccl" and res["vq_collectives"] >= 1
    _check(golden, res, "tiny", comm)


@pytest.mark.skipif(torch.cuda.device_count() < 4, reason="needs four GPUs (RCCL over xGMI)")
def test_four_ranks_four_devices_nccl_match_golden(golden, tmp_path):
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(4, _free_port(), "nccl", False, "tiny4", out, "f32", "f32"), nprocs=4, join=True)
    _check(golden, torch.load(out, weights_only=False), "tiny4", "f32")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("comm", ["f32", "bf16"])
def test_two_ranks_two_devices_nccl_match_golden(golden, tmp_path, comm):
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), "nccl", False, "tiny", out, comm, "f32"), nprocs=2, join=True)
    _check(golden, torch.load(out, weights_only=False), "tiny", comm)


def _bench_line(args, env, cwd):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env, cwd=cwd)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"the contract: ONE JSON line and nothing else on stdout, got {lines[:6]}"
    return json.loads(lines[0])


def test_bench_self_launches_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` as the driver invokes it; on a 1-GPU box the two ranks share cuda:0 and talk through gloo."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env.update(CTCLIP_BENCH_BACKEND="gloo", CTCLIP_BENCH_SINGLE_DEVICE="1")
    rec = _bench_line(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--spatial-depth", "1", "--temporal-depth", "1",
                       "--no-attn-block", "--profile-steps", "0"], env, str(tmp_path))
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert abs(rec["loss"] - 1.386) < 0.2          # ln 4 for random towers: the loss saw the gathered batch
    c = rec["comm"]                                # per-step communication diagnostics of the N > 1 line
    assert c["collectives_per_step"] >= 1 and c["MB_per_step"] > 100 and c["comm_stream_busy_ms_per_step"] > 0 and c["exposed_wait_ms_per_step"] >= 0
    # one invocation decides the reducer's defaults: f32 / bf16 buckets x two bucket thresholds, re-timed back to back
    sw = rec["comm_sweep"]
    assert [(r["wire_dtype"], r["min_bucket_MB"]) for r in sw] == [("float32", 16.0), ("bfloat16", 16.0), ("float32", 64.0), ("bfloat16", 64.0)]
    assert all(r["ms_per_step"] > 0 and r["collectives_per_step"] >= 1 for r in sw)
    assert sw[1]["MB_per_step"] < 0.6 * sw[0]["MB_per_step"]


@pytest.mark.parametrize("workload", ["lipro", "vocabfine"])
def test_bench_finetune_workloads_synchronise_gradients_across_ranks(tmp_path, workload):
    """`python bench.py --workload lipro|vocabfine --gpus 2` (BASELINE.json configs[4] / configs[3] are multi-GPU configurations): the ranks are
    data-parallel -- a `comm` object with the gradient all-reduce (mean) -- not independent replicas."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env.update(CTCLIP_BENCH_BACKEND="gloo", CTCLIP_BENCH_SINGLE_DEVICE="1")
    rec = _bench_line(["--gpus", "2", "--workload", workload, "--image", "120", "--frames", "60", "--spatial-depth", "1", "--temporal-depth", "1", "--batch", "2",
                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--profile-steps", "0", "--text-len", "32"], env, str(tmp_path))
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and "all-reduced (mean)" in rec["config"]["note"]
    c = rec["comm"]
    assert c["op"] == "mean" and c["collectives_per_step"] >= 1
    if workload == "lipro":
        assert c["collectives_per_step"] == 1 and c["MB_per_step"] < 0.1          # "grad all-reduce only": 9 234 head parameters
    else:
        assert c["MB_per_step"] > 100


def test_bench_single_rank_rccl_line_is_alone_on_stdout(tmp_path):
    """bench.py with a ONE-rank "nccl" process group (CTCLIP_DIST_SINGLE_RANK=1): RCCL comes up on the 1-GPU box -- it prints a version banner
    on C stdout, which must not reach the process's stdout next to the contract line -- and the `comm` diagnostics are filled from the
    communication stream's events."""
    env = dict(os.environ, CTCLIP_DIST_SINGLE_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    rec = _bench_line(["--steps", "2", "--warmup", "1", "--batch", "2", "--spatial-depth", "1", "--temporal-depth", "1", "--no-attn-block",
                       "--profile-steps", "0", "--no-cpu-baseline", "--no-pmc", "--no-reference-depth", "--no-text512"], env, str(tmp_path))
    c = rec["comm"]
    assert rec["n_gpus"] == 1 and c["backend"] == "nccl" and c["collectives_per_step"] >= 1 and c["MB_per_step"] > 100
    assert abs(rec["loss"] - 0.693) < 0.2          # ln 2 for random towers at batch 2


@pytest.mark.parametrize("workload,cfgi", [("lipro", 4), ("vocabfine", 3)])
def test_bench_finetune_workloads_emit_the_contract_line(tmp_path, workload, cfgi):
    """bench.py --workload lipro / vocabfine (BASELINE.json configs[4] / configs[3]) at a reduced geometry: one JSON line with the contract's
    keys, a roofline object and a finite loss."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    rec = _bench_line(["--workload", workload, "--image", "120", "--frames", "60", "--spatial-depth", "1", "--temporal-depth", "1", "--batch", "2",
                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--profile-steps", "1", "--text-len", "32"], env, str(tmp_path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert f"configs[{cfgi}]" in rec["metric"] and rec["unit"] == "volumes/s" and rec["value"] > 0 and rec["n_gpus"] == 1
    assert "workload" in rec["config"] and rec["roofline"]["frac"] > 0 and rec["loss"] == rec["loss"]
