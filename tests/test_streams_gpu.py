"""ct_clip_amd/streams.py on the device: the side streams handed out run BESIDE the default stream (they do not share its hardware queue), and
the probe can tell the difference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_side_streams_run_beside_the_default_stream():
    from ct_clip_amd import streams
    dev = torch.device("cuda", 0)
    main = torch.cuda.default_stream(dev)
    a = streams.concurrent_stream(dev, "test_a")
    b = streams.concurrent_stream(dev, "test_b")
    assert a != main and b != main and a != b
    assert streams.concurrent_stream(dev, "test_a") is a, "one stream per purpose, device and process"
    assert streams.runs_beside(a, [main]) and streams.runs_beside(b, [main])
    assert not streams.runs_beside(a, [a]), "the probe must see a kernel queued behind the spin kernel of its own stream"
    rep = streams.report()
    assert rep["test_a@cuda:0"]["probed"] and rep["test_a@cuda:0"]["concurrent_with_default"]
    x = torch.ones(1 << 20, device=dev)
    with torch.cuda.stream(a):
        y = x * 2
    torch.cuda.current_stream(dev).wait_stream(a)
    assert float(y.sum()) == 2.0 * (1 << 20)
    streams.forget(dev, "test_a")
    streams.forget(dev, "test_b")
    assert "test_a@cuda:0" not in streams.report() and streams.concurrent_stream(dev, "test_a") is not None
    streams.forget(dev, "test_a")


def test_the_models_streams_are_probed():
    """The text tower's, the weight-gradient and the communication stream come from streams.concurrent_stream, kernel-carrying streams first."""
    from ct_clip_amd import functional as Fn, streams
    dev = torch.device("cuda", 0)
    Fn.reserve_side_streams(dev)
    text, wg = Fn.shared_side_stream(dev, "text"), Fn._wgrad_stream(dev)
    comm = streams.concurrent_stream(dev, "comm")
    main = torch.cuda.default_stream(dev)
    assert len({text.stream_id, wg.stream_id, comm.stream_id, main.stream_id}) == 4
    for s in (text, wg, comm):
        assert streams.runs_beside(s, [main])
    rep = streams.report()
    if rep["text@cuda:0"]["concurrent_with_all"] and rep["wgrad@cuda:0"]["concurrent_with_all"]:      # (queues were left when they were created)
        assert streams.runs_beside(wg, [main, text]), "image-tower weight gradients and the text tower on different queues"
