"""Side streams that really run beside the main stream.

HIP multiplexes a process's streams onto a handful of hardware queues (4 by default); two streams that land on the same queue are served in
order -- a kernel, or an event wait, of one holds back the other.  Which queue a new `torch.cuda.Stream` gets depends on everything the process
created before it: measured on the MI355X box (`profiles/r05_stream_queues.md`), the text tower's stream sat on its own queue in a
single-process run and on the MAIN stream's queue as soon as a process group existed (RCCL's streams and the communication stream came
first): the text tower then ran serialised behind the image tower and the step was 5.5 ms longer, with not one kernel added.

There is no API that names a stream's queue, so the streams handed out here are PROBED: a candidate is accepted when a small kernel launched
on it completes while a long-running single-thread spin kernel (`ctclip_spin`, csrc/misc.hip: the library's own -- up to round 5 this leaned on
the private `torch.cuda._sleep`) occupies the default stream and every side stream handed out before.  A few milliseconds once per purpose and
device.  `CTCLIP_STREAM_PROBE=0` returns the first stream torch offers (the behaviour up to round 4).

When NO candidate runs beside the default stream the overlap the caller wanted is lost (the step is still correct): that is reported ONCE per
purpose on stderr and as `concurrent_with_default: False` in `report()` (bench.py prints it in the `side_streams` object of its line).

Order matters when the queues run out: ask for the streams that carry kernels first (text tower, weight gradients), for the communication
stream last."""
import os
import sys

import torch

_TAKEN = {}        # device index -> [default stream, side streams handed out so far]
_BY_PURPOSE = {}   # (device index, purpose) -> stream
_REPORT = {}       # (device index, purpose) -> dict(tries=..., concurrent_with_default=..., concurrent_with_all=...)
_SCRATCH = {}
MAX_TRIES = 24     # torch's pool has 32 streams per device and priority


def _dev_index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def _scratch(idx):
    t = _SCRATCH.get(idx)
    if t is None:
        t = _SCRATCH[idx] = torch.zeros(64, dtype=torch.float32, device=torch.device("cuda", idx))
    return t


SPIN_US = 2000     # how long the probe's spin kernels hold their queues


def _spin(stream):
    """The library's spin kernel (ctclip_spin) on `stream`."""
    from . import _lib
    _lib.check(_lib.load().ctclip_spin(SPIN_US, stream.cuda_stream), "ctclip_spin")


def runs_beside(cand, busy):
    """True when a kernel on `cand` completes while every stream of `busy` (busy[0] = the reference clock) is held by a spin kernel."""
    idx = cand.device.index
    dev = torch.device("cuda", idx)
    x = _scratch(idx)
    with torch.cuda.stream(cand):
        x.add_(1.0)      # the first launch on a new stream may create its hardware queue (milliseconds): not part of the measurement
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(busy[0])
    ends = []
    for s in busy:
        _spin(s)
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        ends.append(e)
    with torch.cuda.stream(cand):
        x.add_(1.0)
        ec = torch.cuda.Event(enable_timing=True)
        ec.record(cand)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(ec) < 0.5 * min(e0.elapsed_time(e) for e in ends)


_WARNED = set()


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        print(f"ct_clip_amd.streams: {msg}", file=sys.stderr)


def concurrent_stream(device, purpose):
    """The process-wide side stream for (device, purpose): created once, chosen so that it shares a hardware queue neither with the default
    stream nor -- while queues last -- with the side streams handed out before."""
    idx = _dev_index(device)
    key = (idx, purpose)
    st = _BY_PURPOSE.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if st is not None:
        # a purpose first asked for INSIDE a graph capture got an unprobed stream: probe it the first time it is asked for outside one
        if _REPORT.get(key, {}).get("probed") is False and _REPORT[key].get("why") == "capture" and not capturing:
            forget(device, purpose)
        else:
            return st
    dev = torch.device("cuda", idx)
    if os.environ.get("CTCLIP_STREAM_PROBE", "1") == "0" or capturing:
        st = torch.cuda.Stream(device=dev)      # (inside a graph capture nothing can be launched to probe with)
        _REPORT[key] = dict(tries=0, probed=False, why="capture" if capturing else "CTCLIP_STREAM_PROBE=0")
    else:
        taken = _TAKEN.setdefault(idx, [torch.cuda.default_stream(dev)])
        tried, fallback, st = [], None, None
        for _ in range(MAX_TRIES):
            cand = torch.cuda.Stream(device=dev)
            if any(cand == t for t in tried):
                break                            # the pool wrapped around
            tried.append(cand)
            if any(cand == t for t in taken):
                continue                         # torch's pool is process-wide and wraps: this one already serves another purpose
            if runs_beside(cand, taken):
                st = cand
                break
            if fallback is None and len(taken) > 1 and runs_beside(cand, taken[:1]):
                fallback = cand                  # at least not on the main stream's queue
        free = [t for t in tried if not any(t == u for u in taken)]
        _REPORT[key] = dict(tries=len(tried), probed=True, concurrent_with_all=st is not None,
                            concurrent_with_default=st is not None or fallback is not None)
        if st is None:
            st = fallback if fallback is not None else (free[0] if free else torch.cuda.Stream(device=dev))
            if fallback is None:
                _warn_once(key, f"no stream for '{purpose}' on cuda:{idx} runs beside the default stream (all {len(tried)} candidates share its "
                                "hardware queue): work on it will be SERIALISED with the main stream -- correct, but the overlap is lost "
                                "(see GPU_MAX_HW_QUEUES / profiles/r05_stream_queues.md)")
            else:
                _warn_once(key, f"the stream for '{purpose}' on cuda:{idx} runs beside the default stream but shares a hardware queue with "
                                "another side stream")
        taken.append(st)
    _BY_PURPOSE[key] = st
    return st


def forget(device, purpose):
    """Hand a purpose's stream back (tests): it no longer counts as occupied when later streams are probed."""
    idx = _dev_index(device)
    st = _BY_PURPOSE.pop((idx, purpose), None)
    _REPORT.pop((idx, purpose), None)
    if st is not None and idx in _TAKEN:
        _TAKEN[idx] = [t for t in _TAKEN[idx] if t is not st]


def report():
    """{purpose@device: how the stream was found} -- bench.py prints it with the multi-rank line."""
    return {f"{p}@cuda:{i}": dict(v) for (i, p), v in _REPORT.items()}
