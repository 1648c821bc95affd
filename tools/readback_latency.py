#!/usr/bin/env python3
"""Host-visible latency of reading a small result back after a short kernel: pageable .to('cpu'), pinned copy + stream sync,
pinned copy + event sync, and (kernel writes -> copy -> host polls the pinned word)."""
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(16, dtype=torch.float64, device=dev)
pin = torch.zeros(16, dtype=torch.float64).pin_memory()
K = 2000


def bench(name, fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    dt = (time.perf_counter() - t0) / K * 1e6
    print("%-44s %7.1f us per (tiny kernel + readback)" % (name, dt))


def pageable():
    x.add_(1.0)
    return x.to("cpu")


def pinned_stream_sync():
    x.add_(1.0)
    pin.copy_(x, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return pin


ev = torch.cuda.Event()


def pinned_event_sync():
    x.add_(1.0)
    pin.copy_(x, non_blocking=True)
    ev.record()
    ev.synchronize()
    return pin


def pinned_event_query():
    x.add_(1.0)
    pin.copy_(x, non_blocking=True)
    ev.record()
    while not ev.query():
        pass
    return pin


def pinned_poll_value():
    x.add_(1.0)
    want = float(pin[0]) + 1.0
    pin.copy_(x, non_blocking=True)
    while float(pin[0]) != want:
        pass
    return pin


def kernel_only():
    x.add_(1.0)


bench("kernel launch only (async)", kernel_only)
torch.cuda.synchronize()
bench("pageable .to('cpu')", pageable)
bench("pinned copy_ + stream.synchronize()", pinned_stream_sync)
bench("pinned copy_ + event.synchronize()", pinned_event_sync)
bench("pinned copy_ + spin on event.query()", pinned_event_query)
pin.copy_(x); torch.cuda.synchronize()
bench("pinned copy_ + spin on the value", pinned_poll_value)
