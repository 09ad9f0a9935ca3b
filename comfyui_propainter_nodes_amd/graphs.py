"""hipGraph capture of the launch-bound recurrences.

The two learned recurrences (flow completion: 2 x (T-1) steps of ~8 kernels on a 2 x 45 x 80 pixel problem; feature
propagation: 2 x l_t steps on the windows of the clip) are chains of small dependent kernels.  Each launch through the
C ABI costs ~15-20 us of host time (ctypes struct fill + hipLaunchKernel), which is as long as the kernels themselves
once they are latency-optimised, so the sweep is captured ONCE per problem shape into a hipGraph
(`torch.cuda.CUDAGraph` is a hipGraph on ROCm; our kernels are ordinary launches on torch's current stream, which is the
capture stream inside `torch.cuda.graph`) and replayed afterwards: one host call per sweep.

A captured sweep works on fixed device addresses: inputs are copied into static buffers before the replay (a few hundred
MB per clip at HBM speed: < 0.5 ms), outputs are the static buffers of the capture and stay valid until the next replay of
the same graph.  Capture is skipped (plain eager launches, same kernels, same results) on CPU tensors (the emulator of the
tests), while bench.py times individual launches with HIP events, and with PP_GRAPHS=0.
"""
from __future__ import annotations

import os
import threading
from collections import OrderedDict
from typing import Callable

import torch

from . import ops

_MAX_GRAPHS = int(os.environ.get("PP_GRAPH_CACHE", "24"))
_CAPTURE_LOCK = threading.Lock()


def enabled(*tensors: torch.Tensor) -> bool:
    if os.environ.get("PP_GRAPHS", "1") == "0" or ops.CONV_PROFILE is not None:
        return False
    return all(t.is_cuda for t in tensors) and not torch.cuda.is_current_stream_capturing()


class _Entry:
    __slots__ = ("graph", "static_in", "out")

    def __init__(self, graph, static_in, out):
        self.graph, self.static_in, self.out = graph, static_in, out


class GraphCache:
    """key -> captured sweep.  `fn(*inputs)` must be a pure function of its tensor arguments that only launches
    kernels / torch device ops on the current stream (no host synchronisation, no pageable H2D copies)."""

    def __init__(self, max_entries: int | None = None) -> None:
        """`max_entries`: LRU capacity (default PP_GRAPH_CACHE); a captured sweep owns the memory pool of everything it allocates
        -- RAFT's update block: the all-pairs volume, ~12 GB per clip shape at 640x360 -- so big sweeps get a small cache."""
        self._entries: "OrderedDict[tuple, _Entry]" = OrderedDict()
        self._max = max_entries or _MAX_GRAPHS

    def clear(self) -> None:
        """Drop the captured sweeps and their private memory pools (several GB per clip shape; torch's empty_cache does
        not release them while the graphs are alive).  The caches die with the models: pipeline.drop_model_cache()."""
        self._entries.clear()

    def run(self, key: tuple, fn: Callable, *inputs: torch.Tensor):
        if not enabled(*inputs):
            return fn(*inputs)
        key = key + tuple((tuple(t.shape), t.dtype, str(t.device)) for t in inputs)
        # capture, warm-up and replay on the device of the inputs (not whatever device happens to be current)
        with torch.cuda.device(inputs[0].device):
            return self._run_on_device(key, fn, inputs)

    def run_filled(self, key: tuple, fn: Callable, fill: Callable, device: torch.device):
        """Like run(), for a sweep whose inputs are GATHERED from larger tensors (r06): `fill(None)` allocates and returns the
        list of input tensors, `fill(bufs)` refills a previous list in place.  The capture's static inputs are then written by
        the gather itself -- run() copies every input into its static twin first (feature propagation: the whole clip's encoder
        features, 295 MB at cfg 2, per window group, before gathering from that copy inside the graph)."""
        if (os.environ.get("PP_GRAPHS", "1") == "0" or ops.CONV_PROFILE is not None or device.type != "cuda"
                or torch.cuda.is_current_stream_capturing()):
            return fn(*fill(None))
        with torch.cuda.device(device):
            return self._run_on_device(key + (str(device),), fn, None, fill)

    def _run_on_device(self, key: tuple, fn: Callable, inputs, fill: Callable | None = None):
        e = self._entries.get(key)
        if e is None:
            if fill is not None:
                static_in = inputs = fill(None)
            else:
                static_in = [torch.empty_like(t, memory_format=torch.contiguous_format).copy_(t) for t in inputs]
            # warm-up on a side stream (torch's capture recipe): one-time initialisation (LDS attributes, cached index
            # tensors, allocator growth) must not happen inside the capture
            s = torch.cuda.Stream(inputs[0].device)
            s.wait_stream(torch.cuda.current_stream())
            # one capture at a time per process: torch registers the device's RNG state with a capture and refuses a second
            # concurrent one ("Cannot register the state during capturing stage") -- the rank threads of
            # distributed.run_multi_device capture their sweeps at about the same moment
            _CAPTURE_LOCK.acquire()
            try:
                # (the counter is only touched under the capture lock -- a lost update between rank threads could
                #  have un-pinned index tensors whose addresses a graph had baked in)
                ops.PIN_DEVICE_INTS += 1      # index tensors requested from here on are baked into the graph: never evicted
                with torch.cuda.stream(s):
                    fn(*static_in)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                # capture stream on THIS device (torch's default capture stream lives on whichever device captured first);
                # thread-local error mode: other threads (the other ranks of run_multi_device, ComfyUI's own) may allocate meanwhile
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    out = fn(*static_in)
            finally:
                ops.PIN_DEVICE_INTS = max(0, ops.PIN_DEVICE_INTS - 1)
                _CAPTURE_LOCK.release()
            e = _Entry(g, static_in, out)
            self._entries[key] = e
            while len(self._entries) > self._max:
                self._entries.popitem(last=False)
            # r06: the warm-up run and the capture's own bookkeeping may have WRITTEN the static inputs (the transformer updates its
            # token tensor in place: the first replay ran on the warm-up's output -- 16 dB on the first clip of a process, caught
            # by tests/test_baseline_configs.py); sweeps that only read their inputs never noticed.  Refill before the first replay.
            if fill is not None:
                fill(e.static_in)
            else:
                for dst, src in zip(e.static_in, inputs):
                    dst.copy_(src)
        else:
            self._entries.move_to_end(key)
            if fill is not None:
                fill(e.static_in)
            else:
                for dst, src in zip(e.static_in, inputs):
                    dst.copy_(src)
        e.graph.replay()
        return e.out
