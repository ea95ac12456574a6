"""Weight-gradient launches of the encoder / decoders on a SIDE stream (round 5).

In the backward of an MBConv / decoder block the chain that the next block waits for is the DATA gradient; the weight gradients
(1x1 expand / project, depthwise, decoder 3x3: ~100 launches of 10-35 us per training step, reference graph: mymodels.py:242-258
through torch autograd) are leaves of the graph -- nobody reads them before the optimizer (or the gradient all-reduce).  At planes
of 2x2 .. 32x32 every one of these kernels occupies a few dozen CUs, so two such chains on two HIP streams do run side by side
(tools/graph_branch_probe.py: two 200-kernel chains 1.28 ms on one stream, 0.77 ms on two; a chip-filling kernel next to a small
one does NOT overlap, which is why the head's weight gradients stay where they are).

    with sidework.side(x, dy, dw, ws):          # tensors the enclosed launches touch (allocated on the current stream)
        launch(...)                              # _lib.stream_ptr() is the side stream in here

`side` orders the side stream behind everything queued on the current stream so far, marks the tensors as in use by it
(record_stream: the caching allocator will not recycle them early) and queues ONE join per backward pass -- the current stream
waits for the side stream -- through the autograd engine's end-of-backward callback, so `.grad` is safe to read on the usual stream
as soon as `backward()` returns (optimizer.step(), a drop-in train.py).  `join()` does the same at once (GradReducer calls it
before a bucket's all-reduce).  Inside a stream capture, off a CUDA device or with SRBH_WGRAD_SIDE=0 the launches run in place.
"""
from __future__ import annotations

import os

import torch

# OFF by default: the same-box A/B of the training step (profiles/r05b_ab_wgrad_side.txt) measured 33.26 ms without and 33.41 / 33.57 ms
# with the side stream -- the ~100 cross-stream event pairs cost the main queue more than the overlap returns on this driver (the step
# equals its kernel chain either way: eager == one captured graph, profiles/r05c_ab_train_graph.txt).  Kept as a tested switch.
ENABLED = os.environ.get("SRBH_WGRAD_SIDE", "0") == "1"
_STATE = {}          # device index -> {"stream": Stream, "dirty": bool}


def _state(dev_index):
    st = _STATE.get(dev_index)
    if st is None:
        st = _STATE[dev_index] = {"stream": torch.cuda.Stream(device=dev_index), "dirty": False}
    return st


def join(device=None):
    """the current stream waits for every side launch issued so far (no host synchronisation)"""
    for idx, st in _STATE.items():
        if st["dirty"] and (device is None or torch.device(device).index in (None, idx)):
            torch.cuda.current_stream(idx).wait_stream(st["stream"])
            st["dirty"] = False


def _end_of_backward():
    join()


class side:
    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None

    def __enter__(self):
        if not ENABLED or not self.tensors or not self.tensors[0].is_cuda or torch.cuda.is_current_stream_capturing():
            return self
        idx = self.tensors[0].device.index
        st = _state(idx)
        cur = torch.cuda.current_stream(idx)
        st["stream"].wait_stream(cur)
        for t in self.tensors:
            t.record_stream(st["stream"])
        # a join at the end of THIS backward pass (queued per use: a callback is one list append, a join with nothing pending a no-op; a flag
        # remembered across passes would be wrong after a backward that raised).  queue_callback is only legal while a backward pass is
        # running -- the only place this context is used from; anywhere else the launches run in place
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        except RuntimeError:
            return self
        st["dirty"] = True
        self.ctx = torch.cuda.stream(st["stream"])
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False
