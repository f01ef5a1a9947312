"""A second HIP stream that really runs beside the compute stream (the trainer's weight-gradient stream, the inference pipeline's
post-processing stream)."""
import os

import torch

_SIDE = {}


def side_stream(device=None):
    """One stream per device, chosen once.  HIP places streams on a few hardware queues in creation order; a side stream that shares the
    compute stream's queue serialises with it, and the cross-stream events then cost more than one stream would (training: 1 456 vs 1 622
    vs 1 780 tiles/s; inference with the post-processing stream: 9 490 vs 10 300 vs 10 700 tiles/s - the stream a process gets from
    torch.cuda.Stream() depends on how many it created before).  Candidates are timed with two spin kernels: together they take as long
    as one when the queues differ, twice as long when they do not.  CDNET_SIDE_STREAM_PROBE=0: the first candidate, unprobed."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    hit = _SIDE.get(key)
    if hit is not None:
        return hit
    cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
    spin = getattr(torch.cuda, '_sleep', None)
    pick = cands[0]
    if spin is not None and os.environ.get('CDNET_SIDE_STREAM_PROBE', '1') != '0':
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream()

            def timed(other):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                if other is not None:
                    other.wait_stream(main)                 # (starts with the compute stream's spin, not before the first event)
                    with torch.cuda.stream(other):
                        spin(400000)
                spin(400000)
                if other is not None:
                    main.wait_stream(other)
                e1.record(main)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1)
            timed(None)
            alone = min(timed(None), timed(None))
            for c in cands:
                if min(timed(c), timed(c)) < 1.5 * alone:
                    pick = c
                    break
    _SIDE[key] = pick
    return pick
