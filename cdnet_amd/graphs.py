"""HIP-graph capture of the launch-bound inner loops (no tracing compiler: the Python step is simply run once under stream capture).

`GraphedTrainStep` : forward + 5-term loss + backward of `Trainer.train_step` as ONE graph launch (~400 kernel launches for the
                     DAM-Unet, > 1000 for HRNet18); the gradient all-reduce, the fused Adam step and the weight re-packs stay eager
                     (host-computed bias correction, RCCL), so the update is bit-identical to the eager step.
`GraphedCallable`  : any inference function of static-shape CUDA tensors (pipeline.infer_tiles) -> static output tensors.

PyTorch is used for what it is used for everywhere else here - memory and streams: torch.cuda.CUDAGraph owns the capture stream
and a private memory pool, every kernel inside is a libcdnet_hip.so launch on that stream."""
import torch

from . import runtime


def _walk(o):
    if torch.is_tensor(o):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _walk(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _walk(v)


class GraphedCallable:
    def __init__(self, fn, *static_inputs, warmup=2):
        self.fn, self.inputs = fn, [t for t in static_inputs]
        for _ in range(warmup):                      # kernel attributes, weight packs, BatchNorm folds, workspaces
            fn(*self.inputs)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn(*self.inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            if src is not dst:
                dst.copy_(src)
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """trainer.train_step with forward + loss + backward replayed from a graph.  The batch tensors given at construction are the
    static inputs; a call with other tensors copies them in first."""

    def __init__(self, trainer, batch, warmup=3):
        assert trainer.world == 1, 'graph capture covers the single-GPU step (the overlapped all-reduce is launched from Python)'
        self.tr, self.batch = trainer, [t for t in batch]
        trainer._packb_stream = False                 # no cross-stream event from eager code into the captured region (this trainer only)
        for _ in range(warmup):
            trainer.train_step(*self.batch)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._fwd_bwd()
        trainer._forwards -= 1                        # the capture ran no kernel
        # the captured pass left gradients of nothing: run it for real so that the state equals `warmup + 1` eager steps
        self.graph.replay()
        trainer._forwards += 1
        trainer.allreduce_and_step()

    def _fwd_bwd(self):
        tr = self.tr
        out = tr.forward(self.batch[0])
        if len(self.batch) == 5:
            g = tr.loss_and_grads(out[0], out[1], out[2], *self.batch[1:])
            tr.backward(*g)
        else:                                         # UNetTrainer
            tr.backward(tr.loss_and_grads(out, *self.batch[1:]))

    def __call__(self, *batch):
        for dst, src in zip(self.batch, batch):
            if src is not dst:
                dst.copy_(src)
        self.graph.replay()
        self.tr._forwards += 1
        self.tr.allreduce_and_step()
        return getattr(self.tr, 'unet_losses', None) if len(self.batch) != 5 else self.tr.losses
