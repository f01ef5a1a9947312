"""Training step of the CDNet model on the HIP kernels: forward (batch-statistics BatchNorm), the five-term loss,
backward, gradient all-reduce over RCCL, fused Adam - the device-side replacement of the body of the reference's
train_util_dam.train (train_util_dam.py:54-311) with the host loops (:73-142, :278-289) moved onto the GPU.

No autograd: the forward records a tape of the convolution layers it ran; backward walks the tape in reverse.  For every
layer it (1) turns the consumers' gradients into the gradient of the raw convolution output (BatchNorm + residual + ReLU
+ max-pool/pad routing fused, cdnet_bn_backward), (2) computes dW on the matrix cores (cdnet_conv_backward_weight) and
(3) computes the input gradient as a forward convolution with a flipped/transposed weight pack (cdnet_conv_forward).
Parameters, gradients and the Adam moments live in flat fp32 buffers (one fused Adam launch, one all-reduce bucket
sequence); the nn.Parameters of the model are views into them.
"""
import ctypes as C
import os

import torch

from . import _lib, engine, runtime, streams
from .engine import Src

HEAD_PARAMS = ['point_conv.weight', 'direction_conv.weight', 'mask_conv.weight', 'point_conv.bias',
               'direction_conv.bias', 'mask_conv.bias', 'directionAtt.Conv1x1.weight', 'maskAtt.Conv1x1.weight']


class GradIn(C.Structure):
    _fields_ = [('g', C.c_void_p), ('Hg', C.c_int), ('Wg', C.c_int), ('oy', C.c_int), ('ox', C.c_int),
                ('pooled', C.c_int), ('coff', C.c_int), ('cstride', C.c_int), ('pad_', C.c_int)]


class BnBwdArgs(C.Structure):
    _fields_ = [('raw', C.c_void_p), ('res', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('mean', C.c_void_p), ('invstd', C.c_void_p), ('gin', GradIn * 3), ('ngin', C.c_int),
                ('f16', C.c_int), ('relu', C.c_int), ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('C', C.c_int)]


# workgroups of one weight-gradient launch while it runs beside the input-gradient chain (bf16 mode; see _weight_backward):
# layers of up to 128 x 128 pixels / larger ones
_WGRAD_WGS_DEEP, _WGRAD_WGS_SHALLOW = 128, 160
_WGRAD_WGS_KQ = 128                        # wgrad_ws_kernel<1, 1> on 512 x 512 layers (HRNet's branch 1)
_WGRAD_WGS_F32 = 160                      # (128 / 160 / 256: 989 / 994 / 987 tiles/s, three runs each on one box)
_RU_1X1_SIDE = True      # residual units' 1x1 backward-data beside the chain (tests flip it: same gradients either way)
WGRAD_STREAM = True       # weight gradients on a second stream beside the input-gradient chain
_WGRAD_DEFER = 0x100                       # CDNET_WGRAD_DEFER_REDUCE (include/cdnet_hip.h)
_WGRAD_DEEP_HW = 16384


class _on_stream:
    """`with torch.cuda.stream(s)` without its bookkeeping (device checks, Stream objects: ~15 us per use, once per layer of the
    backward pass): make `s` current, restore the previous stream on exit"""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = torch.cuda.current_stream()
        torch.cuda.set_stream(self.s)

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


class _G:
    """a gradient contribution for a stored tensor"""
    __slots__ = ('t', 'Hg', 'Wg', 'oy', 'ox', 'pooled', 'coff', 'cstride', 'event')

    def __init__(self, t, Hg, Wg, oy=0, ox=0, pooled=0, coff=0, cstride=0):
        self.t, self.Hg, self.Wg, self.oy, self.ox, self.pooled, self.coff, self.cstride = t, Hg, Wg, oy, ox, pooled, coff, cstride
        self.event = None                    # produced on another stream: the consumer's stream waits for this event first


class FlatState:
    """fp32 flat buffers: parameters, gradients, Adam moments.  Head parameters first (in the kernel's block layout),
    then every other parameter that takes part in forward, then the reference's never-used parameters (no gradient,
    never stepped - torch.optim.Adam skips parameters whose .grad is None)."""

    def __init__(self, model):
        # a model that computes on zero-padded parameter copies (HRNet) hands those over; its own parameters stay views
        named = model.trainer_named_parameters() if hasattr(model, 'trainer_named_parameters') else dict(model.named_parameters())
        unused = [n for n in named if n.startswith(tuple(getattr(model, 'UNUSED_PREFIXES', ())))]
        head = [n for n in HEAD_PARAMS if n in named and n not in unused]
        rest = [n for n in named if n not in head and n not in unused]
        self.order = head + rest + unused
        sizes = [named[n].numel() for n in self.order]
        total = sum(sizes)
        dev = next(model.parameters()).device
        self.P = torch.empty((total,), dtype=torch.float32, device=dev)
        self.G = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.M = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.V = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for n, sz in zip(self.order, sizes):
                p = named[n]
                self.P[off:off + sz].copy_(p.detach().reshape(-1))         # one-time host-side setup
                p.data = self.P[off:off + sz].view(p.shape)
                p.grad = self.G[off:off + sz].view(p.shape)
                self.offsets[n] = (off, sz)
                off += sz
        if hasattr(model, 'rebind_views'):
            model.rebind_views()
        self.n_used = sum(named[n].numel() for n in head + rest)
        self.n_head = sum(named[n].numel() for n in head)
        self.step_count = 0


def _choose_ci_tiles(C_src, Cout):
    if Cout <= 32 and runtime.PRECISION != 'fp32':
        return 1            # 16-bit path: 32-input-channel blocks; with at most 32 output channels the library runs ONE 32 x 32 block per
                            # workgroup and splits the tile's rows over the consumer waves (wgrad_ws_kernel<1, 1>)
    best, best_cost = 2, None
    for ci_t in (2, 1, 4):
        CI, CO = ci_t * 32, (4 // ci_t) * 32
        cost = -(-C_src // CI) * CI * -(-Cout // CO) * CO
        if best_cost is None or cost < best_cost:
            best, best_cost = ci_t, cost
    return best


class GradTerm(C.Structure):
    _fields_ = [('g', C.c_void_p), ('cstride', C.c_int), ('coff', C.c_int)]


class Trainer:
    G = _G

    def __init__(self, model, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.99), eps=1e-8, quirk_sample0=True,
                 world_size=1, bucket_mb=25):
        self.model = model
        self._pack_jobs = None
        self._ar = None
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.quirk = int(quirk_sample0)
        self.world = world_size
        self.bucket = int(float(bucket_mb) * (1 << 20) // 4)
        self.flat = FlatState(model)
        model._head_flat = self.flat.P[:self.flat.n_head] if self.flat.n_head == 855 else None
        self.dev = self.flat.P.device
        self._bufs = {}
        self._ws_bn = None
        self._ws_slab = None
        self._ws_head = None
        self._ws_loss = None
        self.losses = torch.zeros((11,), dtype=torch.float32, device=self.dev)     # 6 loss values + 5 pixel metrics
        self.tape = []
        self._cat_cache = {}
        self._wstream, self._events = None, {}
        self.ar_stats = None                     # buckets of the last step's all-reduce: total / released before backward ended
        self._packb_pending = False
        self._packb_stream = True              # backward-data re-packs beside the next forward
        self._side_active = False
        # split-K sums of the weight gradients: deferred and batched (one cdnet_wgrad_reduce_batch launch per ~CDNET_WGRAD_REDUCE_MB of
        # slabs instead of one reduce behind every weight-gradient launch; every call keeps its own slab buffer - 1.4 GB for the UNet)
        # Memory: with the default every weight-gradient call keeps its own split-K slab buffer for the trainer's lifetime (`buf(('wslab', ...))`,
        # ~1.4 GB for the DAM-Unet at 16 tiles, more for HRNet at 512²; a batch-size change re-allocates the buffers of the new shapes and frees
        # the old ones); CDNET_WGRAD_REDUCE_MB=0 returns to one shared slab and a reduce behind every launch.
        self._rd_mb = float(os.environ.get('CDNET_WGRAD_REDUCE_MB', '300'))        # 0: the reduce inside every call (one shared slab)
        self._rd_pending, self._rd_params, self._rd_bytes, self._rd_tables = [], [], 0, {}
        self._forwards, self._bn_base = 0, 0                 # training forwards run here / counted in a loaded checkpoint
        if world_size > 1:
            self.sync_from_rank0()                           # replicas start identical whatever each rank's RNG / checkpoint did

    # ------------------------------------------------------------------------------------------------
    def sync_from_rank0(self, src=0):
        """Broadcast rank `src`'s parameters (the whole flat buffer incl. the never-used ones), Adam moments, step counter and
        every module buffer (BatchNorm running statistics) to all ranks - what nn.DataParallel's per-iteration replicate
        (train.py:185) guarantees in the reference.  Called at construction and after a checkpoint load; without it a rank whose
        RNG drifted or that loaded a different file would silently train a different replica."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        f = self.flat
        works = [dist.broadcast(t, src=src, async_op=True) for t in (f.P, f.M, f.V)]
        meta = torch.tensor([float(f.step_count), float(self._bn_base + self._forwards)], dtype=torch.float64, device=self.dev)
        works.append(dist.broadcast(meta, src=src, async_op=True))
        bufs = [b for b in self.model.buffers() if b.is_floating_point()]
        if bufs:
            flatb = torch.cat([b.detach().reshape(-1).float() for b in bufs])
            dist.broadcast(flatb, src=src)
            off = 0
            with torch.no_grad():
                for b in bufs:
                    b.copy_(flatb[off:off + b.numel()].view(b.shape))
                    off += b.numel()
        for w in works:
            w.wait()
        f.step_count = int(meta[0].item())
        self._bn_base, self._forwards = int(meta[1].item()), 0
        self.refresh_parameters()

    def reduce_scalars(self, values):
        """mean over ranks of a small vector of logging scalars (the 11 values of train_util_dam.train): with nn.DataParallel the
        reference computes them on the gathered global batch; here every rank holds its shard's means"""
        import torch.distributed as dist
        if self.world <= 1 or not (dist.is_available() and dist.is_initialized()):
            return values
        t = torch.as_tensor(values, dtype=torch.float64).to(self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return (t / dist.get_world_size()).cpu().numpy()

    # ------------------------------------------------------------------------------------------------
    def buf(self, key, shape, dtype):
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.empty(shape, dtype=dtype, device=self.dev)
            self._bufs[key] = b
        return b

    def _bn_ws(self, Cc):
        need = _lib.load().cdnet_bn_backward_workspace_floats(Cc)
        if self._ws_bn is None or self._ws_bn.numel() < need:
            self._ws_bn = torch.empty((need,), dtype=torch.float32, device=self.dev)
        return self._ws_bn

    def grad_sum(self, gl, mask, npix, Cc, out):
        """out = [mask > 0] * sum of the gradient contributions `gl` (cdnet_grad_sum)"""
        assert 1 <= len(gl) <= 6
        arr = (GradTerm * len(gl))()
        for k, g in enumerate(gl):
            assert not g.pooled and g.oy == 0 and g.ox == 0
            arr[k].g, arr[k].cstride, arr[k].coff = g.t.data_ptr(), g.cstride or Cc, g.coff
        entry = 'cdnet_grad_sum_f32' if out.dtype == torch.float32 else 'cdnet_grad_sum'
        _lib.call(entry, C.byref(arr), len(gl), None if mask is None else _lib.ptr(mask), npix, Cc, _lib.ptr(out), _lib.stream_ptr())

    def take(self, grads, key):
        """pop the gradient contributions of a stored tensor for a consumer on the CURRENT stream: a contribution computed beside the chain
        (_RU_1X1_SIDE: the residual units' 1x1 backward-data on the weight-gradient stream) carries an event the consumer's stream must
        wait for first - wherever the list is consumed (a layer's BatchNorm backward, cat_grad, FuseNode.backward)"""
        gl = grads.pop(key, None)
        if gl is not None:
            for g_ in gl:
                if g_.event is not None:
                    torch.cuda.current_stream().wait_event(g_.event)
                    g_.event = None
        return gl

    def cat_grad(self, o, grads):
        """gradient of a concatenation buffer (several consumers, several writers): summed once per backward"""
        key = id(o)
        if key not in self._cat_cache:
            gl = self.take(grads, key)
            if gl is None:
                d = None
            elif len(gl) == 1 and not gl[0].coff and gl[0].cstride in (0, o.shape[3]):
                d = gl[0].t
            else:
                d = self.buf(('dcat', key), o.shape, runtime.act_dtype())
                self.grad_sum(gl, None, o.shape[0] * o.shape[1] * o.shape[2], o.shape[3], d)
            self._cat_cache[key] = d
        return self._cat_cache[key]

    def _slab(self, n):
        if self._ws_slab is None or self._ws_slab.numel() < n:
            if self._wstream is not None:
                self._wstream.synchronize()          # the old workspace may still be in use on the weight-gradient stream
            self._ws_slab = torch.empty((n,), dtype=torch.float32, device=self.dev)
        return self._ws_slab

    # ------------------------------------------------------------------------------------------------
    def forward(self, x):
        m = self.model
        m.train()
        self._forwards += 1
        runtime.TAPE = self.tape
        del self.tape[:]
        try:
            out = m(x)
        finally:
            runtime.TAPE = None
        return out

    def loss_and_grads(self, mask, point, direction, label, dirlab, point_t, weight):
        B, _, H, W = mask.shape
        lib = _lib.load()
        ND = direction.shape[1]                       # 5 / 9 / 17 direction classes (options.py:45)
        need = lib.cdnet_dam_loss_classes_workspace_floats(B, H * W, ND)
        if self._ws_loss is None or self._ws_loss.numel() < need:
            self._ws_loss = torch.empty((need,), dtype=torch.float32, device=self.dev)
        dmask = self.buf('dmask', mask.shape, torch.float32)
        dpoint = self.buf('dpoint', point.shape, torch.float32)
        ddir = self.buf('ddir', direction.shape, torch.float32)
        assert label.dtype == torch.uint8 and dirlab.dtype == torch.uint8 and weight.dtype == torch.uint8
        assert point_t.dtype == torch.float16
        _lib.call('cdnet_dam_loss_classes', _lib.ptr(mask), _lib.ptr(point), _lib.ptr(direction), _lib.ptr(label.contiguous()),
                  _lib.ptr(dirlab.contiguous()), _lib.ptr(point_t.contiguous()), _lib.ptr(weight.contiguous()), B, H, W, ND,
                  self.quirk, _lib.ptr(self._ws_loss), self._ws_loss.numel(), _lib.ptr(self.losses), _lib.ptr(dmask),
                  _lib.ptr(dpoint), _lib.ptr(ddir), _lib.stream_ptr())
        return dmask, dpoint, ddir

    # ------------------------------------------------------------------------------------------------
    def backward(self, dmask, dpoint, ddir):
        m = self.model
        f1, f2, f3 = m._last_feats
        N, H, W, _ = f1.x.shape
        grads = {}                      # id(stored tensor) -> [_G]

        def add(t, g):
            grads.setdefault(id(t), []).append(g)

        # head
        df = [self.buf('dF%d' % k, (N, H, W, 64), runtime.act_dtype()) for k in range(3)]
        hf = [runtime.head_feat(f) for f in (f1, f2, f3)]
        dhead = self.flat.G[:self.flat.n_head]
        need = _lib.load().cdnet_dam_head_backward_workspace_floats(N, H, W)
        if self._ws_head is None or self._ws_head.numel() < need:
            self._ws_head = torch.empty((need,), dtype=torch.float32, device=self.dev)
        _lib.call('cdnet_dam_head_backward', C.byref(hf[0]), C.byref(hf[1]), C.byref(hf[2]), _lib.ptr(m.head_weight_block()),
                  _lib.ptr(dmask), _lib.ptr(dpoint), _lib.ptr(ddir), N, H, W, _lib.ptr(df[0]), _lib.ptr(df[1]),
                  _lib.ptr(df[2]), _lib.ptr(self._ws_head), self._ws_head.numel(), _lib.ptr(dhead), _lib.stream_ptr())
        for f, d in zip((f1, f2, f3), df):
            add(getattr(f, 'grad_to', (f.x,))[0], _G(d, H, W))
        self._overlap_begin()
        self._overlap_done(None)
        self._backward_tape(grads, add)

    def _backward_tape(self, grads, add):
        """walk the forward tape backwards: BatchNorm(+ReLU, residual, pool/pad/concat routing) backward, then the
        weight and input gradients of every convolution that received a gradient"""
        self._cat_cache = {}
        # (a backward that raised midway - CDNET_REQUIRE failure, OOM - must not leave its deferred split-K descriptors to this one)
        self._rd_pending, self._rd_params, self._rd_bytes = [], [], 0
        if len(self._rd_tables) > 16:
            self._rd_tables.clear()                 # (one small device table per distinct call sequence: batch-size changes add entries)
        # who reads what: a BatchNorm layer whose raw output has exactly one reader - a 3x3 convolution - gets the first pass of its
        # backward (the channel sums) from that reader's backward-data launch (_input_backward), see _stats_fusable
        self._readers, self._producer, self._bn_partials = {}, {}, {}
        self._tape_pos = {id(Lt): k for k, Lt in enumerate(self.tape)}
        for Lt in self.tape:
            if isinstance(Lt, runtime.FuseNode):
                for t in Lt.saved[0]:                # (counted too: FuseNode.backward hands a residual sum's BatchNorm term its gradients unsummed
                    self._readers[id(t.x)] = self._readers.get(id(t.x), 0) + 1       # only when nothing else reads that term)
                continue
            for sx in Lt.saved[0]:
                self._readers[id(sx.x)] = self._readers.get(id(sx.x), 0) + 1
                if sx.res is not None:
                    self._readers[id(sx.res)] = self._readers.get(id(sx.res), 0) + 1
            self._producer[id(Lt.saved[1])] = Lt
        side = self._side_stream()
        self._side_active = side is not None
        if self._packb_pending:
            torch.cuda.current_stream().wait_event(self._event('packb'))      # backward-data packs made beside the forward
            self._packb_pending = False
        for k, L in enumerate(reversed(self.tape)):
            if isinstance(L, runtime.FuseNode):
                L.backward(self, grads, add)
                continue
            gl = self.take(grads, id(L.saved[1]))
            if gl is None:
                continue
            owner = getattr(L, 'fused_res_of', None)
            if owner is not None:
                # ResidualUnit.conv_1x1 with the fused residual epilogue: its stored output is the unit's output f.  The consumers'
                # gradients of f first pass bn2 + ReLU of the other branch (mask read from f); that dz is this layer's gradient.
                grads.setdefault(id(owner.saved[1]), []).extend(gl)
                owner.deferred_layer = L
                continue
            self._layer_backward(k, L, gl, grads, add, side)
            d = getattr(L, 'deferred_layer', None)
            if d is not None:
                L.deferred_layer = None
                self._layer_backward(('deferred', k), d, self.take(grads, id(d.saved[1])), grads, add, side)
        if side is not None:
            with _on_stream(side):
                self._flush_reduces()
            ev = self._event(-1)
            ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
        else:
            self._flush_reduces()

    def _weights_done(self, params):
        """the weight-gradient launches of a layer are queued (current stream = the one they run on): sum their slabs now or later, then
        hand the layer's parameters to the all-reduce"""
        if not self._rd_pending:
            self._overlap_done(params)
            return
        self._rd_params.append(params)
        if self._rd_bytes >= self._rd_mb * (1 << 20):
            self._flush_reduces()

    def _flush_reduces(self):
        """one cdnet_wgrad_reduce_batch launch over every deferred weight gradient (bit-identical to the per-call reduction).  The
        descriptor table of a launch is built once per distinct sequence of calls and kept on the device."""
        if self._rd_pending:
            key = tuple(self._rd_pending)
            tab = self._rd_tables.get(key)
            if tab is None:
                lib = _lib.load()
                descs = (_lib.WgradReduceDesc * len(key))()
                b0 = 0
                for i, a in enumerate(key):
                    _lib.call('cdnet_wgrad_reduce_desc_fill', *a, b0, C.byref(descs[i]))
                    b0 += descs[i].blocks
                dev_tab = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.dev)
                tab = self._rd_tables[key] = (dev_tab, len(key), b0)
            _lib.call('cdnet_wgrad_reduce_batch', _lib.ptr(tab[0]), tab[1], tab[2], _lib.stream_ptr())
        for params in self._rd_params:
            self._overlap_done(params)
        self._rd_pending, self._rd_params, self._rd_bytes = [], [], 0

    def _layer_backward(self, k, L, gl, grads, add, side):
        """one convolution layer: BatchNorm / residual / ReLU backward of its output, weight gradient (side stream), input gradient"""
        for g_ in gl:
            if g_.event is not None:         # a contribution computed beside the chain (_RU_1X1_SIDE)
                torch.cuda.current_stream().wait_event(g_.event)
                g_.event = None
        srcs, out, Hl, Wl = L.saved
        No, Ho, Wo, Co = out.shape
        params = (L.weight, L.bias, None if L.bn is None else L.bn.weight, None if L.bn is None else L.bn.bias)
        part = self._bn_partials.pop(id(out), None)
        if part is not None:
            # the reader's backward-data launch left the channel sums: finalize + second pass only
            assert len(gl) == 1, (L.name, len(gl))
            a, ktab = self._bn_backward_stats(L, out, gl[0], partial=part)
            g = self.buf(('draw', L.name), (No, Ho, Wo, Co), runtime.act_dtype())
            _lib.call('cdnet_bn_backward_apply', C.byref(a), _lib.ptr(ktab), _lib.ptr(g), _lib.stream_ptr())
        if part is not None:
            pass
        elif L.bn is not None or len(gl) > 1 or gl[0].pooled or gl[0].coff or (gl[0].cstride not in (0, Co)):
            g = self._bn_backward(L, out, gl, add)
        else:
            g = gl[0].t                                     # plain pass-through (conv_1x1 residual branch)
        if side is None:
            self._weight_backward(L, srcs, g, Hl, Wl)
            self._weights_done(params)
            self._input_backward(L, srcs, g, Hl, Wl, add)
        else:
            # the weight gradient only feeds the optimiser: it runs on a second stream beside the input-gradient chain.  The chain's
            # next kernel is issued first (under rocprofv3, whose interception slows every launch call, the chain otherwise sits idle
            # in the 512-channel layers while the host is still issuing the side stream's launches; without the profiler the host
            # keeps its lead either way)
            ev = self._event(k)
            ev.record()             # (recorded after backward-data instead, the weight gradient overlaps the HBM-bound BatchNorm passes of
            beside = _RU_1X1_SIDE and L.kind == 'conv1' and isinstance(k, tuple) and getattr(L, 'needs_input_grad', True)
            if not beside:
                self._input_backward(L, srcs, g, Hl, Wl, add)      # the next layer rather than the convolution: -2 %)
            side.wait_event(ev)
            with _on_stream(side):
                if beside:
                    # the 1x1 branch of a residual unit: its input gradient is first needed by the BatchNorm backward of the PREVIOUS unit,
                    # three convolutions down the chain - it runs beside the chain, in front of the weight gradients queued from here on
                    ev2 = self._event(('beside', k))

                    def add_beside(tgt, g_):
                        g_.event = ev2
                        add(tgt, g_)
                    self._input_backward(L, srcs, g, Hl, Wl, add_beside)
                    ev2.record()
                self._weight_backward(L, srcs, g, Hl, Wl)
                self._weights_done(params)         # (a bucket released here is ordered after both streams' work so far)

    def _bn_backward_stats(self, L, out, g, partial):
        """finalize pass over the partial channel sums a backward-data launch left (cdnet_conv_args.ws = 2, fp32 mode): dgamma, dbeta and
        the [7][C] table the second pass reads"""
        a = BnBwdArgs()
        No, Ho, Wo, Co = out.shape
        a.raw = out.data_ptr()
        a.res = None
        a.scale, a.shift = L.scale.data_ptr(), L.shift.data_ptr()
        a.mean, a.invstd = L.save_mean.data_ptr(), L.save_invstd.data_ptr()
        a.ngin = 1
        a.gin[0].g = g.t.data_ptr()
        a.gin[0].Hg, a.gin[0].Wg, a.gin[0].oy, a.gin[0].ox = g.Hg, g.Wg, 0, 0
        a.gin[0].pooled, a.gin[0].coff, a.gin[0].cstride = 0, 0, Co
        a.f16 = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[out.dtype]
        a.relu = 1
        a.N, a.H, a.W, a.C = No, Ho, Wo, Co
        ktab = self.buf(('ktab', L.name), (7, Co), torch.float32)
        bn = L.bn
        _lib.call('cdnet_bn_backward_finalize', C.byref(a), _lib.ptr(bn.weight.detach()), _lib.ptr(bn.weight.grad), _lib.ptr(bn.bias.grad),
                  _lib.ptr(partial), partial.shape[0], _lib.ptr(ktab), _lib.stream_ptr())
        return a, ktab

    def _stats_fusable(self, L, srcs, H, W, N, wpb, cfgb, cin_total):
        """the single lazily transformed BatchNorm + ReLU source of a 3x3 convolution that is its only reader, and a backward-data
        launch that runs on the producer / consumer kernel: returns (producer layer, partial-row buffer) or None"""
        # fp32 mode only (conv_ws32_kernel): the sums ride in the CONSUMERS' deferred epilogue (raw quarters by DMA into LDS).  Round 2's
        # 16-bit form - the sums accumulated by the movers of conv_ws_kernel - saved 11 reduce passes and was not faster (the movers'
        # per-element arithmetic costs the matrix pipe issue time); it was removed in round 3.
        f32 = runtime.PRECISION == 'fp32'
        if not f32 or os.environ.get('CDNET_BN_STATS_FUSE', '1') != '1' or getattr(runtime, 'DEBUG_NORELU', False):
            return None
        if L.kind != 'conv3' or L.transposed or len(srcs) != 1:
            return None
        sx = srcs[0]
        if getattr(sx, 'is_input', False) or sx.scale is None or sx.shift is None or sx.relu is not True or sx.res is not None or sx.pool \
                or tuple(sx.off) != (0, 0) or sx.x.dtype != runtime.raw_dtype() or hasattr(sx, 'grad_to') or sx.row_stride:
            return None
        if f32 and cin_total % 64:
            return None
        P = self._producer.get(id(sx.x))
        if P is None or P.bn is None or getattr(P, 'node_res', None) is not None or getattr(P, 'node_relu', True) is not True \
                or self._readers.get(id(sx.x), 0) != 1 or tuple(sx.x.shape) != (N, H, W, cin_total):
            return None
        key = ('statsfusable', L.name, N, H, W, runtime.PRECISION)
        hit = self._bufs.get(key)
        if hit is None:
            part = torch.zeros((1024, 2, cin_total), dtype=torch.float32, device=self.dev)
            gin = self.buf(('din', L.name), (N, H, W, cin_total), runtime.act_dtype())
            gdummy = self.buf(('draw', L.name), (N, H, W, L.Cout), runtime.act_dtype())
            ok = engine.conv_forward([Src(gdummy)], wpb, cin_total, cfgb, taps=L.taps, out=gin, H=H, W=W, query_ws=True,
                                     bns=(sx.x, P.scale, P.shift, P.save_mean, P.save_invstd, part))
            hit = self._bufs[key] = (part if ok else False)
        if hit is False:
            return None
        return P, hit

    def _bn_backward(self, L, out, gl, add):
        a = BnBwdArgs()
        No, Ho, Wo, Co = out.shape
        a.raw = out.data_ptr()
        res = getattr(L, 'node_res', None)
        a.res = None if res is None else res.data_ptr()
        has_bn = L.bn is not None
        a.scale = L.scale.data_ptr() if has_bn else None
        a.shift = L.shift.data_ptr() if has_bn else None
        a.mean = L.save_mean.data_ptr() if has_bn else None
        a.invstd = L.save_invstd.data_ptr() if has_bn else None
        if len(gl) > 3:
            # more consumers than the kernel takes gradient sources (an ablation head's first residual unit feeds two units, four
            # backward-data terms): fold the plain same-size terms beyond the second into one tensor first
            plain = [g for g in gl if not g.pooled and g.oy == 0 and g.ox == 0 and (g.Hg, g.Wg) == (Ho, Wo)]
            rest = [g for g in gl if g not in plain]
            assert len(rest) <= 2 and len(plain) >= 2, (L.name, len(gl))
            d = self.buf(('gsum', L.name), (No, Ho, Wo, Co), runtime.act_dtype())
            self.grad_sum(plain, None, No * Ho * Wo, Co, d)
            gl = rest + [_G(d, Ho, Wo)]
        a.ngin = len(gl)
        assert 1 <= len(gl) <= 3, (L.name, len(gl))
        for k, g in enumerate(gl):
            a.gin[k].g = g.t.data_ptr()
            a.gin[k].Hg, a.gin[k].Wg, a.gin[k].oy, a.gin[k].ox = g.Hg, g.Wg, g.oy, g.ox
            a.gin[k].pooled, a.gin[k].coff, a.gin[k].cstride = int(g.pooled), g.coff, g.cstride or Co
        a.f16 = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[out.dtype]
        a.relu = int(getattr(L, 'node_relu', True))
        a.N, a.H, a.W, a.C = No, Ho, Wo, Co
        draw = self.buf(('draw', L.name), (No, Ho, Wo, Co), runtime.act_dtype())
        dz = None
        if res is not None:
            dz = self.buf(('dz', L.name), (No, Ho, Wo, Co), runtime.act_dtype())
        ws = self._bn_ws(Co)
        bn = L.bn
        _lib.call('cdnet_bn_backward', C.byref(a), _lib.ptr(bn.weight.detach()) if has_bn else None,
                  _lib.ptr(bn.weight.grad) if has_bn else None, _lib.ptr(bn.bias.grad) if has_bn else None,
                  _lib.ptr(ws), ws.numel(), _lib.ptr(draw), _lib.ptr(dz), _lib.stream_ptr())
        if res is not None:
            # gradient of the residual branch = dz; its producer (conv_1x1) is on the tape.  (HRNet's residual sums: `res` is the stored
            # output the mask is read from, the branch is the block's input - FuseNode.backward names it)
            add(getattr(L, 'node_res_grad_to', None) if getattr(L, 'node_res_grad_to', None) is not None else res, _G(dz, Ho, Wo))
            L.node_res_grad_to = None
        return draw

    def _side_stream(self):
        """second HIP stream for the weight-gradient kernels (trainer.WGRAD_STREAM = False: everything on one stream)"""
        if not WGRAD_STREAM or self.dev.type != 'cuda':
            return None
        if self._wstream is None:
            self._wstream = streams.side_stream(self.dev)      # (probed: a stream on another hardware queue than the compute stream's)
            self._events = {}
        return self._wstream

    def _event(self, k):
        e = self._events.get(k)
        if e is None:
            e = self._events[k] = torch.cuda.Event()
        return e

    def _weight_backward(self, L, srcs, g, H, W):
        lib = _lib.load()
        N = g.shape[0]
        Cout = L.Cout
        cin_total = sum(s.C for s in srcs)
        cin_real = L.Cin
        mode = {'conv3': 0, 'conv1': 0, 'convT4': 2, 'convT2': 3, 'conv3s2': 6}[L.kind]
        taps, npar, ostride = L.taps, (4 if L.transposed else 1), (2 if L.transposed else 1)
        coff = 0
        for s in srcs:
            ci_t = _choose_ci_tiles(s.C, Cout)
            CI, CO = ci_t * 32, (4 // ci_t) * 32
            other = -(-s.C // CI) * -(-Cout // CO) * npar
            ntiles = N * (-(-H // 8)) * (-(-W // 16))
            # beside the input-gradient chain the weight-gradient kernels take fewer workgroups than there are CUs: a full grid of
            # them holds every CU's LDS, and the chain's producer / consumer convolutions (one 157 KB workgroup per CU) then queue
            # behind it - measured 1 663 -> 1 745 / 1 730 -> 1 813 tiles/s (two boxes).  fp32 mode: 160 workgroups since the
            # end of round 3 (+0.7 %; with round 2's kernels the caps cost 1.7 %).
            if not self._side_active or not getattr(L, 'needs_input_grad', True):
                cap = 256                                   # (the first layer's weight gradient runs after the chain has ended: whole chip)
            elif runtime.PRECISION == 'fp32':
                cap = _WGRAD_WGS_F32
            else:
                cap = _WGRAD_WGS_DEEP if H * W <= _WGRAD_DEEP_HW or H * W > 65536 else _WGRAD_WGS_SHALLOW    # (512 x 512 layers of HRNet: 128 again)
                if Cout <= 32 and H * W > 65536:
                    cap = _WGRAD_WGS_KQ
            ksplit = max(1, min(ntiles, cap // other if other < cap else 1))      # one 8-wave workgroup per CU
            nslab = lib.cdnet_conv_wgrad_slab_floats(s.C, Cout, taps, npar, ci_t, ksplit)
            defer = self._rd_mb > 0
            slab = self.buf(('wslab', L.name, coff), (nslab,), torch.float32) if defer else self._slab(nslab)
            cs = engine.ConvSrc()
            s.fill(cs)
            csrc_real = min(s.C, cin_real - coff) if cin_total != cin_real else s.C
            _lib.call('cdnet_conv_backward_weight', C.byref(cs), coff, csrc_real, cin_real, _lib.ptr(g), Cout, N, H, W, taps,
                      npar, ostride, ci_t, ksplit, _lib.ptr(slab), _lib.ptr(L.weight.grad), mode | (_WGRAD_DEFER if defer else 0),
                      _lib.stream_ptr())
            if defer:
                self._rd_pending.append((s.C, coff, csrc_real, cin_real, Cout, taps, npar, ci_t, ksplit, slab.data_ptr(),
                                         L.weight.grad.data_ptr(), mode))
                self._rd_bytes += nslab * 4
            coff += s.C
        if L.bias is not None and L.bn is None:
            # bias of a BN-less conv (ResidualUnit.conv_1x1): sum of its output gradient == dbeta of the unit's bn2
            owner = getattr(L, 'bias_grad_from', None)
            if owner is not None:
                L.bias.grad.copy_(owner.bn.bias.grad)
            else:
                # stand-alone biased convolution (plain UNet's ConvTranspose2d): db = sum of the output gradient
                need = lib.cdnet_bias_grad_workspace_floats(Cout)
                ws = self._slab(need)
                _lib.call('cdnet_bias_grad_f32' if g.dtype == torch.float32 else 'cdnet_bias_grad', _lib.ptr(g), g.numel() // Cout, Cout,
                          _lib.ptr(ws), ws.numel(), _lib.ptr(L.bias.grad), _lib.stream_ptr())

    def _input_backward(self, L, srcs, g, H, W, add):
        """g: the gradient w.r.t. the layer's raw output, or a prepared Src (BatchNorm-backward source of the fused path)"""
        if not getattr(L, 'needs_input_grad', True):
            return
        N = g.N if isinstance(g, Src) else g.shape[0]
        Cout = L.Cout
        cin_total = sum(s.C for s in srcs)
        # input gradient: forward convolution with the backward-data pack
        wpb, cfgb = L.backward_pack(cin_total, H, W)
        if L.kind == 'conv3s2':
            # stride-2 convolution: gradient in the space-to-depth layout [N,H,W,(a,b,c)], then permuted to [N,2H,2W,C]
            Cp = cin_total // 4
            gs2d = self.buf(('ds2d', L.name), (N, H, W, cin_total), runtime.act_dtype())
            engine.conv_forward([Src(g)], wpb, cin_total, cfgb, taps=9, out=gs2d, H=H, W=W)
            gin = self.buf(('din', L.name), (N, 2 * H, 2 * W, Cp), runtime.act_dtype())
            _lib.call('cdnet_s2d_to_nhwc_f32' if gin.dtype == torch.float32 else 'cdnet_s2d_to_nhwc', _lib.ptr(gs2d), N, H, W, Cp, _lib.ptr(gin),
                      _lib.stream_ptr())
            add(srcs[0].x, _G(gin, 2 * H, 2 * W))
            return
        if not L.transposed:
            gin = self.buf(('din', L.name), (N, H, W, cin_total), runtime.act_dtype())
            fuse = None if isinstance(g, Src) else self._stats_fusable(L, srcs, H, W, N, wpb, cfgb, cin_total)
            if fuse is not None:
                # this launch also leaves the channel sums of the BatchNorm backward of its only source's producer
                P, part = fuse
                engine.conv_forward([Src(g)], wpb, cin_total, cfgb, taps=L.taps, out=gin, H=H, W=W,
                                    bns=(srcs[0].x, P.scale, P.shift, P.save_mean, P.save_invstd, part))
                self._bn_partials[id(srcs[0].x)] = part
            else:
                engine.conv_forward([g if isinstance(g, Src) else Src(g)], wpb, cin_total, cfgb, taps=L.taps, out=gin, H=H, W=W)
        else:
            # space-to-depth view of g [N,2H,2W,Cout]: two row-parity sources of 2*Cout channels each
            gin = self.buf(('din', L.name), (N, H, W, cin_total), runtime.act_dtype())
            views = [Src(g, view=(a * 2 * W * Cout, H, W, 2 * Cout, 4 * W * Cout)) for a in (0, 1)]
            engine.conv_forward(views, wpb, cin_total, cfgb, taps=(9 if L.kind == 'convT4' else 1), out=gin, H=H, W=W)
        coff = 0
        for s in srcs:
            if getattr(s, 'is_input', False):
                coff += s.C
                continue
            tgt, pool = getattr(s, 'grad_to', (s.x, int(s.pool)))          # a materialised max-pool hands its gradient to the producer
            add(tgt, _G(gin, H, W, oy=s.off[0], ox=s.off[1], pooled=pool, coff=coff, cstride=cin_total))
            coff += s.C

    # ------------------------------------------------------------------------------------------------
    # Gradient all-reduce overlapped with backward.  The flat gradient buffer is laid out in forward order, backward
    # fills it from the end: after every layer, each fixed-size bucket that lies completely above the highest parameter
    # still waiting for its gradient is handed to RCCL (async: the collective waits for the kernels already queued on
    # the compute stream and runs beside the rest of backward).  CDNET_ALLREDUCE_OVERLAP=0 keeps the single
    # post-backward pass.
    def _overlap_begin(self):
        self._ar = None
        if not self._reduce_active() or os.environ.get('CDNET_ALLREDUCE_OVERLAP', '1') == '0':
            return
        f = self.flat
        base = f.G.data_ptr()
        pend = {}
        owners = [p for L in self.tape if not isinstance(L, runtime.FuseNode) for p in (L.weight, L.bias, None if L.bn is None else L.bn.weight,
                                                 None if L.bn is None else L.bn.bias) if p is not None]
        for p in owners:
            off = (p.grad.data_ptr() - base) // 4
            if 0 <= off < f.n_used:
                pend[off] = off + p.numel()
        if f.n_head:
            pend[0] = f.n_head                       # the head block, written by the head backward kernel
        self._ar = BucketReducer(f.G, f.n_used, self.bucket, pend)

    def _overlap_done(self, params):
        """the gradients of `params` (or the head block when None) are final: launch every bucket now complete"""
        if self._ar is None:
            return
        if params is None:
            self._ar.done([0])
        else:
            base = self.flat.G.data_ptr()
            self._ar.done([(p.grad.data_ptr() - base) // 4 for p in params if p is not None])

    def _reduce_active(self):
        return self.world > 1 or os.environ.get('CDNET_FORCE_ALLREDUCE', '0') == '1'

    def allreduce_and_step(self):
        f = self.flat
        gscale = 1.0
        ranges = [(0, f.n_used)]
        if self._reduce_active():
            if getattr(self, '_ar', None) is not None:
                # whatever backward did not release yet; then the optimiser follows the collectives bucket by bucket (top-down, the
                # order they were launched in): the last bucket's all-reduce - the first layers' gradients, complete only when backward
                # ends - runs while Adam already updates the ranges above it
                ranges = self._ar.finish(wait=False)
                works = self._ar.works
                self.ar_stats = dict(buckets=len(works), released_during_backward=self._ar.early)
                self._ar = None
            else:
                bucketed_allreduce(f.G, f.n_used, self.bucket)
                works = None
            gscale = 1.0 / self.world
        else:
            works = None
        f.step_count += 1

        def adam(a, b):
            sl = slice(a, b)
            _lib.call('cdnet_adam_step', _lib.ptr(f.P[sl]), _lib.ptr(f.G[sl]), _lib.ptr(f.M[sl]), _lib.ptr(f.V[sl]), b - a, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.wd, f.step_count, gscale, _lib.stream_ptr())
        if works is None:
            adam(0, f.n_used)
        else:
            for w, (a, b) in zip(works, ranges):
                w.wait()                                 # (the current stream waits, not the host)
                adam(a, b)
        runtime.WEIGHTS_EPOCH[0] += 1
        self._repack_all()

    def _repack_all(self):
        """Re-pack every layer's forward / backward-data weights in one launch (they would otherwise be re-packed one by
        one, lazily, by the next forward and backward).  The job table is built after the first step, once every layer
        has its packed buffers."""
        f = self.flat
        lo, hi = f.P.data_ptr(), f.P.data_ptr() + f.P.numel() * 4
        if self._pack_jobs is not None and getattr(self, '_pack_prec', None) != runtime.PRECISION:
            self._pack_jobs = None                           # the layers re-made their packs for the other precision
        if self._pack_jobs is None:
            groups = ([], []), ([], [])                      # (jobs, owners) of the forward packs / the backward-data packs
            for L in runtime.LAYERS:
                if not (lo <= L.weight.data_ptr() < hi):
                    continue
                w = L.weight.detach()
                if L.wp is not None and not L.wp_padded:
                    groups[0][0].append(engine.pack_job(w, L.cfg, L.pack_mode, L.wp, split=L.cfg_f32)); groups[0][1].append((L, 'wp_version'))
                if L.wpb is not None:
                    mode = (4 if L.kind == 'convT4' else 5) if L.transposed else (7 if L.kind == 'conv3s2' else 1)
                    groups[1][0].append(engine.pack_job(w, L.cfg_bwd, mode, L.wpb, split=L.cfg_f32)); groups[1][1].append((L, 'wpb_version'))
            if not groups[0][0] or not groups[1][0] or self.flat.step_count < 1:
                return
            self._pack_jobs = []
            self._pack_prec = runtime.PRECISION
            for jobs, owners in groups:
                arr = (engine.PackJob * len(jobs))(*jobs)
                nbytes = _lib.load().cdnet_pack_batch_table_bytes(len(jobs))
                table = torch.empty((nbytes,), dtype=torch.uint8, device=f.P.device)
                self._pack_jobs.append((arr, len(jobs), table, owners, [True]))
        side = self._side_stream()
        for k, (arr, n, table, owners, first) in enumerate(self._pack_jobs):
            if k == 1 and side is not None and self._packb_stream:
                # the backward-data packs are not needed before the next backward: pack them beside the next forward
                ev = self._event('adam')
                ev.record()
                side.wait_event(ev)
                with _on_stream(side):
                    _lib.call('cdnet_pack_conv_weights_batch', C.byref(arr), n, _lib.ptr(table), table.numel(), int(first[0]), _lib.stream_ptr())
                    self._event('packb').record()
                self._packb_pending = True
            else:
                _lib.call('cdnet_pack_conv_weights_batch', C.byref(arr), n, _lib.ptr(table), table.numel(), int(first[0]), _lib.stream_ptr())
            first[0] = False
            for L, attr in owners:
                setattr(L, attr, (L.weight._version, runtime.WEIGHTS_EPOCH[0]))

    # ------------------------------------------------------------------------------------------------
    # Optimiser state in torch.optim.Adam's state_dict format (what the reference stores under checkpoint['optimizer'],
    # train.py:421-427, and reads back at :302): parameter indices follow model.parameters(); parameters that never received
    # a gradient have no entry (Adam creates state lazily).
    def _real_pieces(self, buf, name, p):
        """[(index into the real parameter, view of `buf`)] covering parameter `name` (the model's own shape), whether the flat
        storage holds it as is, zero-padded (leading corner) or scattered over channel segments (HRNet)"""
        off, sz = self.flat.offsets[name]
        slots = {id(real): (pp, segs) for real, pp, segs in getattr(self.model, '_slots', [])}
        if id(p) not in slots:
            return [(Ellipsis, buf[off:off + sz].view(p.shape))]
        pp, segs = slots[id(p)]
        full = buf[off:off + sz].view(pp.shape)
        if segs is None:
            return [(Ellipsis, full[tuple(slice(0, n) for n in p.shape)])]
        return [((slice(None), slice(r0, r0 + n)), full[:, p0:p0 + n]) for r0, n, p0 in segs]

    def write_bn_counters(self):
        """nn.BatchNorm2d.num_batches_tracked of every BatchNorm that ran = the number of training forwards (PyTorch adds one per
        forward, `nn.BatchNorm2d.forward`); kept as a host counter during training and written out for checkpoints"""
        with torch.no_grad():
            unused = tuple(getattr(self.model, 'UNUSED_PREFIXES', ()))
            for name, mod in self.model.named_modules():
                if isinstance(mod, torch.nn.BatchNorm2d) and mod.num_batches_tracked is not None and not (name + '.').startswith(unused or ('\0',)):
                    mod.num_batches_tracked.fill_(self._bn_base + self._forwards)

    def refresh_parameters(self):
        """call after writing parameters behind the trainer's back (load_state_dict does not need it: the module's parameters
        ARE views of the flat buffer; this only invalidates every packed bf16 weight copy and BatchNorm fold)"""
        if hasattr(self.model, 'sync_real_parameters') and getattr(self.model, '_rt', None) is not None:
            self.model._ensure_runtime()
        runtime.WEIGHTS_EPOCH[0] += 1

    def state_dict(self):
        f = self.flat
        params = list(self.model.named_parameters())
        state = {}
        if f.step_count > 0:
            for i, (n, p) in enumerate(params):
                if f.offsets[n][0] >= f.n_used:
                    continue                                        # the reference's never-used parameters: no gradient, no state
                m, v = torch.empty(p.shape, dtype=torch.float32), torch.empty(p.shape, dtype=torch.float32)
                for idx, piece in self._real_pieces(f.M, n, p):
                    m[idx] = piece.cpu()
                for idx, piece in self._real_pieces(f.V, n, p):
                    v[idx] = piece.cpu()
                state[i] = {'step': torch.tensor(float(f.step_count)), 'exp_avg': m, 'exp_avg_sq': v}
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.wd, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None, params=list(range(len(params))))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        f = self.flat
        params = list(self.model.named_parameters())
        group = sd['param_groups'][0]
        assert len(sd['param_groups']) == 1 and len(group['params']) == len(params), 'optimizer state of a different model'
        self.lr, self.betas, self.eps, self.wd = group['lr'], tuple(group['betas']), group['eps'], group['weight_decay']
        f.M.zero_()
        f.V.zero_()
        steps = set()
        for i, st in sd['state'].items():
            n, p = params[int(i)]
            assert f.offsets[n][0] < f.n_used, 'state for a parameter that is never stepped: ' + n
            for idx, piece in self._real_pieces(f.M, n, p):
                piece.copy_(st['exp_avg'][idx])
            for idx, piece in self._real_pieces(f.V, n, p):
                piece.copy_(st['exp_avg_sq'][idx])
            steps.add(int(st['step']))
        assert len(steps) <= 1, 'per-parameter step counts differ: not a state of torch.optim.Adam over the whole model'
        f.step_count = steps.pop() if steps else 0

    def train_step(self, x, label, dirlab, point_t, weight):
        """x f32 [B,3,H,W]; label u8 [B,H,W] in {0,1,2}; dirlab u8 [B,H,W] 0..8; point_t f16 [B,H,W]; weight u8 [B,H,W]
        (the png weight map; /20 on the fly).  Returns the device tensor of 11 values: [total, direction CE, direction dice,
        MSE, CE, dice, pixel accuracy, IoU, recall, precision, F1] (train_util_dam.py:297-299; slot 5 holds the mask dice
        term where the reference logs its unused variance term)."""
        mask, point, direction = self.forward(x)
        dmask, dpoint, ddir = self.loss_and_grads(mask, point, direction, label, dirlab, point_t, weight)
        self.backward(dmask, dpoint, ddir)
        self.allreduce_and_step()
        return self.losses


class UNetTrainer(Trainer):
    """Body of the plain-UNet train iteration (train_util.py:58-200 with the default options: log-softmax + NLL x weight
    map mean, + MulticlassDiceLoss on the softmax, alpha = 0, no boundary loss) -> backward -> Adam.
    The two loss terms are exactly the mask terms of the DAM loss kernel, which is fed constant point / direction
    branches here; `losses` = [total, CE x weight, dice]."""

    def __init__(self, model, **kw):
        super().__init__(model, **kw)
        self.unet_losses = torch.zeros((3,), dtype=torch.float32, device=self.dev)

    def loss_and_grads(self, logits, label, weight):
        B, K, H, W = logits.shape
        assert K == 3, 'the fused loss serves the 3-class configuration (options.py: out_c = 3)'
        z = lambda shape, dt: self.buf(('zero',) + tuple(shape) + (dt,), shape, dt)
        point, dirn = z((B, 1, H, W), torch.float32), z((B, 9, H, W), torch.float32)
        dirlab, point_t = z((B, H, W), torch.uint8), z((B, H, W), torch.float16)
        for t in (point, dirn, dirlab, point_t):
            t.zero_()
        dmask, _, _ = super().loss_and_grads(logits, point, dirn, label, dirlab, point_t, weight)
        self.unet_losses[1:3] = self.losses[4:6]
        self.unet_losses[0] = self.losses[4] + self.losses[5]
        return dmask

    def backward(self, dlogits):
        m = self.model
        feat = m._last_feat
        N, H, W, _ = feat.x.shape
        grads = {}

        def add(t, g):
            grads.setdefault(id(t), []).append(g)
        K = m.num_classes
        df = self.buf('dF', (N, H, W, 64), runtime.act_dtype())
        lib = _lib.load()
        ws = self._slab(lib.cdnet_final_conv1x1_backward_workspace_floats())
        hf = runtime.head_feat(feat)
        w = m.final_conv.weight.detach().reshape(K, 64)
        _lib.call('cdnet_final_conv1x1_backward', C.byref(hf), _lib.ptr(w), _lib.ptr(dlogits), K, N, H, W, _lib.ptr(df), _lib.ptr(ws),
                  ws.numel(), _lib.ptr(m.final_conv.weight.grad), _lib.ptr(m.final_conv.bias.grad), _lib.stream_ptr())
        add(feat.x, _G(df, H, W))
        self._overlap_begin()
        self._overlap_done([m.final_conv.weight, m.final_conv.bias])
        self._backward_tape(grads, add)

    def train_step(self, x, label, weight):
        """x f32 [B,3,H,W]; label u8 [B,H,W] in {0,1,2}; weight u8 [B,H,W] (png weight map, /20 on the fly,
        train_util.py:109).  Returns the device tensor [total, CE, dice]."""
        logits = self.forward(x)
        dlogits = self.loss_and_grads(logits, label, weight)
        self.backward(dlogits)
        self.allreduce_and_step()
        return self.unet_losses


class AblationTrainer(Trainer):
    """Train iteration of the ablation heads (models/dam/model_unet_MandD.py / model_unet_MandDandP.py through
    train_util_dam.train, which unpacks the model's outputs by their number, :152-166): the same five-term loss - without the point
    term for the two-output model (options direction = 1, mseloss = 0) - plain 1x1 classifiers instead of the gated head
    (cdnet_final_conv1x1 / cdnet_final_conv1x1_backward), everything else the rev1 tape.  The direction branch has 5, 9 or 17
    classes (model_unet_MandD4 / MandD / MandD16; cdnet_dam_loss_classes)."""

    def __init__(self, model, **kw):
        assert getattr(model, 'VARIANT', 'rev1') in ('MandD', 'MandDandP') and model.DIRECTION_OUT in (5, 9, 17), \
            'AblationTrainer serves model_unet_MandD / MandD4 / MandD16 / MandDandP'
        super().__init__(model, **kw)

    def loss_and_grads(self, outputs, label, dirlab, point_t, weight):
        if len(outputs) == 3:
            return super().loss_and_grads(outputs[0], outputs[1], outputs[2], label, dirlab, point_t, weight)
        mask, direction = outputs
        B, _, H, W = mask.shape
        point = self.buf('zero_point', (B, 1, H, W), torch.float32)
        pt = self.buf('zero_point_t', (B, H, W), torch.float16)
        point.zero_()
        pt.zero_()
        dmask, _, ddir = super().loss_and_grads(mask, point, direction, label, dirlab, pt, weight)
        return dmask, None, ddir

    def backward(self, dmask, dpoint, ddir):
        m = self.model
        f1m, f2, f3 = m._last_feats
        N, H, W, _ = f1m.x.shape
        grads = {}

        def add(t, g):
            grads.setdefault(id(t), []).append(g)
        lib = _lib.load()
        ws = self._slab(lib.cdnet_final_conv1x1_backward_workspace_floats())
        params = []
        for name, f, conv, dl in (('m', f1m, m.mask_conv, dmask), ('d', f2, m.direction_conv, ddir), ('p', f3, m.point_conv, dpoint)):
            if f is None or dl is None:
                continue
            K = conv.out_channels
            df = self.buf('dF' + name, (N, H, W, 64), runtime.act_dtype())
            hf = runtime.head_feat(f)
            w = conv.weight.detach().reshape(K, 64)
            _lib.call('cdnet_final_conv1x1_backward', C.byref(hf), _lib.ptr(w), _lib.ptr(dl), K, N, H, W, _lib.ptr(df), _lib.ptr(ws),
                      ws.numel(), _lib.ptr(conv.weight.grad), _lib.ptr(conv.bias.grad), _lib.stream_ptr())
            add(getattr(f, 'grad_to', (f.x,))[0], _G(df, H, W))
            params += [conv.weight, conv.bias]
        self._overlap_begin()
        self._overlap_done(params)
        self._backward_tape(grads, add)

    def train_step(self, x, label, dirlab, point_t, weight):
        out = self.forward(x)
        g = self.loss_and_grads(out, label, dirlab, point_t, weight)
        self.backward(*g)
        self.allreduce_and_step()
        return self.losses


class BucketReducer:
    """Releases buckets of a flat gradient buffer to the all-reduce as soon as they are complete.
    The buffer is laid out in forward order and backward fills it from the end: `pending` maps the start offset of
    every tensor that still waits for its gradient to its end offset; a bucket [a, b) is launched (async all-reduce,
    top-down) once no pending tensor reaches into or above it.  Bucket boundaries are counted from the TOP of the used
    range (n, n - B, n - 2B, ..., 0): the remainder bucket is then the lowest one - the one that completes last, with the
    first layers' gradients at the very end of backward, and whose collective nothing is left to hide (59 MB of
    gradients in 25 MB buckets: a 6 MB tail instead of a 25 MB one).  Every rank runs the same schedule, so the
    collectives are issued in the same order everywhere.  Backend-agnostic (RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, flat, n_used, bucket_elems, pending):
        self.flat, self.n, self.bucket = flat, n_used, bucket_elems
        self.pending = dict(pending)
        self.works = []
        self.bounds = [n_used]                       # descending bucket boundaries
        while self.bounds[-1] > 0:
            self.bounds.append(max(0, self.bounds[-1] - bucket_elems))
        self.next = 0                                # buckets [bounds[j + 1], bounds[j]) with j < next are in flight
        self.early = 0                               # buckets released before finish() (overlap actually happened)

    def _launch(self, top):
        import torch.distributed as dist
        while self.next + 1 < len(self.bounds) and self.bounds[self.next + 1] >= top:
            a, b = self.bounds[self.next + 1], self.bounds[self.next]
            self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self.next += 1

    def done(self, offsets):
        for off in offsets:
            self.pending.pop(off, None)
        before = len(self.works)
        self._launch(max(self.pending.values()) if self.pending else 0)
        self.early += len(self.works) - before

    def finish(self, wait=True):
        """launch what is left; wait=False returns the buckets' ranges in launch order instead (the caller waits per bucket)"""
        self._launch(0)
        if wait:
            for w in self.works:
                w.wait()
        return [(self.bounds[j + 1], self.bounds[j]) for j in range(len(self.works))]


def bucketed_allreduce(flat, n, bucket_elems):
    """Sum-all-reduce of the first n elements of a flat gradient buffer in fixed-size buckets (RCCL over xGMI on the GPU,
    gloo in the CPU tests).  The reference's nn.DataParallel reduce_add of the replicas' gradients (train.py:185) becomes
    one process per GPU + this call; unused parameters sit beyond n and are never communicated."""
    import torch.distributed as dist
    works = []
    for off in range(0, n, bucket_elems):
        works.append(dist.all_reduce(flat[off:min(n, off + bucket_elems)], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()


# ----------------------------------------------------------------------------------------------------------
def synthetic_batch(B, dev, seed=2022, H=256, W=256):
    """SURVEY 8d recipe: uniform RGB tiles, ellipse instances -> 3-class label / centripetal classes / point map,
    constant weight map 20."""
    import numpy as np
    from . import synth
    rs = np.random.RandomState(seed)
    x = (rs.randint(0, 256, size=(B, 3, H, W)).astype(np.float32) / 255.0)
    lab = np.zeros((B, H, W), np.uint8)
    dirn = np.zeros((B, H, W), np.uint8)
    point = np.zeros((B, H, W), np.float16)
    for b in range(B):
        inst = synth.ellipse_instances(H, W, 60, rs, 5, 12, 10)
        inside = inst > 0
        ero = synth.erode8(inside)
        lab[b][ero] = 1
        lab[b][inside & ~ero] = 2
        d, cents = synth.centroid_direction(inst)
        d[~ero] = 0
        dirn[b] = d
        pt = np.zeros((H, W), np.float64)
        for cy, cx in cents:
            pt[cy, cx] = 255.0
        point[b] = synth.gaussian_blur(pt, 2.0).astype(np.float16)
    weight = np.full((B, H, W), 20, np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(x), t(lab), t(dirn), t(point), t(weight)


def make_bench_step(model, B, dev, rank, world):
    """the benchmark's training step on a fixed synthetic batch: eager launches (a HIP-graph replay of forward + loss + backward was measured
    6-20 % slower on this step - the GPU, not the launch path, bounds it; cdnet_amd.graphs stays for the launch-bound HRNet step,
    tools/bench_hrnet.py)"""
    tr = Trainer(model, world_size=world)
    batch = synthetic_batch(B, dev, seed=2022 + rank)

    def step():
        return tr.train_step(*batch)
    metric = 'tiles/sec (train fwd+bwd+Adam), 256x256'
    workload = ('CDNet UNet2RevA1_vgg16 (UNet+DAM) training step: forward, 5-term loss, backward%s, %s fused Adam; '
                '256x256x3 synthetic tiles, batch %d per GPU' % ('', 'RCCL gradient all-reduce,' if world > 1 else '', B))
    return step, metric, workload
