"""The 3x3 convolutions of the RL train step (train_RL.py:103-122 -> network.py:28,30: 38 convolutions 256 -> 256, fp32, forward + backward) on the matrix cores at
fp32's accuracy: hi + lo f16 operands (22 bits of mantissa), three MFMAs per product, f32 accumulation (csrc/sz_nn_split.hip).  f16's range is handled by exact
power-of-two scaling: the weights are packed times 2^10, every board (forward / backward-data) or the whole tensor (weight gradient) is scaled so that its largest
magnitude lands in [2^11, 2^12), the output is scaled back — gradients of 1e-7 and activations of 1e+3 are treated alike.
  forward, backward-data  k_conv3x3_split_f32: the inference tower's K loop, one board per workgroup (two workgroups per board at up to #CUs/2 boards); backward-data is
                          the same convolution of the output gradient with the weights transposed and flipped; both weight streams are packed on the device by one
                          launch per convolution and step;
  weight gradient         k_wgrad3x3_split + k_wgrad_reduce: the MFMA's reduction dimension is the position (a lane's 8 k-elements = one board row), the x block is
                          staged per board in LDS in three column-shifted copies (the next board's loads in flight under this board's MFMAs), 16 board groups write
                          partial sums that a second kernel adds.
One call through the C ABI per direction (sz_nn_conv3x3_train_fwd / _bwd): the host's time per convolution counts as much as the GPU's here.
  BatchNorm + skip + ReLU  k_bn_act_fwd / k_bn_act_bwd (csrc/sz_train.hip): one launch per direction and site; ConvBNAct makes convolution + BatchNorm + skip + ReLU ONE autograd node.
MIOpen's fp32 kernels take 90-105 us (forward), 216 us (backward) per convolution at batch 128 — 78 % of an optimiser step; these take 41 us and 95 us.

    with split_convs(model):            # or enable_split_convs(model) / disable_split_convs(model)
        loss, mse, ce = train_rl.loss_fn(model, batch, device); loss.backward()

`train_rl.train` switches it on by default for an fp32 model on a GPU (`split_convs=False` / `--train-convs torch`: MIOpen).  Measured at batch 128
(tools/trainconv_probe.py, profiles/r03zze_trainconv_probe.txt, r03zzr_train_loop_same_box_ab_counters.txt): optimiser step 12.9 -> 6.8-7.3 ms; forward / input gradient / weight gradient of one convolution
5.0e-7 / 5.1e-7 / 2.7e-7 relative L2 from fp64 (torch fp32: 4.9e-7 / 5.1e-7 / 2.5e-7), also on inputs scaled by 1e3 or 1e-6; whole-network gradient 3.8e-3 from an fp64
step (3.45e-3 with torch's BatchNorm launches; MIOpen's fp32 step: 3.37e-3; the 39 train-mode BatchNorms amplify every rounding).  OPERANDS_F16 = False selects hi + lo bf16 operands for forward / backward-data
(16 bits: 4.5e-6 per convolution, gradient 1.1e-2) with torch's weight gradient; WGRAD_KERNEL = False keeps torch's weight gradient.
"""
import contextlib
import ctypes as C

import torch

from . import _native as N

_STREAM_BYTES = 72 * 2048 * 16
_scratch = {}
OPERANDS_F16 = True          # hi + lo f16 operands with power-of-two scaling (22 bits: fp32's class); False: hi + lo bf16 (16 bits)


def _bufs(dev):
    key = (dev.type, dev.index)
    if key not in _scratch:
        # forward weight stream (rewritten by every convolution), unused slot, zero bias, the weight gradient's partial sums
        _scratch[key] = (torch.empty(_STREAM_BYTES, dtype=torch.uint8, device=dev), None,
                         torch.zeros(256, dtype=torch.float32, device=dev), torch.empty(16 * 9 * 256 * 256, dtype=torch.float32, device=dev))
    return _scratch[key]


WGRAD_KERNEL = True          # the weight gradient on the matrix cores too (f16 operands only); False: torch.nn.grad.conv2d_weight (MIOpen)


class SplitConv3x3(torch.autograd.Function):
    """forward packs BOTH weight streams in one launch when the input wants a gradient (the backward-data stream lives in a buffer of its own until backward: 2.4 MB per
    convolution in flight) and zeroes the two maximum slots with it; backward then is the convolution, the weight-gradient kernel and its reduction — no pack, no fill."""

    @staticmethod
    def forward(ctx, x, w):
        f16 = int(bool(OPERANDS_F16))
        use_wg = bool(WGRAD_KERNEL and f16)
        x, wd = x.contiguous(), w.detach().contiguous()
        dev = x.device
        amax = torch.empty(2, dtype=torch.int32, device=dev) if use_wg else None
        bwd_buf = torch.empty(_STREAM_BYTES, dtype=torch.uint8, device=dev) if ctx.needs_input_grad[0] else None
        fwd_buf, _, zero, _ = _bufs(dev)
        y = torch.empty_like(x)
        N.check(N.lib().sz_nn_conv3x3_train_fwd(x.data_ptr(), wd.data_ptr(), f16, fwd_buf.data_ptr(), bwd_buf.data_ptr() if bwd_buf is not None else None, zero.data_ptr(),
                                                y.data_ptr(), x.shape[0], amax.data_ptr() if use_wg else None, torch.cuda.current_stream(dev).cuda_stream), "sz_nn_conv3x3_train_fwd")
        ctx.save_for_backward(x, w)
        ctx.amax, ctx.bwd_buf, ctx.f16, ctx.use_wg = amax, bwd_buf, f16, use_wg
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        dev = gy.device
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = torch.empty_like(gy) if want_gx else None
        gw = None
        own_gw = want_gw and ctx.use_wg
        if own_gw:
            gw = torch.empty(w.shape, dtype=torch.float32, device=dev)
            if not want_gx:                                         # no backward-data kernel leaves max|gy| behind: take it with torch
                ctx.amax[1:].copy_(gy.detach().abs().amax().reshape(1).view(torch.int32))
        if want_gx or own_gw:
            _, _, zero, part = _bufs(dev)
            N.check(N.lib().sz_nn_conv3x3_train_bwd(gy.data_ptr(), x.data_ptr(), ctx.bwd_buf.data_ptr() if want_gx else None, zero.data_ptr(), gx.data_ptr() if want_gx else None,
                                                    ctx.amax.data_ptr() if ctx.use_wg else None, part.data_ptr(), gw.data_ptr() if own_gw else None, gy.shape[0], ctx.f16,
                                                    torch.cuda.current_stream(dev).cuda_stream), "sz_nn_conv3x3_train_bwd")
        if want_gw and not own_gw:
            gw = torch.nn.grad.conv2d_weight(x, w.shape, gy, padding=1)
        return gx, gw


FUSED_BN = True              # train-mode BatchNorm + skip connection + ReLU of the tower's blocks as one launch per direction (csrc/sz_train.hip); False: torch's five to six


class BNAct(torch.autograd.Function):
    """y = relu(batch_norm_train(x; bn) [+ residual]) — network.py:62-83 in train mode — in one launch forward and one backward (sz_bn_act_train_fwd / _bwd): batch
    statistics with torch's formulas (two-pass variance, eps inside the root, unbiased variance into running_var), running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        dev = x.device
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        stats = torch.empty(2, x.shape[1], dtype=torch.float32, device=dev)          # mean, 1 / sqrt(var + eps) of the batch
        N.check(N.lib().sz_bn_act_train_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr() if running_mean is not None else None,
                                            running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps), res.data_ptr() if res is not None else None,
                                            y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), x.shape[0], x.shape[1], torch.cuda.current_stream(dev).cuda_stream),
                "sz_bn_act_train_fwd")
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, stats = ctx.saved_tensors
        gy = gy.contiguous()
        dev = gy.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgb = torch.empty(2, x.shape[1], dtype=torch.float32, device=dev)
        N.check(N.lib().sz_bn_act_train_bwd(gy.data_ptr(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), dx.data_ptr(),
                                            dres.data_ptr() if dres is not None else None, dgb[0].data_ptr(), dgb[1].data_ptr(), x.shape[0], x.shape[1],
                                            torch.cuda.current_stream(dev).cuda_stream), "sz_bn_act_train_bwd")
        return dx, dgb[0], dgb[1], dres, None, None, None, None


def _bn_act(bn, x, residual=None):
    """relu(bn(x) [+ residual]) through BNAct when `bn` is a train-mode BatchNorm2d with affine parameters and a momentum on fp32 CUDA data, else through torch"""
    if (FUSED_BN and bn.training and bn.affine and bn.momentum is not None and bn.track_running_stats and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2:] == (8, 8) and bn.weight.dtype == torch.float32):
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        return BNAct.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    y = bn(x)
    return torch.relu(y if residual is None else y + residual)


class ConvBNAct(torch.autograd.Function):
    """y = relu(bn(conv3x3(x, w)) [+ residual]) — one half of a BasicBlock (network.py:62-83) — as ONE autograd node: forward = weight pack + convolution + fused
    BatchNorm/skip/ReLU (three launches, two C calls), backward = BatchNorm backward + backward-data convolution + weight gradient + reduction.  Half as many Python
    autograd nodes as SplitConv3x3 followed by BNAct: the host's time per node is what bounds the eager step once the kernels are this short."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, running_mean, running_var, momentum, eps):
        f16 = int(bool(OPERANDS_F16))
        use_wg = bool(WGRAD_KERNEL and f16)
        x, wd = x.contiguous(), w.detach().contiguous()
        dev = x.device
        st = torch.cuda.current_stream(dev).cuda_stream
        amax = torch.empty(2, dtype=torch.int32, device=dev) if use_wg else None
        bwd_buf = torch.empty(_STREAM_BYTES, dtype=torch.uint8, device=dev) if ctx.needs_input_grad[0] else None
        fwd_buf, _, zero, _ = _bufs(dev)
        t = torch.empty_like(x)                                      # the convolution's output = BatchNorm's input
        N.check(N.lib().sz_nn_conv3x3_train_fwd(x.data_ptr(), wd.data_ptr(), f16, fwd_buf.data_ptr(), bwd_buf.data_ptr() if bwd_buf is not None else None, zero.data_ptr(),
                                                t.data_ptr(), x.shape[0], amax.data_ptr() if use_wg else None, st), "sz_nn_conv3x3_train_fwd")
        res = residual.contiguous() if residual is not None else None
        y = torch.empty_like(x)
        stats = torch.empty(2, x.shape[1], dtype=torch.float32, device=dev)
        N.check(N.lib().sz_bn_act_train_fwd(t.data_ptr(), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), float(momentum), float(eps),
                                            res.data_ptr() if res is not None else None, y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), x.shape[0], x.shape[1], st),
                "sz_bn_act_train_fwd")
        ctx.save_for_backward(x, w, t, y, gamma, stats)
        ctx.amax, ctx.bwd_buf, ctx.f16, ctx.use_wg, ctx.has_res = amax, bwd_buf, f16, use_wg, res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, t, y, gamma, stats = ctx.saved_tensors
        gy = gy.contiguous()
        dev = gy.device
        st = torch.cuda.current_stream(dev).cuda_stream
        gt = torch.empty_like(t)
        dres = torch.empty_like(t) if ctx.has_res else None
        dgb = torch.empty(2, t.shape[1], dtype=torch.float32, device=dev)
        N.check(N.lib().sz_bn_act_train_bwd(gy.data_ptr(), t.data_ptr(), y.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), gt.data_ptr(),
                                            dres.data_ptr() if dres is not None else None, dgb[0].data_ptr(), dgb[1].data_ptr(), t.shape[0], t.shape[1], st), "sz_bn_act_train_bwd")
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = torch.empty_like(gt) if want_gx else None
        gw = None
        own_gw = want_gw and ctx.use_wg
        if own_gw:
            gw = torch.empty(w.shape, dtype=torch.float32, device=dev)
            if not want_gx:
                ctx.amax[1:].copy_(gt.abs().amax().reshape(1).view(torch.int32))
        if want_gx or own_gw:
            _, _, zero, part = _bufs(dev)
            N.check(N.lib().sz_nn_conv3x3_train_bwd(gt.data_ptr(), x.data_ptr(), ctx.bwd_buf.data_ptr() if want_gx else None, zero.data_ptr(), gx.data_ptr() if want_gx else None,
                                                    ctx.amax.data_ptr() if ctx.use_wg else None, part.data_ptr(), gw.data_ptr() if own_gw else None, gt.shape[0], ctx.f16, st),
                    "sz_nn_conv3x3_train_bwd")
        if want_gw and not own_gw:
            gw = torch.nn.grad.conv2d_weight(x, w.shape, gt, padding=1)
        return gx, gw, dgb[0], dgb[1], dres, None, None, None, None


def _conv_bn_act(conv, bn, x, residual=None):
    """relu(bn(conv(x)) [+ residual]): one ConvBNAct node when both layers qualify (train-mode affine BatchNorm2d with momentum, 3x3 256 -> 256 convolution, fp32 CUDA
    boards), else the layers one by one (each through its own fast path where that applies)"""
    if (FUSED_BN and bn.training and bn.affine and bn.momentum is not None and bn.track_running_stats and hasattr(conv, "_sz_orig_forward") and x.is_cuda
            and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1:] == (256, 8, 8) and conv.weight.dtype == torch.float32 and bn.weight.dtype == torch.float32):
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        return ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    return _bn_act(bn, conv(x), residual)


def _fused_block_forward(self, x):
    """ResidualBlock.forward (network.py:62-83): two convolution + BatchNorm (+ skip) + ReLU nodes"""
    y = _conv_bn_act(self.conv1, self.bn1, x)
    return _conv_bn_act(self.conv2, self.bn2, y, x)


def _eligible(m):
    return (isinstance(m, torch.nn.Conv2d) and m.in_channels == 256 and m.out_channels == 256 and m.kernel_size == (3, 3) and m.padding == (1, 1)
            and m.stride == (1, 1) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None)


def enable_split_convs(model):
    """every 3x3 256->256 convolution of `model` runs SplitConv3x3 on fp32 cuda inputs of 8x8 boards (anything else falls through to torch)"""
    n = 0
    for m in model.modules():
        if _eligible(m) and not hasattr(m, "_sz_orig_forward"):
            m._sz_orig_forward = m.forward

            def fwd(x, m=m):
                if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1:] == (256, 8, 8) and m.weight.dtype == torch.float32:
                    return SplitConv3x3.apply(x, m.weight)
                return m._sz_orig_forward(x)
            m.forward = fwd
            n += 1
    from .network import ResidualBlock
    import types
    for m in model.modules():
        if isinstance(m, ResidualBlock) and not hasattr(m, "_sz_orig_block_forward"):
            m._sz_orig_block_forward = m.forward
            m.forward = types.MethodType(_fused_block_forward, m)
    return n


def disable_split_convs(model):
    for m in model.modules():
        if hasattr(m, "_sz_orig_forward"):
            m.forward = m._sz_orig_forward
            del m._sz_orig_forward
        if hasattr(m, "_sz_orig_block_forward"):
            m.forward = m._sz_orig_block_forward
            del m._sz_orig_block_forward


@contextlib.contextmanager
def split_convs(model):
    enable_split_convs(model)
    try:
        yield model
    finally:
        disable_split_convs(model)
