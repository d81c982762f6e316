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
MIOpen's fp32 kernels take 90-105 us (forward), 216 us (backward) per convolution at batch 128 — 78 % of an optimiser step; these take 41 us and 95 us.

    with split_convs(model):            # or enable_split_convs(model) / disable_split_convs(model)
        loss, mse, ce = train_rl.loss_fn(model, batch, device); loss.backward()

`train_rl.train` switches it on by default for an fp32 model on a GPU (`split_convs=False` / `--train-convs torch`: MIOpen).  Measured at batch 128
(tools/trainconv_probe.py, profiles/r03zze_trainconv_probe.txt, r03zzr_train_loop_same_box_ab_counters.txt): optimiser step 12.9 -> 7.1-7.6 ms; forward / input gradient / weight gradient of one convolution
5.0e-7 / 5.1e-7 / 2.7e-7 relative L2 from fp64 (torch fp32: 4.9e-7 / 5.1e-7 / 2.5e-7), also on inputs scaled by 1e3 or 1e-6; whole-network gradient 3.45e-3 from an fp64
step (MIOpen's fp32 step: 3.37e-3; the 39 train-mode BatchNorms amplify every rounding).  OPERANDS_F16 = False selects hi + lo bf16 operands for forward / backward-data
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
        ctx.bwd_buf = None
        return gx, gw


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
    return n


def disable_split_convs(model):
    for m in model.modules():
        if hasattr(m, "_sz_orig_forward"):
            m.forward = m._sz_orig_forward
            del m._sz_orig_forward


@contextlib.contextmanager
def split_convs(model):
    enable_split_convs(model)
    try:
        yield model
    finally:
        disable_split_convs(model)
