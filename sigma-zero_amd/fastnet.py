"""Inference form of policyNN (network.py) on the hand-written CDNA4 MFMA kernels (csrc/sz_nn.hip).

BatchNorm (eval mode) is folded into the convolution weights and a per-channel bias.  A forward is two-three launches:
`sz_nn_tower_bf16` (stem + all BasicBlocks in one persistent kernel, activations resident in LDS) and `sz_nn_heads_bf16`
(conv_p1 -> conv_p2 -> softmax and conv_v1 -> value MLP from one read of the tower output).  The per-layer entry points
(`sz_nn_conv_bf16`, `sz_nn_block_bf16`, separate head kernels, the torch heads) stay selectable as cross-checks:
`persistent_max_boards`, `fuse_blocks`, `fused_heads`, `native_heads`, `mfma16`.
Input: the engine's NHWC planes [B, 64, 128] bf16 (119 real channels, the rest zero), or the same image bit-packed
([B, 1024] uint8, engine planes_dtype="bits128"), which the stem expands while staging its LDS tile.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


def _fold_bn(conv_w, bn, conv_b=None):
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    w = conv_w.detach().double() * scale.view(-1, 1, 1, 1)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    if conv_b is not None:
        b = b + conv_b.detach().double() * scale
    return w.float(), b.float()


def _pack(w, cin_padded, ksize, device, w16=False, f16=False):
    """[256, cin, k, k] f32 -> MFMA A-fragment order (uint16 bf16 bits; f16 bits with f16=True) on `device`.
    w16: fragment order of the 16x16x32 kernels (sz_nn_pack_weights16) instead of the 32x32x16 ones."""
    w = w.contiguous().cpu().float().numpy()
    co, cin = w.shape[0], w.shape[1]
    assert co == 256 and (w16 or not f16)
    out = np.zeros(ksize * ksize * cin_padded * 256, dtype=np.uint16)
    fn = (N.lib().sz_nn_pack_weights16_f16 if f16 else N.lib().sz_nn_pack_weights16) if w16 else N.lib().sz_nn_pack_weights
    N.check(fn(w.ctypes.data_as(C.c_void_p), cin, cin_padded, ksize, out.ctypes.data_as(C.c_void_p)), "sz_nn_pack_weights")
    return torch.from_numpy(out.view(np.int16)).to(device)


class FastPolicyNet:
    """operands: "bf16" (default) or "fp16" — the element type of the MFMA operands (weights and stored activations; accumulation, bias,
    residual add and the value MLP are f32 either way).  fp16 keeps 11 bits of mantissa instead of 8 at the same speed; it needs the
    16x16x32 kernels with the persistent tower and the fused heads (the defaults)."""

    def __init__(self, model, device=None, mfma16=True, operands="bf16"):
        model = model.eval()
        assert operands in ("bf16", "fp16") and (operands == "bf16" or mfma16)
        self.operands = operands
        self.f16 = operands == "fp16"
        self.eflag = N.SZ_NN_F16 if self.f16 else 0
        self.force_wgb = 0                    # tests: N.SZ_NN_TOWER_WGB1 / _WGB2 force the one- / two-board workgroup form of the persistent tower
        if device is None:                   # the GPU the fp32 module lives on, else this process's current device (never a silent cuda:0)
            pdev = next(model.parameters()).device
            device = pdev if pdev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        dev = self.device
        self.w16 = bool(mfma16)              # 16x16x32 MFMA kernels (higher sustained clock) vs 32x32x16
        self.flag = N.SZ_NN_W16 if self.w16 else 0
        pack_fn = _pack
        def _pack_w(w, cin_padded, ksize, device):
            return pack_fn(w, cin_padded, ksize, device, w16=self.w16, f16=self.f16)
        self.layers = []          # (packed_w, bias, cin, ksize)
        w, b = _fold_bn(model.conv1.weight, model.norm_layer)
        self.stem = (_pack_w(w, 128, 3, dev), b.to(dev).contiguous())
        self.blocks = []
        for blk in model.resnet_blocks:
            w1, b1 = _fold_bn(blk.conv1.weight, blk.bn1)
            w2, b2 = _fold_bn(blk.conv2.weight, blk.bn2)
            self.blocks.append((_pack_w(w1, 256, 3, dev), b1.to(dev).contiguous(), _pack_w(w2, 256, 3, dev), b2.to(dev).contiguous()))
        wp, bp = _fold_bn(model.conv_p1.weight, model.p_norm1)
        self.p1 = (_pack_w(wp, 256, 1, dev), bp.to(dev).contiguous())
        # heads (tiny): policy 1x1 256->73 (+bias), value 1x1 256->1 + BN + ReLU + MLP
        self.wp2 = model.conv_p2.weight.detach().view(73, 256).t().contiguous().to(dev).to(torch.bfloat16)      # [256,73]
        self.bp2 = model.conv_p2.bias.detach().float().to(dev)
        wv, bv = _fold_bn(model.conv_v1.weight, model.v_norm)
        self.wv = wv.view(1, 256).t().contiguous().to(dev).to(torch.bfloat16)                                    # [256,1]
        self.bv = bv.to(dev)
        self.fc1_w = model.fc_v1.weight.detach().float().t().contiguous().to(dev)
        self.fc1_b = model.fc_v1.bias.detach().float().to(dev)
        self.fc2_w = model.fc_v2.weight.detach().float().t().contiguous().to(dev)
        self.fc2_b = model.fc_v2.bias.detach().float().to(dev)
        # native heads (csrc/sz_nn.hip k_policy_head / k_value_head): packed conv_p2, folded conv_v1, fc weights in f32
        wp2 = model.conv_p2.weight.detach().view(73, 256).contiguous().cpu().float().numpy()
        packed = np.zeros(8 * 5 * 64 * 8, dtype=np.uint16)
        N.check((N.lib().sz_nn_pack_head16_f16 if self.f16 else N.lib().sz_nn_pack_head16)(wp2.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)), "sz_nn_pack_head16")
        self.wp2_packed = torch.from_numpy(packed.view(np.int16)).to(dev)
        self.wv_f32 = wv.view(256).contiguous().to(dev)
        self.bv_f = float(bv.view(-1)[0])
        self.fc2_w_vec = self.fc2_w.view(256).contiguous()
        self.fc2_b_f = float(self.fc2_b.view(-1)[0])
        self.native_heads = True
        self.fused_heads = self.w16          # one pass over the tower output for both heads (sz_nn_heads_bf16)
        # whole-tower persistent kernel (sz_nn_tower_bf16): host arrays of device pointers, 16x16x32 weight order only
        self.persistent_tower = self.w16
        # measured (tools/tower_vs_blocks.py): one launch for the whole tower beats 20 per-layer launches at every batch size since the
        # K loop interleaves its loads into the MFMA gaps (B=512: 0.96 vs 1.23 ms, B=4096: 7.48 vs 7.67 ms); lower this to force per-block launches
        self.persistent_max_boards = 1 << 30
        if self.w16:
            ws = [self.stem[0]] + [w for blk in self.blocks for w in (blk[0], blk[2])]
            bs = [self.stem[1]] + [b for blk in self.blocks for b in (blk[1], blk[3])]
            self._tower_w = (C.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
            self._tower_b = (C.c_void_p * len(bs))(*[t.data_ptr() for t in bs])
        self._bufs, self._full, self._cap = {}, None, 0
        self.fuse_blocks = True      # one launch per BasicBlock (sz_nn_block_bf16); False = two sz_nn_conv_bf16 launches
        self.timing = None          # optional list: (start, end) HIP event pairs around every 3x3 C_in=256 conv launch

    def parameters(self):
        return iter([self.wp2])        # dtype probe used by callers (bf16)

    def to(self, *a, **k):
        return self

    def _buffers(self, B):
        """activation / output buffers for a batch of B boards: views of ONE capacity-sized set that only ever grows (a self-play
        run whose live-board count shrinks ply by ply must not leave a 400 MB allocation behind per distinct batch size)"""
        if B > self._cap:
            dev = self.device
            self._full = [torch.empty(B, 64, 256, dtype=torch.bfloat16, device=dev) for _ in range(3)] + \
                         [torch.empty(B, 4672, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev),
                          torch.empty(B, 64, dtype=torch.float32, device=dev)]
            self._cap = B
            self._bufs = {}
        if B not in self._bufs:
            if len(self._bufs) > 64:
                self._bufs = {}
            self._bufs[B] = [t[:B] for t in self._full]
        return self._bufs[B]

    def _conv(self, x, w, b, res, out, B, cin, ksize, relu=1):
        ev = None
        if self.timing is not None and cin == 256 and ksize == 3 and not self.fuse_blocks:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._launch(x, w, b, res, out, B, cin, ksize, relu)
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def _launch(self, x, w, b, res, out, B, cin, ksize, relu):
        N.check(N.lib().sz_nn_conv_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                        C.c_void_p(res.data_ptr()) if res is not None else None, C.c_void_p(out.data_ptr()),
                                        B, cin, ksize, relu | self.flag, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sz_nn_conv_bf16")

    @torch.no_grad()
    def tower(self, planes):
        """planes [B,64,128] bf16 NHWC -> tower output [B,64,256] bf16 NHWC"""
        B = planes.shape[0]
        a, t, c = self._buffers(B)[:3]
        in_bits = N.SZ_NN_IN_BITS if planes.dtype == torch.uint8 else 0      # engine planes_dtype="bits128" (1 KiB per board)
        if self.f16 and planes.dtype == torch.bfloat16:
            planes = planes.to(torch.float16)            # the engine's "nhwc128" image is bf16; the f16 kernels read f16 (the default bit-packed image needs no conversion)
        if in_bits and not self.w16:
            raise ValueError("bit-packed planes need the 16x16x32 kernels (mfma16=True)")
        if self.persistent_tower and (B <= self.persistent_max_boards):
            ev = None
            if self.timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            N.check(N.lib().sz_nn_tower_bf16(C.c_void_p(planes.data_ptr()), self._tower_w, self._tower_b, len(self.blocks), C.c_void_p(a.data_ptr()), B, in_bits | self.eflag | self.force_wgb,
                                             C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sz_nn_tower_bf16")
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)
            return a, t
        if self.f16:
            raise ValueError("fp16 operands need the persistent tower")
        self._conv(planes, self.stem[0], self.stem[1], None, a, B, 128, 3, relu=1 | in_bits)
        for (w1, b1, w2, b2) in self.blocks:
            if self.fuse_blocks:
                self._block(a, w1, b1, w2, b2, c, B)
            else:
                self._conv(a, w1, b1, None, t, B, 256, 3)
                self._conv(t, w2, b2, a, c, B, 256, 3)
            a, c = c, a
        return a, t

    def _block(self, x, w1, b1, w2, b2, out, B):
        ev = None
        if self.timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        N.check(N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w1.data_ptr()), C.c_void_p(b1.data_ptr()), C.c_void_p(w2.data_ptr()),
                                         C.c_void_p(b2.data_ptr()), C.c_void_p(out.data_ptr()), B, self.flag,
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sz_nn_block_bf16")
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    @torch.no_grad()
    def __call__(self, planes, inference=True):
        if torch.cuda.current_device() != self.device.index:
            # the launches below go to this network's GPU even when the caller's current device is another one (a null stream handle
            # carries no device: the C ABI can only guard calls made on a real stream)
            with torch.cuda.device(self.device):
                return self._forward(planes, inference)
        return self._forward(planes, inference)

    def _forward(self, planes, inference=True):
        B = planes.shape[0]
        x, scratch = self.tower(planes)
        if self.native_heads:
            policy, value, v1 = self._buffers(B)[3:6]
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if self.fused_heads and self.w16:
                P = lambda t: C.c_void_p(t.data_ptr())
                N.check(N.lib().sz_nn_heads_bf16(P(x), P(self.p1[0]), P(self.p1[1]), P(self.wp2_packed), P(self.bp2), P(self.wv_f32), self.bv_f,
                                                 P(self.fc1_w), P(self.fc1_b), P(self.fc2_w_vec), self.fc2_b_f, P(policy), P(value), P(v1),
                                                 B, int(bool(inference)) | self.eflag, st), "sz_nn_heads_bf16")
                return policy, value.view(B, 1)
            if self.f16:
                raise ValueError("fp16 operands need the fused heads")
            self._conv(x, self.p1[0], self.p1[1], None, scratch, B, 256, 1)
            N.check(N.lib().sz_nn_policy_head_bf16(C.c_void_p(scratch.data_ptr()), C.c_void_p(self.wp2_packed.data_ptr()), C.c_void_p(self.bp2.data_ptr()),
                                                   C.c_void_p(policy.data_ptr()), B, int(bool(inference)), st), "sz_nn_policy_head_bf16")
            N.check(N.lib().sz_nn_value_head_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(self.wv_f32.data_ptr()), self.bv_f, C.c_void_p(self.fc1_w.data_ptr()),
                                                  C.c_void_p(self.fc1_b.data_ptr()), C.c_void_p(self.fc2_w_vec.data_ptr()), self.fc2_b_f,
                                                  C.c_void_p(value.data_ptr()), B, st), "sz_nn_value_head_bf16")
            return policy, value.view(B, 1)
        self._conv(x, self.p1[0], self.p1[1], None, scratch, B, 256, 1)
        logits = torch.matmul(scratch.view(B * 64, 256), self.wp2).float().view(B, 64, 73) + self.bp2        # [B,pos,plane]
        logits = logits.transpose(1, 2).reshape(B, 73 * 64)                                                  # flatten of [73,8,8]
        policy = torch.softmax(logits, dim=1) if inference else logits
        v = torch.relu(torch.matmul(x.view(B * 64, 256), self.wv).float().view(B, 64) + self.bv)
        v = torch.relu(v @ self.fc1_w + self.fc1_b)
        value = torch.tanh(v @ self.fc2_w + self.fc2_b)
        return policy, value


def planes_nchw_to_nhwc128(planes):
    """[B,119,8,8] (any float dtype) -> [B,64,128] bf16, for tests"""
    B = planes.shape[0]
    out = torch.zeros(B, 64, 128, dtype=torch.bfloat16, device=planes.device)
    out[:, :, :119] = planes.reshape(B, 119, 64).transpose(1, 2).to(torch.bfloat16)
    return out


class SplitPolicyNet:
    """policyNN inference at the REFERENCE's precision class on the matrix cores (csrc/sz_nn_split.hip k_tower_split).

    network.py is fp32 end to end.  FastPolicyNet's bf16 operands keep 8 bits of mantissa (search-level effect measured in
    tests/test_gpu_train_and_precision.py: single visits move).  Here every tower operand is carried as two bf16 numbers, x = hi + lo
    (16 bits), every product as three MFMAs (hi*hi + lo*hi + hi*lo) with f32 accumulation, the residual and BatchNorm-folded bias in f32;
    both heads run inside the same launch on the tile while it is still in LDS (conv_p1 / conv_p2 on hi + lo operands, conv_v1 and the value
    MLP in f32; `fused_heads=False`: fp32 GEMMs through torch, `module_heads=True`: the module's own heads — cross-checks).  About 2.6x the
    time of the bf16 tower, 10x faster than the fp32 torch/MIOpen forward, logits within ~1e-5 relative of it; a board's outputs do not depend
    on the batch it is evaluated in (bit for bit).  Input: the engine's bit-packed planes ("bits128") or the bf16 NHWC image ("nhwc128").
    The returned tensors are views of buffers that the next call reuses."""

    def __init__(self, model, device=None, operands="bf16"):
        """operands: "bf16" (hi + lo bf16: 16 bits of mantissa, logits ~6e-6 from fp64 — reproduces the fp32 network's search results in every test) or
        "fp16" (hi + lo f16 with the weights scaled by 2^10: 22 bits, logits ~5e-7 = fp32 itself; same MFMA count, the chip holds a ~5 % lower clock)."""
        model = model.eval()
        assert operands in ("bf16", "fp16")
        self.operands, self.f16 = operands, operands == "fp16"
        if device is None:
            pdev = next(model.parameters()).device
            device = pdev if pdev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        dev = self.device
        self.model = model.to(dev).float()               # heads in fp32 (policy_head / value_head of the module)
        self.w16 = True                                   # fragment order of the 16x16x32 kernels; play_games() picks bit-packed planes from this
        convs = [_fold_bn(model.conv1.weight, model.norm_layer)]
        for blk in model.resnet_blocks:
            convs.append(_fold_bn(blk.conv1.weight, blk.bn1))
            convs.append(_fold_bn(blk.conv2.weight, blk.bn2))
        self.n_blocks = len(model.resnet_blocks)
        # the whole tower's weights as ONE device buffer in k-step order {w_hi fragments, w_lo fragments} (csrc/sz_nn_split.hip), biases [n_convs, 256]
        L = N.lib()
        stream = np.zeros(int(L.sz_nn_split_stream_elems(self.n_blocks)), dtype=np.uint16)
        wp1, bp1 = _fold_bn(model.conv_p1.weight, model.p_norm1)
        for k, (w, b) in enumerate(convs + [(wp1, bp1)]):                  # conv_p1 (1x1) rides behind the tower in the stream: the heads are fused onto the tile
            wk = w.contiguous().cpu().float().numpy()
            assert wk.shape[0] == 256 and wk.shape[2] == wk.shape[3] == (1 if k == len(convs) else 3)
            N.check((L.sz_nn_pack_split_stream_f16 if self.f16 else L.sz_nn_pack_split_stream)(wk.ctypes.data_as(C.c_void_p), wk.shape[1], wk.shape[2], k,
                                                                                             stream.ctypes.data_as(C.c_void_p)), "sz_nn_pack_split_stream")
        self._wstream = torch.from_numpy(stream.view(np.int16)).to(dev)
        self._bias = (torch.stack([b for _, b in convs] + [bp1]).float() * (1024.0 if self.f16 else 1.0)).contiguous().to(dev)    # f16 operands: weights and biases times 2^10
        self._eflag = N.SZ_NN_F16 if self.f16 else 0
        self.force_wgb = 0                                # tests: N.SZ_NN_SPLIT_WGB1 / _WGB2 force one- / two-board workgroups
        wp2 = model.conv_p2.weight.detach().view(73, 256).contiguous().cpu().float().numpy()
        p2 = np.zeros(8 * 2 * 5 * 64 * 8, dtype=np.uint16)
        N.check((L.sz_nn_pack_split_head_f16 if self.f16 else L.sz_nn_pack_split_head)(wp2.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p)), "sz_nn_pack_split_head")
        self._wp2 = torch.from_numpy(p2.view(np.int16)).to(dev)
        self.fused_heads = True                           # both heads inside the tower launch (sz_nn_forward_split); False: fp32 GEMM heads through torch (cross-check)
        self._hbuf, self._hcap = None, 0
        # heads as fp32 GEMMs on the NHWC activation (BatchNorm folded in double): [B*64,256] x [256,256] -> ReLU -> x [256,73]; value 256 -> 1 -> MLP
        self.h_wp1, self.h_bp1 = wp1.view(256, 256).t().contiguous().to(dev), bp1.to(dev)
        self.h_wp2, self.h_bp2 = model.conv_p2.weight.detach().float().view(73, 256).t().contiguous().to(dev), model.conv_p2.bias.detach().float().to(dev)
        wv, bv = _fold_bn(model.conv_v1.weight, model.v_norm)
        self.h_wv, self.h_bv = wv.view(1, 256).t().contiguous().to(dev), bv.to(dev)
        self._wv_vec, self._bv_f = wv.view(256).contiguous().to(dev), float(bv.view(-1)[0])
        self.h_fc1_w, self.h_fc1_b = model.fc_v1.weight.detach().float().t().contiguous().to(dev), model.fc_v1.bias.detach().float().to(dev)
        self.h_fc2_w, self.h_fc2_b = model.fc_v2.weight.detach().float().t().contiguous().to(dev), model.fc_v2.bias.detach().float().to(dev)
        self._fc2_vec, self._fc2_b_f = self.h_fc2_w.view(256).contiguous(), float(self.h_fc2_b.view(-1)[0])
        self.module_heads = False                         # True: run policy_head / value_head of the torch module itself (cross-check)
        self._out, self._cap = None, 0
        self._p = torch.zeros(1, dtype=torch.float32, device=dev)
        self.timing = None

    def parameters(self):
        return iter([self._p])

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def tower(self, planes):
        """planes [B,1024] uint8 (bit-packed) or [B,64,128] bf16 -> tower activation [B,64,256] f32 (NHWC)"""
        B = planes.shape[0]
        if B > self._cap:
            self._out, self._cap = torch.empty(B, 64, 256, dtype=torch.float32, device=self.device), B
        out = self._out[:B]
        flags = (N.SZ_NN_IN_BITS if planes.dtype == torch.uint8 else 0) | self.force_wgb | self._eflag
        if self.f16 and planes.dtype == torch.bfloat16:
            planes = planes.to(torch.float16)
        ev = None
        if self.timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        N.check(N.lib().sz_nn_tower_split(C.c_void_p(planes.data_ptr()), C.c_void_p(self._wstream.data_ptr()), C.c_void_p(self._bias.data_ptr()), self.n_blocks,
                                          C.c_void_p(out.data_ptr()), B, flags, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sz_nn_tower_split")
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        return out

    @torch.no_grad()
    def __call__(self, planes, inference=True):
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                return self._forward(planes, inference)
        return self._forward(planes, inference)

    def _forward(self, planes, inference=True):
        B = planes.shape[0]
        if self.fused_heads and not self.module_heads:
            if B > self._hcap:
                self._hbuf = (torch.empty(B, 4672, dtype=torch.float32, device=self.device), torch.empty(B, dtype=torch.float32, device=self.device),
                              torch.empty(B, 64, dtype=torch.float32, device=self.device))
                self._hcap = B
            policy, value, v1 = (t[:B] for t in self._hbuf)
            P = lambda t: C.c_void_p(t.data_ptr())
            flags = (N.SZ_NN_IN_BITS if planes.dtype == torch.uint8 else 0) | self.force_wgb | self._eflag
            if self.f16 and planes.dtype == torch.bfloat16:
                planes = planes.to(torch.float16)
            ev = None
            if self.timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            N.check(N.lib().sz_nn_forward_split(P(planes), P(self._wstream), P(self._bias), self.n_blocks, P(self._wp2), P(self.h_bp2), P(self._wv_vec), self._bv_f,
                                                P(self.h_fc1_w), P(self.h_fc1_b), P(self._fc2_vec), self._fc2_b_f, P(policy), P(value), P(v1), None,
                                                B, int(bool(inference)), flags, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "sz_nn_forward_split")
            if ev is not None:
                ev[1].record()
                self.timing.append(ev)
            return policy, value.view(B, 1)
        out = self.tower(planes)
        if self.module_heads:
            x = out.view(B, 8, 8, 256).permute(0, 3, 1, 2)                   # logical NCHW, channels_last memory: no copy
            policy, value = self.model.policy_head(x), self.model.value_head(x)
        else:
            x2 = out.view(B * 64, 256)
            t = torch.relu(torch.addmm(self.h_bp1, x2, self.h_wp1))
            policy = torch.addmm(self.h_bp2, t, self.h_wp2).view(B, 64, 73).transpose(1, 2).reshape(B, 73 * 64)      # flatten of [73,8,8]
            v = torch.relu(torch.addmm(self.h_bv, x2, self.h_wv)).view(B, 64)
            value = torch.tanh(torch.addmm(self.h_fc2_b, torch.relu(torch.addmm(self.h_fc1_b, v, self.h_fc1_w)), self.h_fc2_w))
        if inference:
            policy = torch.softmax(policy, dim=1)
        return policy, value
