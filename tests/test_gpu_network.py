"""The hand-written MFMA convolution (csrc/sz_nn.hip) and FastPolicyNet against plain PyTorch fp32 references.
Floating point: tolerances are stated per test (bf16 storage, f32 accumulation)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import sigma_zero_amd as sz
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import FastPolicyNet, _pack, planes_nchw_to_nhwc128
from sigma_zero_amd.selfplay import SelfPlayEngine

pytestmark = pytest.mark.gpu


def _run_conv(x_nhwc, w, bias, res, cin_padded, ksize, relu, mode=0):
    B = x_nhwc.shape[0]
    out = torch.empty(B, 64, 256, dtype=torch.bfloat16, device="cuda")
    wp = _pack(w, cin_padded, ksize, "cuda", w16=bool(mode & N.SZ_NN_W16))
    N.check(N.lib().sz_nn_conv_bf16(C.c_void_p(x_nhwc.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bias.data_ptr()),
                                    C.c_void_p(res.data_ptr()) if res is not None else None, C.c_void_p(out.data_ptr()),
                                    B, cin_padded, ksize, int(relu) | mode, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("mode", [0, 16, N.SZ_NN_W16], ids=["wg2", "wg4", "mfma16"])
@pytest.mark.parametrize("B,cin,ksize,use_res,relu", [(4, 256, 3, False, True), (7, 256, 3, True, True), (3, 119, 3, False, True),
                                                       (5, 256, 1, False, True), (8, 256, 3, True, False), (130, 256, 3, True, True),
                                                       (1031, 256, 3, True, True)])
def test_conv_matches_torch_fp32(B, cin, ksize, use_res, relu, mode):
    g = torch.Generator(device="cuda").manual_seed(B * 100 + cin + ksize)
    cin_padded = 128 if cin == 119 else cin
    x = torch.randn(B, cin, 8, 8, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(256, cin, ksize, ksize, generator=g, device="cuda") * (1.0 / (cin * ksize * ksize) ** 0.5))
    bias = torch.randn(256, generator=g, device="cuda")
    res = torch.randn(B, 256, 8, 8, generator=g, device="cuda").to(torch.bfloat16) if use_res else None
    x_nhwc = torch.zeros(B, 64, cin_padded, dtype=torch.bfloat16, device="cuda")
    x_nhwc[:, :, :cin] = x.reshape(B, cin, 64).transpose(1, 2)
    res_nhwc = res.reshape(B, 256, 64).transpose(1, 2).contiguous() if use_res else None
    out = _run_conv(x_nhwc, w, bias, res_nhwc, cin_padded, ksize, relu, mode)
    # reference: same bf16-rounded inputs and weights, fp32 math
    mid = F.conv2d(x.float(), w.to(torch.bfloat16).float(), bias, padding=ksize // 2)
    ref = mid + res.float() if use_res else mid
    if relu:
        ref = torch.relu(ref)
    got = out.float().transpose(1, 2).reshape(B, 256, 8, 8)
    # tolerance: bf16 rounding (2^-8 relative) of the stored output and — on residual layers — of the conv+bias value
    # before the residual add (like torch's bf16 graph), plus f32 accumulation-order noise over K <= 2304
    tol = 2 ** -7 * ref.abs() + (2 ** -7 * mid.abs() if use_res else 0) + 2e-3
    err = (got - ref).abs()
    assert bool((err <= tol).all()), float((err - tol).max())


@pytest.mark.parametrize("w16", [False, True], ids=["mfma32", "mfma16"])
@pytest.mark.parametrize("B", [2, 5, 131])
def test_fused_block_matches_torch_fp32(B, w16):
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.randn(B, 256, 8, 8, generator=g, device="cuda").to(torch.bfloat16)
    w1 = torch.randn(256, 256, 3, 3, generator=g, device="cuda") * (1.0 / 2304 ** 0.5)
    w2 = torch.randn(256, 256, 3, 3, generator=g, device="cuda") * (1.0 / 2304 ** 0.5)
    b1 = torch.randn(256, generator=g, device="cuda") * 0.3
    b2 = torch.randn(256, generator=g, device="cuda") * 0.3
    x_nhwc = x.reshape(B, 256, 64).transpose(1, 2).contiguous()
    out = torch.empty(B, 64, 256, dtype=torch.bfloat16, device="cuda")
    wp1, wp2 = _pack(w1, 256, 3, "cuda", w16=w16), _pack(w2, 256, 3, "cuda", w16=w16)   # keep the packed tensors alive across the launch
    N.check(N.lib().sz_nn_block_bf16(C.c_void_p(x_nhwc.data_ptr()), C.c_void_p(wp1.data_ptr()), C.c_void_p(b1.data_ptr()),
                                     C.c_void_p(wp2.data_ptr()), C.c_void_p(b2.data_ptr()), C.c_void_p(out.data_ptr()), B, N.SZ_NN_W16 if w16 else 0,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    t = torch.relu(F.conv2d(x.float(), w1.to(torch.bfloat16).float(), b1, padding=1))
    mid = F.conv2d(t.to(torch.bfloat16).float(), w2.to(torch.bfloat16).float(), b2, padding=1)     # t is stored as bf16 (in LDS)
    ref = torch.relu(mid + x.float())
    got = out.float().transpose(1, 2).reshape(B, 256, 8, 8)
    # bf16 roundings: t (propagated through conv2: ~2^-8 * |t| * sqrt(K) * |w| ~ 2^-8), conv2+bias, output
    tol = 2 ** -7 * ref.abs() + 2 ** -7 * mid.abs() + 2e-2
    err = (got - ref).abs()
    assert bool((err <= tol).all()), float((err - tol).max())


@pytest.mark.parametrize("mfma16", [False, True], ids=["mfma32", "mfma16"])
def test_fast_network_matches_fp32_policynn(mfma16):
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    # give BatchNorm non-trivial running statistics so that the folding is exercised
    g = torch.Generator(device="cuda").manual_seed(1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g, device="cuda") * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g, device="cuda") + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g, device="cuda") + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g, device="cuda") * 0.1)
    fast = FastPolicyNet(net, mfma16=mfma16)
    if mfma16:
        # the persistent whole-tower kernel against the per-block kernels (same weights; the only numeric difference is
        # the residual add: f32 from LDS vs. after a bf16 rounding)
        xin = planes_nchw_to_nhwc128((torch.rand(9, 119, 8, 8, generator=g, device="cuda") < 0.12).float())
        assert fast.persistent_tower and xin.shape[0] <= fast.persistent_max_boards
        y_p = fast.tower(xin)[0].float().clone()
        fast.persistent_tower = False
        y_b = fast.tower(xin)[0].float().clone()
        fast.persistent_tower = True
        assert float((y_p - y_b).norm() / y_b.norm()) < 0.02
    x = (torch.rand(37, 119, 8, 8, generator=g, device="cuda") < 0.12).float()
    with torch.no_grad():
        p_ref, v_ref = net(x, inference=True)
        p, v = fast(planes_nchw_to_nhwc128(x), inference=True)
    # bf16 storage through 40 layers (8 mantissa bits, ~sqrt(60) roundings): logits agree to a few percent in
    # relative L2 norm, values to 5e-2 absolute
    assert p.shape == p_ref.shape and torch.allclose(p.sum(1), torch.ones(37, device="cuda"), atol=1e-4)
    lg, lg_ref = torch.log(p), torch.log(p_ref)
    lg, lg_ref = lg - lg.mean(1, keepdim=True), lg_ref - lg_ref.mean(1, keepdim=True)
    rel = float((lg - lg_ref).norm() / lg_ref.norm())
    assert rel < 0.04, rel
    assert float((v.reshape(-1) - v_ref.reshape(-1)).abs().max()) < 5e-2, float((v.reshape(-1) - v_ref.reshape(-1)).abs().max())
    # rank agreement of the top move on almost every board
    agree = (p.argmax(1) == p_ref.argmax(1)).float().mean()
    assert agree >= 0.8, float(agree)


def test_engine_nhwc_planes_equal_nchw_planes():
    eng_a = SelfPlayEngine(None, {"C": 2, "num_searches": 8}, 16, chess960=True, planes_dtype=torch.bfloat16)
    eng_b = SelfPlayEngine(None, {"C": 2, "num_searches": 8}, 16, chess960=True, planes_dtype="nhwc128")
    sch = list(range(100, 116))
    g = torch.Generator(device="cuda").manual_seed(0)
    for e in (eng_a, eng_b):
        e.new_games(sch)
        e.begin()
    for step in range(8):
        torch.cuda.synchronize()
        a = eng_a.planes.float()
        b = eng_b.planes.float()
        assert torch.equal(b[:, :, 119:], torch.zeros_like(b[:, :, 119:]))
        assert torch.equal(b[:, :, :119].transpose(1, 2).reshape(16, 119, 8, 8), a), "step %d" % step
        policy = torch.softmax(torch.randn(16, N.SZ_ACTIONS, generator=g, device="cuda"), 1).contiguous()
        value = torch.rand(16, generator=g, device="cuda") * 2 - 1
        eng_a.step(policy, value)
        eng_b.step(policy, value)
    eng_a.close(); eng_b.close()


def _pack_bits128(img):
    """[B,64,128] 0/1 image -> [B,1024] uint8 in the engine's SZ_PLANES_NHWC128_BITS layout (inverse of unpack_bits128)"""
    B = img.shape[0]
    bits = img.to(torch.uint8).view(B, 16, 4, 16, 8)                 # [q][psub][cq][k]
    byte = (bits << torch.arange(8, device=img.device, dtype=torch.uint8)).sum(-1).to(torch.uint8)      # [B,q,psub,cq]
    return byte.permute(0, 2, 3, 1).reshape(B, 1024).contiguous()     # [psub][cq][q]


def test_engine_bit_planes_equal_nhwc_planes():
    """SZ_PLANES_NHWC128_BITS (1 KiB per board) decodes to exactly the bf16 NHWC image, white and black to move, with history"""
    from sigma_zero_amd.selfplay import unpack_bits128
    eng_a = SelfPlayEngine(None, {"C": 2, "num_searches": 12}, 16, chess960=True, planes_dtype="nhwc128")
    eng_b = SelfPlayEngine(None, {"C": 2, "num_searches": 12}, 16, chess960=True, planes_dtype="bits128")
    sch = list(range(300, 316))
    g = torch.Generator(device="cuda").manual_seed(0)
    urng = np.random.RandomState(0)
    for e in (eng_a, eng_b):
        e.new_games(sch)
    for ply in range(3):
        for e in (eng_a, eng_b):
            e.begin()
        for step in range(12):
            torch.cuda.synchronize()
            assert eng_b.planes.dtype == torch.uint8 and eng_b.planes.shape == (16, 1024)
            assert torch.equal(unpack_bits128(eng_b.planes), eng_a.planes), "ply %d step %d" % (ply, step)
            assert torch.equal(_pack_bits128(eng_a.planes.float()), eng_b.planes)
            policy = torch.softmax(torch.randn(16, N.SZ_ACTIONS, generator=g, device="cuda"), 1).contiguous()
            value = torch.rand(16, generator=g, device="cuda") * 2 - 1
            eng_a.step(policy, value)
            eng_b.step(policy, value)
        u = urng.random_sample(16)
        for e in (eng_a, eng_b):
            e.play(u)
            e.fetch_ply()
    eng_a.close(); eng_b.close()


def test_fastnet_bit_planes_input_is_bit_identical():
    """the stem expands bit-packed planes to the same LDS tile: outputs equal the bf16-NHWC-input run bit for bit
    (per-block kernels and the persistent whole-tower kernel; ragged batch)"""
    torch.manual_seed(5)
    net = sz.policyNN({}).cuda().eval()
    fast = FastPolicyNet(net)
    g = torch.Generator(device="cuda").manual_seed(2)
    x = planes_nchw_to_nhwc128((torch.rand(37, 119, 8, 8, generator=g, device="cuda") < 0.15).float())
    xb = _pack_bits128(x.float())
    for persistent in (True, False):
        fast.persistent_tower = persistent
        p0, v0 = fast(x, inference=True); p0, v0 = p0.clone(), v0.clone()
        p1, v1 = fast(xb, inference=True)
        assert torch.equal(p0, p1) and torch.equal(v0, v1), persistent
    with pytest.raises(ValueError):
        FastPolicyNet(net, mfma16=False)(xb, inference=True)


def test_fused_heads_match_separate_head_kernels():
    """sz_nn_heads_bf16 (one pass over the tower output) against conv_p1 + policy head + value head launches: same bf16 intermediate and
    the same MFMA products, so logits are identical; softmax / conv_v1 sums differ only in f32 summation order (1e-5)."""
    torch.manual_seed(11)
    net = sz.policyNN({}).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(4)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g, device="cuda") * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g, device="cuda") + 0.5)
    net.conv_p2.bias.data.copy_(torch.randn(73, generator=g, device="cuda"))
    fast = FastPolicyNet(net)
    assert fast.fused_heads
    for B in (37, 2, 1):
        x = planes_nchw_to_nhwc128((torch.rand(B, 119, 8, 8, generator=g, device="cuda") < 0.15).float())
        for inference in (True, False):
            fast.fused_heads = False
            p0, v0 = fast(x, inference=inference); p0, v0 = p0.clone(), v0.clone()
            fast.fused_heads = True
            p1, v1 = fast(x, inference=inference)
            if inference:
                assert torch.allclose(p0, p1, rtol=2e-5, atol=1e-10), float((p0 - p1).abs().max())
                assert torch.allclose(p1.sum(1), torch.ones(B, device="cuda"), atol=1e-5)
            else:
                assert torch.equal(p0, p1)
            assert float((v0 - v1).abs().max()) < 1e-5, float((v0 - v1).abs().max())


def test_split_precision_network_with_f16_operands_is_as_close_to_fp64_as_fp32_is():
    """SplitPolicyNet(operands="fp16"): hi + lo f16 operands (22 bits of mantissa; weights and biases times 2^10) — against an fp64 copy of the module, next to the
    fp32 module itself.  Measured on MI355X: tower activation / centred logits within a factor 2 of fp32's own distance from fp64 (~5e-7), ten times closer than
    the hi + lo bf16 form; forms and batch composition do not change a bit."""
    from sigma_zero_amd.fastnet import SplitPolicyNet
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.05)
    net64 = sz.policyNN({}).cuda().eval().double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    s16, sb = SplitPolicyNet(net, operands="fp16"), SplitPolicyNet(net)
    g = torch.Generator(device="cuda").manual_seed(13)
    c = lambda t: t - t.mean(1, keepdim=True)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    for B in (1, 37, 300):
        x = (torch.rand(B, 119, 8, 8, generator=g, device="cuda") < 0.12).float()
        planes = planes_nchw_to_nhwc128(x)
        with torch.no_grad():
            p64, v64 = net64(x.double(), inference=False)
            p32, v32 = net(x, inference=False)
            p16, v16 = (t.clone() for t in s16(planes, inference=False))
            pb, vb = (t.clone() for t in sb(planes, inference=False))
            e32, e16, eb = rel(c(p32), c(p64)), rel(c(p16), c(p64)), rel(c(pb), c(p64))
            assert e16 < 3e-6 and e16 < 4 * e32 + 1e-7 and e16 < eb / 3, (e32, e16, eb)
            assert float((v16.double().view(-1) - v64.view(-1)).abs().max()) < 1e-6
            s16.force_wgb = N.SZ_NN_SPLIT_WGB1
            pa, va = (t.clone() for t in s16(planes, inference=True))
            s16.force_wgb = N.SZ_NN_SPLIT_WGB2
            pc, vc = (t.clone() for t in s16(planes, inference=True))
            s16.force_wgb = 0
            assert torch.equal(pa, pc) and torch.equal(va, vc) and torch.allclose(pa.sum(1), torch.ones(B, device="cuda"), atol=1e-5)
            y16 = s16.tower(planes).clone()
            y64 = net64.resnet_blocks(torch.relu(net64.norm_layer(net64.conv1(x.double()))))
            assert rel(y16.view(B, 8, 8, 256).permute(0, 3, 1, 2), y64) < 3e-6
    print("split f16x2: centred logits rel L2 vs fp64 %.2e (fp32 module %.2e, split bf16x2 %.2e)" % (e16, e32, eb))


def test_fp16_operand_network_matches_fp32_policynn():
    """FastPolicyNet(operands="fp16"): the persistent tower and the fused heads on f16 MFMA operands (11 bits of mantissa; accumulation, bias, residual add in
    f32) against the fp32 module.  Measured on MI355X: tower activation 7.3e-4, centred logits 6.4e-4 relative L2 from fp64 (bf16 operands: 5.8e-3 / 5.1e-3)."""
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.05)
    f16, b16 = FastPolicyNet(net, operands="fp16"), FastPolicyNet(net)
    g = torch.Generator(device="cuda").manual_seed(12)
    c = lambda t: t - t.mean(1, keepdim=True)
    for B in (1, 37, 600):
        x = (torch.rand(B, 119, 8, 8, generator=g, device="cuda") < 0.12).float()
        with torch.no_grad():
            p_ref, v_ref = net(x, inference=False)
            y_ref = net.resnet_blocks(torch.relu(net.norm_layer(net.conv1(x))))
            p, v = (t.clone() for t in f16(planes_nchw_to_nhwc128(x), inference=False))
            pb, vb = (t.clone() for t in b16(planes_nchw_to_nhwc128(x), inference=False))
            y = f16.tower(planes_nchw_to_nhwc128(x))[0].view(torch.float16).float().view(B, 8, 8, 256).permute(0, 3, 1, 2)
            pi, _ = (t.clone() for t in f16(planes_nchw_to_nhwc128(x), inference=True))
            pi_ref, _ = net(x, inference=True)
        e16, eb = float((c(p) - c(p_ref)).norm() / c(p_ref).norm()), float((c(pb) - c(p_ref)).norm() / c(p_ref).norm())
        assert float((y - y_ref).norm() / y_ref.norm()) < 3e-3 and e16 < 3e-3 and e16 < eb / 3, (e16, eb)       # and at least 3x closer than bf16 operands
        assert float((v.view(-1) - v_ref.view(-1)).abs().max()) < 5e-3
        assert torch.allclose(pi.sum(1), torch.ones(B, device="cuda"), atol=1e-5) and float((pi.log() - pi_ref.log()).abs().max()) < 5e-3
        # one board per workgroup (the default up to #CUs boards) and two give the same bits, for both operand types: a board's result does not depend on the batch
        with torch.no_grad():
            for fnet in (f16, b16):
                fnet.force_wgb = N.SZ_NN_TOWER_WGB1
                p1, v1 = (t.clone() for t in fnet(planes_nchw_to_nhwc128(x), inference=True))
                fnet.force_wgb = N.SZ_NN_TOWER_WGB2
                p2, v2 = (t.clone() for t in fnet(planes_nchw_to_nhwc128(x), inference=True))
                fnet.force_wgb = 0
                p0, v0 = (t.clone() for t in fnet(planes_nchw_to_nhwc128(x), inference=True))
                assert torch.equal(p1, p2) and torch.equal(v1, v2) and torch.equal(p0, p1) and torch.equal(v0, v1)


def test_last_round_in_the_one_board_form_gives_the_same_bits():
    """sz_nn_tower_bf16 above 2 x #CUs boards: a last round of at most #CUs boards is launched in the one-board form (600 boards on 256 CUs = 512 in two-board tiles + 88
    with a CU each).  Bit-packed engine planes; bf16, f16 and split-precision (k_tower_split with fused heads) networks: identical to the two-board form throughout, and to the same boards evaluated alone."""
    import random
    from sigma_zero_amd.selfplay import SelfPlayEngine
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    B = 2 * n_cu + n_cu // 3 + 3
    from sigma_zero_amd.fastnet import SplitPolicyNet
    for operands in ("bf16", "fp16", "split"):
        fast = SplitPolicyNet(net) if operands == "split" else FastPolicyNet(net, operands=operands)
        wgb2 = N.SZ_NN_SPLIT_WGB2 if operands == "split" else N.SZ_NN_TOWER_WGB2
        eng = SelfPlayEngine(fast, {"C": 2, "num_searches": 2}, B, chess960=True, planes_dtype="bits128")
        eng.new_games([random.Random(3).randrange(960) for _ in range(B)])
        eng.search(); eng.play(np.random.RandomState(0).random_sample(B)); eng.fetch_ply()
        eng.begin()
        planes = eng.planes.clone()
        with torch.no_grad():
            p0, v0 = (t.clone() for t in fast(planes, inference=True))
            fast.force_wgb = wgb2
            p2, v2 = (t.clone() for t in fast(planes, inference=True))
            fast.force_wgb = 0
            pt, vt = (t.clone() for t in fast(planes[2 * n_cu:].contiguous(), inference=True))
        assert torch.equal(p0, p2) and torch.equal(v0, v2)
        assert torch.equal(p0[2 * n_cu:], pt) and torch.equal(v0.reshape(-1)[2 * n_cu:], vt.reshape(-1))
        assert float(p0.sum(1).sub(1).abs().max()) < 1e-4
        eng.close()


def test_split_precision_network_matches_fp32_policynn():
    """SplitPolicyNet (k_tower_split: hi + lo bf16 operands, three MFMAs per product, f32 accumulation, f32 heads) against the fp32 module —
    the reference's precision class (network.py has no reduced precision anywhere).  Tolerances are ~5x what was measured on MI355X
    (tower activation 9e-6, centred logits 7e-6 relative L2 against fp64; the bf16 tower is at 5e-3)."""
    from sigma_zero_amd.fastnet import SplitPolicyNet
    torch.manual_seed(0)
    net = sz.policyNN({}).cuda().eval()
    with torch.no_grad():                                   # non-trivial BatchNorm statistics, as a trained network has
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.7, 1.3); m.bias.normal_(0, 0.05)
    split = SplitPolicyNet(net)
    g = torch.Generator(device="cuda").manual_seed(11)
    for B in (1, 37, 300, 1100):                            # fewer boards than CUs (one board per workgroup), odd counts, more boards than CUs (two per workgroup), more tiles than CUs (the persistent tile loop)
        x = (torch.rand(B, 119, 8, 8, generator=g, device="cuda") < 0.12).float()
        with torch.no_grad():
            p_ref, v_ref = net(x, inference=False)
            p, v = (t.clone() for t in split(planes_nchw_to_nhwc128(x), inference=False))      # the outputs are views of buffers the next call reuses
            y_ref = net.resnet_blocks(torch.relu(net.norm_layer(net.conv1(x))))
            y = split.tower(planes_nchw_to_nhwc128(x)).clone().view(B, 8, 8, 256).permute(0, 3, 1, 2)
            # a board's activation does not depend on the workgroup form (one or two boards per workgroup) nor on its neighbours: bit for bit
            split.force_wgb = N.SZ_NN_SPLIT_WGB1
            y1 = split.tower(planes_nchw_to_nhwc128(x)).clone()
            split.force_wgb = N.SZ_NN_SPLIT_WGB2
            y2 = split.tower(planes_nchw_to_nhwc128(x)).clone()
            y2r = split.tower(planes_nchw_to_nhwc128(x.flip(0))).clone().flip(0)
            split.force_wgb = 0
        assert torch.equal(y1, y2) and torch.equal(y2, y2r) and torch.equal(y1.view(B, 8, 8, 256).permute(0, 3, 1, 2), y)
        assert float((y - y_ref).norm() / y_ref.norm()) < 5e-5
        c = lambda t: t - t.mean(1, keepdim=True)
        assert float((c(p) - c(p_ref)).norm() / c(p_ref).norm()) < 5e-5
        assert float((v - v_ref).abs().max()) < 1e-5
        with torch.no_grad():
            pi, _ = split(planes_nchw_to_nhwc128(x), inference=True)
            pi_ref, _ = net(x, inference=True)
        assert float((pi.log() - pi_ref.log()).abs().max()) < 1e-4 and bool((pi.argmax(1) == pi_ref.argmax(1)).all())
    # the module's own heads and the fp32 GEMM heads (torch) on the split tower's output give the numbers of the fused heads
    split.module_heads = True
    with torch.no_grad():
        p2, v2 = (t.clone() for t in split(planes_nchw_to_nhwc128(x), inference=False))
    split.module_heads, split.fused_heads = False, False
    with torch.no_grad():
        p3, v3 = (t.clone() for t in split(planes_nchw_to_nhwc128(x), inference=False))
    split.fused_heads = True
    assert float((p2 - p3).abs().max()) < 1e-5 and float((v2 - v3).abs().max()) < 1e-6
    assert float((c(p) - c(p3)).norm() / c(p3).norm()) < 5e-5 and float((v - v3).abs().max()) < 1e-5
    # fused heads: a board's policy and value do not depend on the workgroup form, on the batch size or on its neighbours — bit for bit
    with torch.no_grad():
        planes = planes_nchw_to_nhwc128(x)
        outs = []
        for inference in (True, False):
            split.force_wgb = N.SZ_NN_SPLIT_WGB1
            pa, va = (t.clone() for t in split(planes, inference=inference))
            split.force_wgb = N.SZ_NN_SPLIT_WGB2
            pb, vb = (t.clone() for t in split(planes, inference=inference))
            pr, vr = (t.clone().flip(0) for t in split(planes.flip(0).contiguous(), inference=inference))
            split.force_wgb = 0
            ps, vs = (t.clone() for t in split(planes[5:6], inference=inference))                 # one board alone
            p37, v37 = (t.clone() for t in split(planes[:37], inference=inference))               # one board per workgroup (37 <= #CUs)
            assert torch.equal(pa, pb) and torch.equal(va, vb) and torch.equal(pb, pr) and torch.equal(vb, vr)
            assert torch.equal(ps[0], pa[5]) and torch.equal(vs[0], va[5]) and torch.equal(p37, pa[:37]) and torch.equal(v37, va[:37])
            if inference:
                assert torch.allclose(pa.sum(1), torch.ones(B, device="cuda"), atol=1e-5)
