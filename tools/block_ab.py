"""A/B timing of sz_nn_block_bf16 flag variants at B boards: 19 chained fused blocks (the tower's buffer pattern) per sample,
variants interleaved so that clock drift hits all of them alike.  usage: block_ab.py B flag_hex [flag_hex ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from sigma_zero_amd import _native as N
from sigma_zero_amd.fastnet import _pack

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variants = [int(v, 16) for v in sys.argv[2:]] or [0, 0x200000]
torch.manual_seed(0)
a = torch.relu(torch.randn(B, 64, 256, device="cuda")).to(torch.bfloat16)
c = torch.empty_like(a)
ws = [(_pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True), _pack(torch.randn(256, 256, 3, 3) * 0.02, 256, 3, "cuda", w16=True)) for _ in range(19)]
bias = torch.zeros(256, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def tower(flag):
    x, y = a, c
    for w1, w2 in ws:
        N.check(N.lib().sz_nn_block_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w1.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(w2.data_ptr()),
                                         C.c_void_p(bias.data_ptr()), C.c_void_p(y.data_ptr()), B, N.SZ_NN_W16 | flag, st), "block")
        x, y = y, x

for v in variants: tower(v)
torch.cuda.synchronize()
res = {v: [] for v in variants}
for rep in range(16):
    for v in variants:
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): tower(v)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t) / 5 / 19 * 1e3)
for v in variants:
    r = sorted(res[v])
    print("flag %#9x: per block median %.4f ms  min %.4f  max %.4f" % (v, r[len(r) // 2], r[0], r[-1]), flush=True)
