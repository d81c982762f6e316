"""Deterministic stand-in for policyNN used by the reference-derived chess fixtures (tests/golden/gen_reference_chess_fixtures.py)
and by the tests that replay them.  model(x[B,119,8,8], inference=True) -> (policy [B,4672] f32, value [B,1] f32), a pure integer
function of the input planes (64-bit mixing, no floating-point reductions), so the SAME numbers come out in the build container
(where the reference's mcts.py / sim.py called it) and on the GPU box (where the oracle and the HIP engine are fed by it).

modes
  "dyadic"   policy entries k/1024, k in 1..64 (about 1 in 128 exactly 0): every masked sum is exact in fp32 in ANY order, so the
             reference's torch.sum (mcts.py:79) and the engine's fixed-order sum agree bit for bit -> whole trees compare bitwise;
  "rational" policy entries f32(m)/f32(100003), m in 1..9973: sums round, so the summation order matters in the last ulp -> these
             cases measure the north-star tolerance (visit fractions within 1e-4, move indices exact).
"""
import numpy as np
import torch

_M64 = (1 << 64) - 1
_ACTIONS = 4672
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)"""
    z = (z + _GOLD).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


_MULT = _splitmix(np.arange(1, 119 * 8 + 1, dtype=np.uint64))
_IDX = np.arange(_ACTIONS, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)


def pack_planes(x):
    """[...,119,8,8] 0/1 (any dtype) -> [...,119,8] uint8, bit j of a byte = column j (generate_training_supervised.py:91)"""
    x = np.asarray(x).astype(np.uint8)
    return (x << np.arange(8, dtype=np.uint8)).sum(axis=-1).astype(np.uint8)


def unpack_planes(p):
    return ((np.asarray(p)[..., None] >> np.arange(8, dtype=np.uint8)) & 1).astype(np.uint8)


def planes_key(packed):
    """[119,8] uint8 -> uint64 key"""
    with np.errstate(over="ignore"):
        return np.uint64(np.sum((packed.reshape(-1).astype(np.uint64) + np.uint64(1)) * _MULT, dtype=np.uint64))


def evaluate_packed(packed, mode="dyadic", salt=0):
    """one board: packed planes [119,8] uint8 -> (policy f32[4672], value f32)"""
    with np.errstate(over="ignore"):
        key = planes_key(packed) ^ _splitmix(np.uint64(salt & _M64))
        h = _splitmix(key + _IDX)
        if mode == "dyadic":
            pol = (((h >> np.uint64(58)) + np.uint64(1)).astype(np.float32)) / np.float32(1024.0)
            pol[((h >> np.uint64(40)) & np.uint64(127)) == 0] = np.float32(0.0)
        elif mode == "rational":
            pol = (((h >> np.uint64(40)) % np.uint64(9973)) + np.uint64(1)).astype(np.float32) / np.float32(100003.0)
        else:
            raise ValueError(mode)
        hv = _splitmix(key ^ np.uint64(0xA5A5A5A5DEADBEEF))
        val = np.float32((int(hv >> np.uint64(40)) - (1 << 23))) / np.float32(1 << 23)
    return pol.astype(np.float32), np.float32(val)


class HashModel:
    """Callable with the policyNN surface the search touches: .to(), .eval(), .parameters(), __call__(x, inference)."""

    def __init__(self, mode="dyadic", salt=0, record=False):
        self.mode, self.salt = mode, int(salt)
        self.record = record
        self.calls = []                      # packed planes of every board evaluated (when record=True)
        self._p = torch.zeros(1)

    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        yield self._p

    def __call__(self, x, inference=False):
        assert inference, "the search calls model(x, inference=True) (mcts.py:72-75)"
        dev = x.device
        xb = x.detach().to("cpu").float().numpy()
        assert xb.ndim == 4 and xb.shape[1:] == (119, 8, 8)
        assert ((xb == 0) | (xb == 1)).all()
        packed = pack_planes(xb)
        pol = np.zeros((xb.shape[0], _ACTIONS), np.float32)
        val = np.zeros((xb.shape[0], 1), np.float32)
        for b in range(xb.shape[0]):
            pol[b], val[b, 0] = evaluate_packed(packed[b], self.mode, self.salt)
            if self.record:
                self.calls.append(packed[b].copy())
        return torch.from_numpy(pol).to(dev), torch.from_numpy(val).to(dev)
