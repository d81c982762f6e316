"""Batched self-play driver: B boards stepped in lock-step through the HIP engine with one batched
network call per simulation.  This is the GPU counterpart of sim.py:31-99 + mcts.py:39-122."""
import ctypes as C

import numpy as np
import torch

from . import _native as N

NOISE_REFERENCE = float(np.float32(1.0) - np.float32(2.0 ** -24))     # degenerate Dirichlet draw, SURVEY §8(a) A19


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def model_device(model, default=None):
    """The CUDA device a network lives on (FastPolicyNet.device, or its first parameter's), else `default`, else the current device.
    Engines are created THERE: a rank with local_rank > 0 must not silently search on GPU 0."""
    dev = getattr(model, "device", None)
    if dev is None and hasattr(model, "parameters"):
        try:
            dev = next(iter(model.parameters())).device
        except (StopIteration, TypeError):
            dev = None
    if dev is not None and torch.device(dev).type == "cuda":
        dev = torch.device(dev)
        return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
    if default is not None:
        return torch.device(default)
    return torch.device("cuda", torch.cuda.current_device())


class SelfPlayEngine:
    def __init__(self, model, args, n_boards, chess960=False, learning=True, device=None, planes_dtype=torch.float32,
                 noise_value=NOISE_REFERENCE, edges_per_board=0):
        if not torch.cuda.is_available():
            raise RuntimeError("SelfPlayEngine needs an MI355X (HIP) device: the search has no CPU fallback")
        self.device = torch.device(device) if device is not None else model_device(model)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model
        self.args = dict(args)
        self.B = int(n_boards)
        self.S = int(args["num_searches"])
        self.chess960 = bool(chess960)
        self.planes_dtype = planes_dtype
        # planes_dtype: torch.float32 / torch.bfloat16 -> [B,119,8,8] NCHW (reference layout);
        #               "nhwc128" -> [B,64,128] bf16 position-major for FastPolicyNet (csrc/sz_nn.hip)
        #               "bits128" -> the same image bit-packed, [B,1024] uint8 (1 KiB per board), expanded by the stem kernel
        self.nhwc = planes_dtype == "nhwc128"
        self.bits = planes_dtype == "bits128"
        code = N.SZ_PLANES_NHWC128_BITS if self.bits else N.SZ_PLANES_NHWC128_BF16 if self.nhwc else (N.SZ_PLANES_BF16 if planes_dtype == torch.bfloat16 else N.SZ_PLANES_F32)
        cfg = N.sz_config(self.B, self.S, float(args["C"]), int(bool(learning)), float(noise_value), int(self.chess960),
                          int(edges_per_board), code, self.device.index or 0, int(bool(self.args.get("reuse_subtree", False))))
        self._e = C.c_void_p()
        N.check(N.lib().sz_create(C.byref(cfg), C.byref(self._e)), "sz_create")
        dev = self.device
        if self.bits:
            self.planes = torch.zeros(self.B, 1024, dtype=torch.uint8, device=dev)
        elif self.nhwc:
            self.planes = torch.zeros(self.B, 64, 128, dtype=torch.bfloat16, device=dev)
        else:
            self.planes = torch.zeros(self.B, N.SZ_PLANES, 8, 8, dtype=planes_dtype, device=dev)
        self.uniforms = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.root_action = torch.zeros(self.B, N.SZ_MAX_MOVES, dtype=torch.int32, device=dev)
        self.root_visits = torch.zeros(self.B, N.SZ_MAX_MOVES, dtype=torch.int32, device=dev)
        self.root_nchild = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.root_prior = torch.zeros(self.B, N.SZ_MAX_MOVES, dtype=torch.float32, device=dev)
        self.root_wsum = torch.zeros(self.B, N.SZ_MAX_MOVES, dtype=torch.float64, device=dev)
        self.nn_seconds = 0.0
        self.n_rows = self.B          # rows of the network batch in use (sz_compact: live boards only)
        # NON-REFERENCE option (default off): args["root_dirichlet_alpha"] = alpha switches from the reference's noise (the constant
        # 1-2^-24 at every expansion, mcts.py:91-98) to AlphaZero's Dirichlet(alpha) noise on the root's children only
        self.root_alpha = self.args.get("root_dirichlet_alpha")
        if self.root_alpha is not None and self.args.get("reuse_subtree", False):
            raise ValueError("args['root_dirichlet_alpha'] and args['reuse_subtree'] exclude each other: a reused root was expanded as an inner node "
                             "(un-noised priors) and is never expanded again, so the root noise would reach the first ply of a game only")
        self._gamma = None

    def close(self):
        if self._e:
            N.lib().sz_destroy(self._e)
            self._e = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new_games(self, scharnagl=None, active=None):
        sch = None if scharnagl is None else np.ascontiguousarray(scharnagl, dtype=np.int32)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        N.check(N.lib().sz_new_games(self._e, None if sch is None else sch.ctypes.data_as(C.c_void_p),
                                     None if act is None else act.ctypes.data_as(C.c_void_p), self._stream()), "sz_new_games")

    def upload_game(self, board, chess_tensor):
        ring, ply, _ = chess_tensor.export_ring()
        N.check(N.lib().sz_upload_game(self._e, int(board), ring, int(ply), self._stream()), "sz_upload_game")

    def set_active(self, active):
        act = np.ascontiguousarray(active, dtype=np.uint8)
        N.check(N.lib().sz_set_active(self._e, act.ctypes.data_as(C.c_void_p), self._stream()), "sz_set_active")

    def compact(self, enable=True):
        """Number the boards that will search next 0..n_live-1 (sz_compact): the network then runs on planes[:n_live] only.
        Call between searches after play() / new_games() / set_active() changed the live set.  Returns n_live."""
        n = C.c_int32()
        N.check(N.lib().sz_compact(self._e, int(bool(enable)), C.byref(n), self._stream()), "sz_compact")
        self.n_rows = int(n.value) if enable else self.B
        return self.n_rows

    def evaluate(self, planes):
        """model(x, inference=True) -> (policy probabilities [B,4672] f32, value [B] f32)"""
        policy, value = self.model(planes, inference=True)
        return policy.float().contiguous(), value.float().reshape(-1).contiguous()

    def set_root_noise(self, gamma):
        """gamma: [B, SZ_MAX_MOVES] f32 device tensor of Gamma(alpha,1) draws, or None for the reference behaviour"""
        self._gamma = None if gamma is None else gamma.to(self.device, torch.float32).contiguous()
        N.check(N.lib().sz_set_root_noise(self._e, _ptr(self._gamma)), "sz_set_root_noise")

    def begin(self):
        if self.root_alpha is not None:
            alpha = torch.full((self.B, N.SZ_MAX_MOVES), float(self.root_alpha), device=self.device)
            self.set_root_noise(torch._standard_gamma(alpha))
        N.check(N.lib().sz_search_begin(self._e, _ptr(self.planes), self._stream()), "sz_search_begin")

    def step(self, policy, value):
        N.check(N.lib().sz_search_step(self._e, _ptr(policy), _ptr(value), _ptr(self.planes), self._stream()), "sz_search_step")

    @torch.no_grad()
    def search(self, evaluator=None):
        """All num_searches simulations for every active board (mcts.py:49-109)."""
        ev = evaluator or self.evaluate
        self.begin()
        if self.n_rows <= 0:
            return
        planes = self.planes if self.n_rows == self.B else self.planes[:self.n_rows]
        for _ in range(self.S):
            policy, value = ev(planes)
            self.step(policy, value)

    def stats(self):
        st = N.sz_stats()
        N.check(N.lib().sz_get_stats(self._e, C.byref(st), self._stream()), "sz_get_stats")
        return {k: getattr(st, k) for k, _ in N.sz_stats._fields_}

    def check_errors(self):
        st = self.stats()
        if st["boards_error"]:
            raise N.NativeError(st["first_error"], "search (%d boards)" % st["boards_error"])
        return st

    def root_children(self):
        N.check(N.lib().sz_root_children(self._e, _ptr(self.root_action), _ptr(self.root_visits), _ptr(self.root_nchild),
                                         _ptr(self.root_prior), _ptr(self.root_wsum), self._stream()), "sz_root_children")
        torch.cuda.synchronize(self.device)
        return (self.root_action.cpu().numpy(), self.root_visits.cpu().numpy(), self.root_nchild.cpu().numpy(),
                self.root_prior.cpu().numpy(), self.root_wsum.cpu().numpy())

    def play(self, uniforms):
        """sim.py:68-76 for every board: sample with the given uniforms (one per board), play, test game over."""
        self.uniforms.copy_(torch.as_tensor(np.asarray(uniforms, dtype=np.float64)))
        N.check(N.lib().sz_play(self._e, _ptr(self.uniforms), self._stream()), "sz_play")

    def fetch_ply(self):
        B = self.B
        rec = dict(packed=np.zeros((B, N.SZ_PLANES, 8), np.uint8), action=np.zeros((B, N.SZ_MAX_MOVES), np.int32),
                   visits=np.zeros((B, N.SZ_MAX_MOVES), np.int32), n_child=np.zeros(B, np.int32), colour=np.zeros(B, np.uint8),
                   chosen=np.zeros(B, np.int32), game_over=np.zeros(B, np.uint8), result=np.zeros(B, np.int8), active=np.zeros(B, np.uint8))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        N.check(N.lib().sz_fetch_ply(self._e, p(rec["packed"]), p(rec["action"]), p(rec["visits"]), p(rec["n_child"]), p(rec["colour"]),
                                     p(rec["chosen"]), p(rec["game_over"]), p(rec["result"]), p(rec["active"]), self._stream()), "sz_fetch_ply")
        return rec

    def debug_pending(self):
        B = self.B
        mask = np.zeros((B, N.SZ_MASK_WORDS), np.uint64)
        depth, n_nodes, n_edges, status = (np.zeros(B, np.int32) for _ in range(4))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        N.check(N.lib().sz_debug_pending(self._e, p(mask), p(depth), p(n_nodes), p(n_edges), p(status), self._stream()), "sz_debug_pending")
        return mask, depth, n_nodes, n_edges, status

    def debug_tree(self, board):
        """whole tree of one board, depth-first in child order: (depth, action, visits, value_sum, prior) arrays; row 0 = the root"""
        n = C.c_int32()
        N.check(N.lib().sz_debug_tree(self._e, int(board), 0, None, None, None, None, None, C.byref(n), self._stream()), "sz_debug_tree")
        k = n.value
        d, a, v = (np.zeros(k, np.int32) for _ in range(3))
        w, pr = np.zeros(k, np.float64), np.zeros(k, np.float32)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        N.check(N.lib().sz_debug_tree(self._e, int(board), k, p(d), p(a), p(v), p(w), p(pr), C.byref(n), self._stream()), "sz_debug_tree")
        return d, a, v, w, pr

    def debug_position(self, board):
        pos = np.zeros(10, np.uint64)
        ply = C.c_int32()
        N.check(N.lib().sz_debug_position(self._e, int(board), pos.ctypes.data_as(C.c_void_p), C.byref(ply), self._stream()), "sz_debug_position")
        return pos, ply.value


def unpack_bits128(planes):
    """[B,1024] uint8 (SZ_PLANES_NHWC128_BITS) -> [B,64,128] bf16 NHWC image (tests / inspection)."""
    B = planes.shape[0]
    v = planes.view(B, 4, 16, 16)                                   # [psub][cq][q] bytes
    bits = (v.unsqueeze(-1) >> torch.arange(8, device=planes.device, dtype=torch.uint8)) & 1      # [B,psub,cq,q,k]
    img = bits.permute(0, 3, 1, 2, 4).reshape(B, 64, 128)           # position = q*4 + psub, channel = cq*8 + k
    return img.to(torch.bfloat16)


def unpack_planes(packed):
    """(…,119,8) uint8 -> (…,119,8,8) bool; bit j of a byte = column j (train_RL.py:42 collatefn)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    return np.unpackbits(packed[..., None], axis=-1, bitorder="little").view(np.bool_)
