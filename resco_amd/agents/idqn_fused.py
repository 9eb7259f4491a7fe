"""Fused HIP forward of the per-signal IDQN networks (rs_idqn_* of include/resco_sim.h, resco_amd/csrc/resco_policy.h).

`FusedIDQN(net)` packs the weights of a `BatchedIDQN` (= the reference architecture of pfrl_dqn.py:30-39 for every
signal) into the layout the kernel wants and evaluates epsilon-greedy actions for all N environments in ONE launch on
the fp16 observation tensor the step kernel wrote: the 2x2 convolution is computed on the fly inside the MFMA loop of
the first linear layer, so the 64*H*4-wide feature tensor never touches memory.  Inference only (the rollout side
of IDQN); learning keeps using the PyTorch modules (idqn_learn.py) and re-packs after updates (`refresh()`).
"""
import ctypes as C

import numpy as np
import torch

from ..sim import load_library, torch_stream

_LANE = np.arange(64)
_J = np.arange(4)


def _b_fragments(w, n_ksteps, n_tiles):
    """w [S, K, Ncols] (K rows = reduction index) -> f16 B fragments [S, n_ksteps, n_tiles, 64 lanes, 4]:
    element (kk, nt, lane, j) = w[kk*8 + (lane >> 5)*4 + j, nt*32 + (lane & 31)], zero outside w."""
    S, K, N = w.shape
    k = (np.arange(n_ksteps)[:, None, None] * 8 + (_LANE[None, :, None] >> 5) * 4 + _J[None, None, :])      # [kk, lane, j]
    n = (np.arange(n_tiles)[:, None] * 32 + (_LANE[None, :] & 31))                                           # [nt, lane]
    kk = np.broadcast_to(k[:, None, :, :], (n_ksteps, n_tiles, 64, 4))
    nn = np.broadcast_to(n[None, :, :, None], (n_ksteps, n_tiles, 64, 4))
    ok = (kk < K) & (nn < N)
    out = np.zeros((S, n_ksteps, n_tiles, 64, 4), np.float16)
    out[:, ok] = w[:, kk[ok], nn[ok]].astype(np.float16)
    return out


def _fragment_index(K, N, n_ksteps, n_tiles, zero_index):
    """Flat gather indices into a [K, N] matrix (row-major) for the B-fragment layout; `zero_index` where padded."""
    k = (np.arange(n_ksteps)[:, None, None] * 8 + (_LANE[None, :, None] >> 5) * 4 + _J[None, None, :])
    n = (np.arange(n_tiles)[:, None] * 32 + (_LANE[None, :] & 31))
    kk = np.broadcast_to(k[:, None, :, :], (n_ksteps, n_tiles, 64, 4))
    nn = np.broadcast_to(n[None, :, :, None], (n_ksteps, n_tiles, 64, 4))
    return np.where((kk < K) & (nn < N), kk * N + nn, zero_index).astype(np.int64)


def pack_idqn_weights(net):
    """numpy arrays in the order rs_idqn_create takes them."""
    S, lmax = len(net.lanes), net.lmax
    H, hp = lmax - 1, lmax // 2
    assert 2 <= lmax <= 17 and max(net.actions) <= 8
    f = lambda t: t.detach().float().cpu().numpy()
    conv_w = f(net.conv_w).reshape(S, 64, 4)
    conv_b = f(net.conv_b).reshape(S, 64)
    w1 = f(net.fc1_w).reshape(S, 64, H * 4, 64)                       # [S, channel, (h, w), out]
    w1p = np.stack([_b_fragments(w1[:, c], hp, 2) for c in range(64)], axis=1)          # [S, 64, hp, 2, 64, 4]
    w2p = _b_fragments(f(net.fc2_w), 8, 2)
    w3p = _b_fragments(f(net.fc3_w), 8, 1)[:, :, 0]
    b3 = np.zeros((S, 32), np.float32)
    b3[:, :net.amax] = f(net.fc3_b)
    return dict(n_actions=np.asarray(net.actions, np.int32), conv_w=np.ascontiguousarray(conv_w, np.float32),
                conv_b=np.ascontiguousarray(conv_b, np.float32), w1=np.ascontiguousarray(w1p), b1=np.ascontiguousarray(f(net.fc1_b), np.float32),
                w2=np.ascontiguousarray(w2p), b2=np.ascontiguousarray(f(net.fc2_b), np.float32), w3=np.ascontiguousarray(w3p), b3=b3)


class FusedIDQN:
    def __init__(self, net, device=0, seed=0):
        self.net, self.device, self.seed = net, int(device), int(seed) & 0xFFFFFFFF
        self.S, self.lmax = len(net.lanes), net.lmax
        self._lib = load_library()
        L = self._lib
        vp = C.c_void_p
        L.rs_idqn_create.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [vp] * 9 + [C.POINTER(vp)]
        L.rs_idqn_act.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
        L.rs_idqn_destroy.argtypes = [vp]
        L.rs_idqn_destroy.restype = None
        self._h = None
        self._actions = {}
        self.refresh()

    def refresh(self):
        """(Re)pack the network's current weights, e.g. after learner updates."""
        w = pack_idqn_weights(self.net)
        h = C.c_void_p()
        rc = self._lib.rs_idqn_create(self.device, self.S, self.lmax, *[w[k].ctypes.data for k in
                                      ('n_actions', 'conv_w', 'conv_b', 'w1', 'b1', 'w2', 'b2', 'w3', 'b3')], C.byref(h))
        if rc != 0:
            msg = self._lib.rs_last_error(None)
            raise RuntimeError('rs_idqn_create failed (%d): %s' % (rc, msg.decode() if msg else '?'))
        self.close()
        self._h = h
        lanes = np.asarray(self.net.lanes, np.int32)            # the signals' own head sizes: padded lanes are skipped
        self._check_padded_rows_are_zero()
        self._lib.rs_idqn_set_lanes.argtypes = [C.c_void_p, C.c_void_p]
        if self._lib.rs_idqn_set_lanes(self._h, lanes.ctypes.data) != 0:
            raise RuntimeError('rs_idqn_set_lanes failed')

    @torch.no_grad()
    def _check_padded_rows_are_zero(self):
        """rs_idqn_set_lanes makes the kernel skip the fc1 k-steps of a signal's padded lanes: only correct while the fc1 rows of
        those lanes ARE zero (BatchedIDQN builds them so and the masked learner keeps them there).  Weights from anywhere else that
        break this would silently change the Q-values: refuse them here."""
        H = self.lmax - 1
        w = self.net.fc1_w.detach()
        for s, L in enumerate(self.net.lanes):
            if L - 1 < H:
                pad = w[s].reshape(64, H, 4, -1)[:, L - 1:]
                if bool((pad != 0).any()):
                    raise ValueError('signal %d observes %d lanes but its fc1 weights have non-zero rows for padded lanes: the fused kernel '
                                     'would skip them (rs_idqn_set_lanes)' % (s, L))

    @torch.no_grad()
    def refresh_on_device(self):
        """Re-pack the network's current weights on the GPU (no host round trip) and point the kernel at the result:
        what a learner calls after every update.  fp32 parameters whose memory already has the kernel's layout (conv
        weights, biases) are used in place; the three linear layers are gathered into persistent f16 fragment buffers
        (index_select + mask + casting copy each)."""
        net, S = self.net, self.S
        dev = net.fc1_w.device
        H, hp, A = self.lmax - 1, self.lmax // 2, net.amax
        if not hasattr(self, '_idx'):
            base = _fragment_index(H * 4, 64, hp, 2, -1)             # [hp, 2, 64, 4] into one channel's [H*4, 64] block
            i1 = np.stack([np.where(base >= 0, c * H * 4 * 64 + base, -1) for c in range(64)]).reshape(-1)
            i2 = _fragment_index(64, 64, 8, 2, -1).reshape(-1)
            i3 = _fragment_index(64, A, 8, 1, -1).reshape(-1)
            self._idx, self._mask, self._dev = {}, {}, {}
            for k, i in (('w1', i1), ('w2', i2), ('w3', i3)):
                self._idx[k] = torch.as_tensor(np.maximum(i, 0), device=dev)
                self._mask[k] = torch.as_tensor((i >= 0).astype(np.float32), device=dev)
                # + 2 KB: the kernel's copy passes of the last conv channel may read up to 1 KB past the fc1 fragments
                self._pad = getattr(self, '_pad', {})
                self._pad[k] = torch.zeros(S * len(i) + 1024, dtype=torch.float16, device=dev)
                self._dev[k] = self._pad[k][:S * len(i)].view(S, len(i))
            self._dev['b3'] = torch.zeros(S, 32, device=dev, dtype=torch.float32)
        d = self._dev
        for k, w in (('w1', net.fc1_w), ('w2', net.fc2_w), ('w3', net.fc3_w)):
            d[k].copy_(torch.index_select(w.reshape(S, -1).float(), 1, self._idx[k]).mul_(self._mask[k]))
        d['b3'][:, :A] = net.fc3_b
        in_place = net.conv_w.dtype == torch.float32
        for k, t in (('conv_w', net.conv_w), ('conv_b', net.conv_b), ('b1', net.fc1_b), ('b2', net.fc2_b)):
            d[k] = t.detach() if (in_place and t.is_contiguous()) else t.detach().float().contiguous()
        L = self._lib
        L.rs_idqn_set_device_weights.argtypes = [C.c_void_p] * 9
        rc = L.rs_idqn_set_device_weights(self._h, *[d[k].data_ptr() for k in ('conv_w', 'conv_b', 'w1', 'b1', 'w2', 'b2', 'w3', 'b3')])
        if rc != 0:
            raise RuntimeError('rs_idqn_set_device_weights failed (%d)' % rc)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.rs_idqn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def act(self, obs, epsilon=0.0, step_key=0, want_q=False, stream=None, out=None, dyn=None, sample=False, env_base=0):
        """obs: fp16 CUDA tensor [N, S, lmax, 5] (the simulator's drq_norm_f16).  Returns int32 actions [N, S]
        (`out`, e.g. the simulator's own RS_BUF_ACTIONS tensor so that env.step(None) consumes them without a copy,
        or a tensor reused between calls) and, with want_q, the Q-values [N, S, 8] (-inf beyond a signal's actions).
        dyn: optional CUDA tensor of two 32-bit words {epsilon as float32 bits, step key} read by the kernel instead of
        the scalar arguments (for HIP-graph replay).  sample=True: treat the outputs as logits and draw the action
        from softmax(logits) (IPPO policy head) instead of epsilon-greedy.  env_base: global index of obs[0]'s environment (the
        exploration draws are keyed by the global index: pass the pipe's env_base when a batch is split over handles)."""
        assert obs.is_cuda and obs.dtype == torch.float16 and obs.is_contiguous()
        N = obs.shape[0]
        assert tuple(obs.shape[1:]) == (self.S, self.lmax, 5)
        if out is not None:
            assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and tuple(out.shape) == (N, self.S)
            actions = out
        else:
            if N not in self._actions:
                self._actions[N] = torch.empty(N, self.S, dtype=torch.int32, device=obs.device)
            actions = self._actions[N]
        q = torch.empty(N, self.S, 8, dtype=torch.float32, device=obs.device) if want_q else None
        st = torch_stream(self.device) if stream is None else stream
        rc = self._lib.rs_idqn_act(self._h, obs.data_ptr(), N, int(env_base), 1 if sample else 0, float(epsilon), self.seed, int(step_key) & 0xFFFFFFFF,
                                   dyn.data_ptr() if dyn is not None else None, actions.data_ptr(),
                                   q.data_ptr() if want_q else None, st)
        if rc != 0:
            raise RuntimeError('rs_idqn_act failed (%d)' % rc)
        return (actions, q) if want_q else actions
