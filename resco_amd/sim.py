"""ctypes binding of the C ABI in include/resco_sim.h (resco_amd/csrc/libresco_sim.so).

This is the only gateway to the simulator: there is NO CPU fallback.  If the HIP library is
missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

from ._abi import ParamsStruct, pack_scenario

_HERE = os.path.dirname(os.path.abspath(__file__))
# RESCO_SIM_LIB: another build of the same HIP library (A/B variants compiled with different -D switches, tools/ab.py)
LIB_PATH = os.environ.get('RESCO_SIM_LIB') or os.path.join(_HERE, 'csrc', 'libresco_sim.so')

BUFFERS = ['lane_agg', 'drq_norm', 'phase', 'mplight', 'wave', 'wait', 'wait_norm', 'pressure', 'queue_sum',
           'queue_max', 'actions', 'env', 'tls', 'veh_pos', 'veh_speed', 'veh_accel', 'veh_tloss', 'veh_lane',
           'veh_trip', 'veh_cursor', 'veh_swait', 'veh_rwait', 'veh_depart', 'veh_owner', 'stats', 'drq_norm_f16',
           'veh_sf', 'veh_wtot', 'trip_log', 'dep_next', 'veh_coop', 'veh_cooplead', 'arrivals', 'departures',
           'mplight_full', 'lane_arrivals', 'veh_coop_odd', 'veh_cooplead_odd', 'veh_mail']
BUF_ID = {n: i for i, n in enumerate(BUFFERS)}
OUTPUT_GROUPS = ('lane_agg', 'drq_norm', 'drq_norm_f16', 'lane_arrivals', 'mplight', 'wave', 'mplight_full', 'veh_accel')
_NP_DTYPES = [np.float32, np.int32, np.uint16, np.uint8, np.float16, np.int64, np.uint32]
_TYPESTR = ['<f4', '<i4', '<u2', '|u1', '<f2', '<i8', '<u4']
STAT_KEYS = ['inserted', 'arrived', 'sum_duration', 'sum_depart_delay', 'sum_waiting', 'sum_time_loss_q10',
             'active', 'pending', 'active_ticks', 'ticks', 'cap_blocked', 'invariant']
TRIP_NONE = 0xFFFF

# every symbol include/resco_sim.h declares
ABI_SYMBOLS = ['rs_create', 'rs_destroy', 'rs_last_error', 'rs_reset', 'rs_step', 'rs_sync', 'rs_reinit_signals', 'rs_ticks', 'rs_step_sim', 'rs_set_outputs', 'rs_act_random',
               'rs_act_maxwave', 'rs_get_buffer', 'rs_read_buffer', 'rs_stats', 'rs_snapshot', 'rs_restore',
               'rs_snapshot_free', 'rs_timing', 'rs_timing_read', 'rs_set_seed', 'rs_phase_profile', 'rs_info',
               'rs_idqn_create', 'rs_idqn_act', 'rs_idqn_set_device_weights', 'rs_idqn_set_lanes', 'rs_idqn_destroy', 'rs_group_step',
               'rs_default_block']

_lib = None


def bind(L):
    """ctypes signatures of the C ABI (include/resco_sim.h) on an opened library."""
    vp, i32 = C.c_void_p, C.c_int32
    L.rs_create.argtypes = [vp, vp, i32, i32, i32, i32, C.POINTER(vp)]
    L.rs_destroy.argtypes = [vp]
    L.rs_destroy.restype = None
    L.rs_last_error.argtypes = [vp]
    L.rs_last_error.restype = C.c_char_p
    L.rs_reset.argtypes = [vp, vp]
    L.rs_step.argtypes = [vp, vp, i32, vp]
    L.rs_sync.argtypes = [vp]
    L.rs_ticks.argtypes = [vp, i32, vp]
    L.rs_step_sim.argtypes = [vp, i32, vp]
    L.rs_set_outputs.argtypes = [vp, C.c_uint64]
    L.rs_reinit_signals.argtypes = [vp, vp]
    L.rs_act_random.argtypes = [vp, C.c_uint32, vp]
    L.rs_act_maxwave.argtypes = [vp, vp, i32, vp, vp, i32, vp]
    L.rs_get_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(i32), C.POINTER(i32)]
    L.rs_read_buffer.argtypes = [vp, i32, vp, C.c_int64]
    L.rs_stats.argtypes = [vp, vp]
    L.rs_snapshot.argtypes = [vp, C.POINTER(vp)]
    L.rs_restore.argtypes = [vp, vp]
    L.rs_snapshot_free.argtypes = [vp, vp]
    L.rs_snapshot_free.restype = None
    L.rs_timing.argtypes = [vp, i32]
    L.rs_timing_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
    L.rs_set_seed.argtypes = [vp, C.c_uint32]
    L.rs_phase_profile.argtypes = [vp, i32, vp]
    L.rs_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.rs_group_step.argtypes = [vp, i32, vp, i32]
    if hasattr(L, 'rs_default_block'):      # (the host emulation of the CPU tests exports only what it implements)
        L.rs_default_block.argtypes = [i32, i32, i32]
        L.rs_default_block.restype = i32
    return L


def load_library():
    """Load libresco_sim.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('HIP extension %s is missing: build it with `python -m resco_amd.build` '
                           '(there is no CPU fallback)' % LIB_PATH)
    # PyTorch ships its own copy of the HIP runtime: when it is going to be used in this process (device tensors over the
    # library's buffers, torch.distributed), it has to be the one that is loaded first -- a process that initialises
    # /opt/rocm's runtime through this library and torch's afterwards ends up with "No HIP GPUs are available" in torch.
    if os.environ.get('RESCO_NO_TORCH') != '1':
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


AGENT = {'none': 0, 'random': 1, 'maxwave': 2, 'maxpressure': 3, 'idqn': 4}      # enum rs_agent


class GroupAgent(C.Structure):
    """ctypes mirror of rs_group_agent (include/resco_sim.h)"""
    _fields_ = [('kind', C.c_int32), ('step_key', C.c_uint32), ('policy', C.c_void_p), ('mode', C.c_int32),
                ('epsilon', C.c_float), ('epsilon_step', C.c_float), ('seed', C.c_uint32)]


class SimGroup:
    """The handles ("pipes") of one GPU stepped together by ONE call through the C ABI per env-step (rs_group_step): for every
    pipe the agent's kernel and the step kernel on the pipe's own stream.  agent: 'none' | 'random' | 'maxwave' | 'maxpressure' |
    'idqn' (policy = an rs_policy_handle, e.g. FusedIDQN.handle)."""

    def __init__(self, sims):
        self.sims = list(sims)
        self._lib = self.sims[0]._lib
        self._hs = (C.c_void_p * len(self.sims))(*[s._h for s in self.sims])
        self._agent = GroupAgent()

    def step(self, agent='none', step_key=0, n_steps=1, policy=None, mode=0, epsilon=0.0, epsilon_step=0.0, seed=0):
        a = self._agent
        a.kind, a.step_key, a.policy = AGENT[agent], int(step_key) & 0xFFFFFFFF, policy
        a.mode, a.epsilon, a.epsilon_step, a.seed = int(mode), float(epsilon), float(epsilon_step), int(seed) & 0xFFFFFFFF
        if agent in ('maxwave', 'maxpressure'):
            for s in self.sims:
                s.require_output('mplight' if agent == 'maxpressure' else 'wave')
                s._ensure_maxwave_tables(1 if agent == 'maxpressure' else 0)
        rc = self._lib.rs_group_step(self._hs, len(self.sims), C.byref(a), int(n_steps))
        if rc != 0:
            msgs = [m.decode() for m in (self._lib.rs_last_error(s._h) for s in self.sims) if m]
            raise RuntimeError('rs_group_step failed (%d): %s' % (rc, '; '.join(msgs)))

    def sync(self):
        for s in self.sims:
            s.sync()


def _murmur(seed, words):
    """the counter-based hash of the model (resco_amd/csrc/resco_step.h: d_hash) on the host"""
    M = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
    h = seed & M
    for k in words:
        k = (k * 0xcc9e2d51) & M
        k = rotl(k, 15)
        k = (k * 0x1b873593) & M
        h ^= k
        h = rotl(h, 13)
        h = (h * 5 + 0xe6546b64) & M
    h ^= 16
    h ^= h >> 16
    h = (h * 0x85ebca6b) & M
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & M
    h ^= h >> 16
    return h


def speed_factor(seed, env, trip, vt_row, speed_dev=1):
    """speedFactor of a trip (resco_step.h: speed_factor): a function of (seed, global env index, trip) only, so it can be
    recomputed for trips that have left the network (tripinfo output)"""
    mean, dev = np.float32(vt_row[7]), np.float32(vt_row[8])
    f = mean
    if speed_dev:
        s = np.float32(0.0)
        for i in range(4):
            s = np.float32(s + np.float32(_murmur(seed, (env, trip, 0xFFFFFFFF, i)) >> 8) * np.float32(1.0 / 16777216.0))
        z = np.float32((s - np.float32(2.0)) * np.float32(1.7320508))
        f = np.float32(mean + np.float32(dev * z))
    f = min(max(np.float32(f), np.float32(0.2)), np.float32(2.0))
    return float(np.float32(int(np.float32(f * np.float32(4096.0)) + np.float32(0.5))) * np.float32(1.0 / 4096.0))     # RM_SF_QUANT


def torch_stream(device=None):
    """The stream argument that orders a launch with PyTorch's current stream.  PyTorch's default stream has the
    handle 0, which the C ABI reads as "the handle's own stream": it is mapped to hipStreamLegacy (1)."""
    import torch
    p = torch.cuda.current_stream(device).cuda_stream
    return p if p else 1


class _DevArray:
    """Zero-copy view of a library-owned device buffer (__cuda_array_interface__, also honoured on ROCm)."""

    def __init__(self, ptr, shape, dtype_code, owner):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=_TYPESTR[dtype_code],
                                             data=(int(ptr), False), version=2, strides=None)
        self._owner = owner


class BatchedSim:
    """N lock-step environments of one scenario on one GPU (one rs_handle)."""

    def __init__(self, scenario, n_envs, device=0, seed=0, max_distance=200.0, sigma=-1.0, speed_dev=1,
                 fixed_program=0, env_base=0, step_length=10, yellow_length=None, block_threads=0, trip_log=0, step_ratio=1,
                 tls_expiry=1, device_envs=None):
        """device_envs: how many environments share this GPU when the batch is split over several handles (pipes) -- the default
        workgroup shape (block_threads = 0) is chosen for the device's load, not for this handle's share (rs_default_block)"""
        self.sc = scenario
        self.n_envs = int(n_envs)
        self.device = int(device)
        self._lib = self._load()
        if yellow_length is not None and int(yellow_length) != int(scenario.yellow_length):
            # the yellow phases are compiled into the scenario's programmes with ITS yellow_length; the FSM would call
            # set_phase after a different number of ticks
            raise ValueError('scenario %s was compiled with yellow_length=%d (asked %d)' % (scenario.name, scenario.yellow_length, yellow_length))
        self._st, self._keep = pack_scenario(scenario, step_length, yellow_length)
        self._p = ParamsStruct(int(seed) & 0xFFFFFFFF, float(max_distance), float(sigma), int(speed_dev),
                               int(fixed_program), int(trip_log), int(step_ratio), 0 if tls_expiry else 1)     # rs_params.tls_hold
        self.step_ratio = max(1, int(step_ratio))
        self._h = C.c_void_p()
        if not block_threads and device_envs and int(device_envs) != self.n_envs and hasattr(self._lib, 'rs_default_block'):
            block_threads = self._lib.rs_default_block(int(scenario.capacity), int(device_envs), self.device)
        rc = self._lib.rs_create(C.byref(self._st), C.byref(self._p), self.n_envs, int(env_base), self.device,
                                 int(block_threads), C.byref(self._h))
        if rc != 0:
            msg = self._lib.rs_last_error(None)
            self._h = None
            raise RuntimeError('rs_create failed (%d): %s' % (rc, msg.decode() if msg else '?'))
        self.S, self.O, self.C = scenario.n_signals, scenario.n_obs, scenario.capacity
        self.seed, self.env_base, self.speed_dev = int(seed) & 0xFFFFFFFF, int(env_base), int(speed_dev)
        self._maxwave_ready = False
        self._masked_off = frozenset()          # maskable output buffers the observe does not write at present (set_outputs)
        self._dep_cache = None
        self._meta = {}
        for name, bid in BUF_ID.items():
            ptr, shape, nd, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
            self._check(self._lib.rs_get_buffer(self._h, bid, C.byref(ptr), shape, C.byref(nd), C.byref(dt)))
            self._meta[name] = (ptr.value, tuple(shape[:nd.value]), dt.value)

    # ------------------------------------------------------------------ plumbing
    def _load(self):
        return load_library()

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.rs_last_error(self._h)
            raise RuntimeError('resco_sim error %d: %s' % (rc, msg.decode() if msg else '?'))

    def close(self):
        if getattr(self, '_h', None):
            self._lib.rs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.rs_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(n_envs=a.value, block_threads=b.value, lds_bytes=c.value, max_lanes_per_signal=d.value)

    # ------------------------------------------------------------------ stepping
    def reset(self, stream=None):
        self._check(self._lib.rs_reset(self._h, stream))

    def step(self, actions=None, stream=None):
        """actions: None (use the on-device action buffer), a numpy int32 [N,S] array, or a torch
        CUDA int32 tensor [N,S]."""
        if actions is None:
            self._check(self._lib.rs_step(self._h, None, 0, stream))
            return
        if hasattr(actions, 'data_ptr'):        # torch tensor
            assert tuple(actions.shape) == (self.n_envs, self.S) and actions.is_contiguous()
            assert str(actions.dtype) == 'torch.int32'
            self._check(self._lib.rs_step(self._h, actions.data_ptr(), 1 if actions.is_cuda else 0, stream))
            return
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.n_envs, self.S), a.shape
        self._check(self._lib.rs_step(self._h, a.ctypes.data, 0, stream))

    def ticks(self, n, stream=None):
        """n x step_sim() without the signal FSM, then an observe (see rs_ticks)"""
        self._check(self._lib.rs_ticks(self._h, int(n), stream))

    def step_sim(self, n=1, stream=None):
        """n x sumo.simulationStep() and nothing else: no observe, the Signal state stays as it is (see rs_step_sim)"""
        self._check(self._lib.rs_step_sim(self._h, int(n), stream))

    def set_outputs(self, names=None):
        """Only the named per-lane / per-movement buffers ('lane_agg', 'drq_norm', 'drq_norm_f16', 'lane_arrivals', 'mplight',
        'wave', 'mplight_full') are written by the following observes; None = all of them.  The per-signal scalars (phase,
        wait, wait_norm, pressure, queue_sum, queue_max, arrivals, departures) are always written (see rs_set_outputs)."""
        mask = 0
        for n in (OUTPUT_GROUPS if names is None else names):
            if n not in OUTPUT_GROUPS:
                raise ValueError('%r is not a maskable output buffer (%s)' % (n, ', '.join(OUTPUT_GROUPS)))
            mask |= 1 << BUF_ID[n]
        self._check(self._lib.rs_set_outputs(self._h, mask))
        self._masked_off = frozenset(OUTPUT_GROUPS) - frozenset(OUTPUT_GROUPS if names is None else names)

    def require_output(self, name):
        """Raise when `name` is a maskable output buffer that set_outputs has switched off: its contents are stale (or zeros),
        and whoever consumes them (an on-device agent, a derived state) would act on garbage without noticing."""
        if name in self._masked_off:
            raise RuntimeError('output buffer %r is switched off (BatchedSim.set_outputs): re-enable it before reading it' % name)

    def reinit_signals(self, stream=None):
        """fresh Signal objects on the running simulation (see rs_reinit_signals)"""
        self._check(self._lib.rs_reinit_signals(self._h, stream))

    def set_seed(self, seed):
        self.seed = int(seed) & 0xFFFFFFFF
        self._check(self._lib.rs_set_seed(self._h, self.seed))

    def sync(self):
        self._check(self._lib.rs_sync(self._h))

    def act_random(self, step_key, stream=None):
        self._check(self._lib.rs_act_random(self._h, int(step_key) & 0xFFFFFFFF, stream))

    def act_maxwave(self, use_pressure, stream=None):
        if not self._maxwave_ready:
            pairs, valid, order = maxwave_tables(self.sc)
            self._pairs, self._valid, self._order = pairs, valid, order
            self._check(self._lib.rs_act_maxwave(self._h, pairs.ctypes.data, len(pairs), valid.ctypes.data,
                                                 order.ctypes.data, int(use_pressure), stream))
            self._maxwave_ready = True
            return
        self._check(self._lib.rs_act_maxwave(self._h, None, len(self._pairs), None, None, int(use_pressure), stream))

    def _ensure_maxwave_tables(self, use_pressure):
        """the phase-pair tables live on the device after the first rs_act_maxwave call (rs_group_step needs them there)"""
        if not self._maxwave_ready:
            self.act_maxwave(use_pressure)

    # ------------------------------------------------------------------ buffers
    def read(self, name):
        """Synchronous host copy of a buffer as a numpy array."""
        ptr, shape, dt = self._meta[name]
        out = np.empty(shape, _NP_DTYPES[dt])
        if out.nbytes == 0:
            return out
        self._check(self._lib.rs_read_buffer(self._h, BUF_ID[name], out.ctypes.data, out.nbytes))
        return out

    def tensor(self, name, allow_masked=False):
        """Zero-copy torch tensor over the library-owned device buffer (agent boundary).  A buffer that set_outputs has
        switched off is refused (stale contents) unless allow_masked."""
        import torch
        if not allow_masked:
            self.require_output(name)
        ptr, shape, dt = self._meta[name]
        return torch.as_tensor(_DevArray(ptr, shape, dt, self), device='cuda:%d' % self.device)

    def device_pointer(self, name):
        return self._meta[name]

    def outputs(self, names=('lane_agg', 'drq_norm', 'phase', 'mplight', 'wave', 'wait', 'wait_norm', 'pressure',
                             'queue_sum', 'queue_max')):
        return {n: self.read(n) for n in names}

    def stats(self):
        st = self.read('stats')
        return {k: st[:, i].copy() for i, k in enumerate(STAT_KEYS)}

    def time(self):
        return self.read('env')[:, 0]

    def _dep_lanes(self):
        """The departure lanes in the kernel's numbering (resco_tables.h: the first allowed lane of the first edge of every
        ROUTE, ascending) and, per lane, its trips in FIFO order with their departure seconds and running sums."""
        if self._dep_cache is None:
            A = self.sc.arrays
            lane_of_route = A['edge_lane0'][A['route_edge'][A['route_start'][:-1]]]
            lanes = np.unique(lane_of_route)
            n_dep = self._meta['dep_next'][1][1]
            if len(lanes) != n_dep:
                raise RuntimeError('departure lanes of the scenario (%d) do not match the library (%d)' % (len(lanes), n_dep))
            lane_of_trip = lane_of_route[A['trip_route']]
            depart = A['trip_depart'].astype(np.int64)
            tabs = []
            for lane in lanes:
                trips = np.nonzero(lane_of_trip == lane)[0]          # ascending trip index = FIFO order = departure order
                dep = depart[trips]
                tabs.append((trips, dep, np.concatenate([[0], np.cumsum(dep)])))
            self._dep_cache = tabs
        return self._dep_cache

    def backlog(self):
        """Trips that have departed (depart < now) but are not on the network yet, per environment: (count, seconds
        waited so far).  Every departure lane keeps its own backlog (RS_BUF_DEP_NEXT = its next trip); utils/readXML.py:52-68
        charges such trips end_time - depart."""
        head = self.read('dep_next').astype(np.int64)            # [N, n_dep]
        now = self.time().astype(np.int64)
        cnt = np.zeros(self.n_envs, np.int64)
        waited = np.zeros(self.n_envs, np.int64)
        for d, (trips, dep, csum) in enumerate(self._dep_lanes()):
            h = head[:, d]
            first = np.where(h == TRIP_NONE, len(trips), np.searchsorted(trips, h))
            last = np.maximum(first, np.searchsorted(dep, now, side='left'))      # trips [first, last) are due and waiting
            cnt += last - first
            waited += (last - first) * now - (csum[last] - csum[first])
        return cnt, waited

    def trip_delay(self):
        """Per-environment average trip delay exactly as the reference's post-processing computes it, see trip_metrics()."""
        return self.trip_metrics()['delay']

    def trip_metrics(self):
        """The per-episode figures of utils/readXML.py:16-77 per environment.
        `delay`: (timeLoss + departDelay) summed over the tripinfo entries -- the arrived and, as
        --tripinfo-output.write-unfinished writes them, the running vehicles -- divided by their number.  Demand that never got
        onto the network is added only when the rou.xml lists <vehicle> elements (cologne3): readXML.py:59-68 skips every other
        tag, so the <trip> files of five of the six maps are never charged for it.  For <vehicle> files the script's own rule is
        followed (`_readxml_never_departed`): every vehicle SCHEDULED later than the vehicle that actually departed last counts as
        never departed and is charged end_time - depart -- whether or not it is in the tripinfo file as well.
        `delay_all` charges the insertion backlogs (trips that are due and not on the network, now - depart each) on every map:
        this build's own, stricter figure, not the script's.  `duration`, `waiting`, `time_loss`: tripinfo duration /
        waitingTime / timeLoss over the same entries."""
        st = self.stats()
        lane = self.read('veh_lane')
        live = lane != 0xFFFF
        running = (self.read('veh_tloss') * live).sum(axis=1)
        cnt, waited = self.backlog()
        now = self.time().astype(np.int64)
        entries = st['arrived'] + live.sum(axis=1)
        depart = self.read('veh_depart').astype(np.int64)
        run_dur = ((now[:, None] - depart) * live).sum(axis=1)
        loss = st['sum_time_loss_q10'] / 1024.0 + running
        delay_all = (loss + st['sum_depart_delay'] + waited) / np.maximum(1, entries + cnt)
        if self.sc.demand_tag == 'vehicle':
            n_nd, charged = self._readxml_never_departed(live, depart)
            delay = (loss + st['sum_depart_delay'] + charged) / np.maximum(1, entries + n_nd)
        else:
            delay = (loss + st['sum_depart_delay']) / np.maximum(1, entries)
        entries = np.maximum(1, entries)
        return dict(delay=delay, delay_all=delay_all, duration=(st['sum_duration'] + run_dur) / entries,
                    waiting=st['sum_waiting'] / entries, time_loss=loss / entries)

    def _readxml_never_departed(self, live, depart):
        """utils/readXML.py:41-68 for <vehicle> route files, per environment: (count, seconds charged).  The script takes the
        tripinfo entry with the latest actual departure (the first one in file order among equals: arrived vehicles come first,
        in arrival order; the unfinished ones follow in id order), looks up that vehicle's SCHEDULED departure in the rou.xml and
        charges end_time - depart for every <vehicle> scheduled later.  Arrived vehicles are considered when the handle keeps
        per-trip records (trip_log=1); without them the latest departure is taken from the vehicles still on the network."""
        sched = self.sc.arrays['trip_depart'].astype(np.int64)
        horizon = int(self.sc.horizon)
        trips = self.read('veh_trip').astype(np.int64)
        ids = getattr(self.sc, 'trip_ids', None)
        log = self.read('trip_log') if self._p.trip_log else None
        n_nd = np.zeros(self.n_envs, np.int64)
        charged = np.zeros(self.n_envs, np.float64)
        for e in range(self.n_envs):
            best_t, best_k = -1, -1
            if log is not None and log.size:
                arr = np.nonzero(log[e, :, 1] > 0)[0]
                if len(arr):
                    t = log[e, arr, 0]
                    order = np.lexsort((arr, log[e, arr, 1]))           # file order: arrival time, then trip order
                    j = order[np.argmax(t[order] == t.max())]
                    best_t, best_k = int(t[j]), int(arr[j])
            run = np.nonzero(live[e])[0]
            if len(run):
                t = depart[e, run]
                if int(t.max()) > best_t:
                    cand = trips[e, run[t == t.max()]]
                    best_k = int(min(cand, key=(lambda k: ids[k]) if ids else None))
                    best_t = int(t.max())
            if best_k < 0:
                continue
            late = sched > sched[best_k]
            n_nd[e] = int(late.sum())
            charged[e] = float((horizon - sched[late]).sum())
        return n_nd, charged

    # ------------------------------------------------------------------ snapshots / timing
    def snapshot(self):
        snap = C.c_void_p()
        self._check(self._lib.rs_snapshot(self._h, C.byref(snap)))
        return snap

    def restore(self, snap):
        self._check(self._lib.rs_restore(self._h, snap))

    def free_snapshot(self, snap):
        self._lib.rs_snapshot_free(self._h, snap)

    def phase_profile(self, enable):
        """Enable/disable the in-kernel phase timers; returns the 16 accumulators collected so far."""
        out = (C.c_uint64 * 16)()
        self._check(self._lib.rs_phase_profile(self._h, 1 if enable else 0, out))
        return [int(x) for x in out]

    def timing(self, enable):
        self._check(self._lib.rs_timing(self._h, 1 if enable else 0))

    def timing_read(self):
        ms, n = C.c_float(), C.c_int32()
        self._check(self._lib.rs_timing_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def maxwave_tables(sc):
    """phase_pairs [P][2], valid [S][P] (local action of pair p or -1) and order [S][P] (pair indices in
    the reference's iteration order, -1 padded) for the batched MAXWAVE / MAXPRESSURE agent
    (agents/maxwave.py:18-38).  Without valid_acts every pair k maps to action k (np.argmax branch)."""
    pairs = np.asarray(sc.phase_pairs, np.int32).reshape(-1, 2)
    S, P = sc.n_signals, len(pairs)
    valid = np.full((S, P), -1, np.int32)
    order = np.full((S, P), -1, np.int32)
    for si, sid in enumerate(sc.signal_ids):
        va = None if sc.valid_acts is None else sc.valid_acts.get(sid)
        if va is None:
            valid[si, :] = np.arange(P)
            order[si, :] = np.arange(P)
        else:
            for j, (gidx, act) in enumerate(va.items()):        # dict order = the reference's iteration order
                valid[si, int(gidx)] = int(act)
                order[si, j] = int(gidx)
    return np.ascontiguousarray(pairs), np.ascontiguousarray(valid), np.ascontiguousarray(order)
