"""Scenario compiler: SUMO net.xml / rou.xml + signal_config -> flat tables.

The HIP simulator (resco_amd/csrc) and the CPU oracle (oracle/) never see XML or
Python dicts; they consume the flat int32/float32 tables built here.  The same
tables are shared by every environment instance of a batch (read-only, L2
resident on the GPU).

Reference surface this replaces (file:line are relative to /root/reference):
  * the SUMO start-up that loads ``<map>.sumocfg`` -> net.xml + rou.xml
    (resco_benchmark/multi_signal.py:33-47,116-137)
  * green-phase discovery (multi_signal.py:52-59)
  * ``create_yellows`` (traffic_signal.py:7-24) and the program re-install
    (traffic_signal.py:93-100)
  * ``Signal.__init__`` lane-set / outbound derivation (traffic_signal.py:49-87)
  * SUMO's load-time routing of ``<trip>`` elements (fastest path) [SUMO-K]

Everything tagged [SUMO-K] is this build's own restatement of SUMO behaviour
from general knowledge and is parity-unpinned (no SUMO in the build container).
"""
from __future__ import annotations

import heapq
import io
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# TLS link-state encoding used by every backend
TLS_R, TLS_Y, TLS_g, TLS_G = 0, 1, 2, 3
_TLS_CODE = {'r': TLS_R, 's': TLS_R, 'y': TLS_Y, 'Y': TLS_Y, 'u': TLS_R,
             'g': TLS_g, 'G': TLS_G, 'o': TLS_g, 'O': TLS_G}

MOVEMENT_SLOTS = 12            # states.mplight / wave iterate the 12 movement keys
BIG = np.float32(1.0e30)

# vType parameter columns (float32 table, one row per vType)
VT_LENGTH, VT_MINGAP, VT_ACCEL, VT_DECEL, VT_TAU, VT_SIGMA, VT_MAXSPEED, \
    VT_SF_MEAN, VT_SF_DEV, VT_EMERGENCY = range(10)
VT_COLS = 10

# [SUMO-K] vClass defaults of SUMO 1.9
_VCLASS_DEFAULTS = {
    'passenger': dict(length=5.0, minGap=2.5, accel=2.6, decel=4.5, tau=1.0, sigma=0.5,
                      maxSpeed=55.55, speedFactor=1.0, speedDev=0.1, emergencyDecel=9.0),
    'bus': dict(length=12.0, minGap=2.5, accel=1.2, decel=4.0, tau=1.0, sigma=0.5,
                maxSpeed=23.61, speedFactor=1.0, speedDev=0.1, emergencyDecel=7.0),
}


# --------------------------------------------------------------------------- XML model
@dataclass
class _Lane:
    id: str
    index: int
    speed: float
    length: float
    passenger: bool
    edge: str


@dataclass
class _Edge:
    id: str
    internal: bool
    frm: str
    to: str
    lanes: list


@dataclass
class _Conn:
    frm: str
    to: str
    from_lane: int
    to_lane: int
    via: str | None
    tl: str | None
    link_index: int
    state: str
    dir: str


@dataclass
class _Junction:
    id: str
    type: str
    inc_lanes: list
    int_lanes: list
    requests: list      # [(response:str, foes:str, cont:int)]


@dataclass
class Net:
    edges: dict
    lanes: dict
    conns: list
    junctions: dict
    tllogics: dict      # id -> [(duration:int, state:str)], in file order
    tl_order: list


def _passenger_allowed(attrib) -> bool:
    allow = attrib.get('allow')
    if allow is not None:
        return 'passenger' in allow.split()
    dis = attrib.get('disallow')
    if dis is not None:
        return 'passenger' not in dis.split()
    return True


def parse_net(path_or_file) -> Net:
    edges, lanes, conns, junctions, tllogics, tl_order = {}, {}, [], {}, {}, []
    for _, el in ET.iterparse(path_or_file, events=('end',)):
        tag = el.tag
        if tag == 'edge':
            eid = el.get('id')
            internal = el.get('function') == 'internal'
            if el.get('function') in ('crossing', 'walkingarea'):
                el.clear()
                continue
            ls = []
            for ln in el.findall('lane'):
                lane = _Lane(ln.get('id'), int(ln.get('index')), float(ln.get('speed')),
                             float(ln.get('length')), _passenger_allowed(ln.attrib), eid)
                ls.append(lane)
                lanes[lane.id] = lane
            ls.sort(key=lambda l: l.index)
            edges[eid] = _Edge(eid, internal, el.get('from'), el.get('to'), ls)
            el.clear()
        elif tag == 'connection':
            conns.append(_Conn(el.get('from'), el.get('to'), int(el.get('fromLane')), int(el.get('toLane')),
                               el.get('via'), el.get('tl'),
                               int(el.get('linkIndex')) if el.get('linkIndex') is not None else -1,
                               el.get('state', 'M'), el.get('dir', 's')))
            el.clear()
        elif tag == 'junction':
            reqs = [(r.get('response'), r.get('foes'), int(r.get('cont', '0'))) for r in el.findall('request')]
            junctions[el.get('id')] = _Junction(el.get('id'), el.get('type'),
                                                (el.get('incLanes') or '').split(),
                                                (el.get('intLanes') or '').split(), reqs)
            el.clear()
        elif tag == 'tlLogic':
            tid = el.get('id')
            if tid not in tllogics:           # programID 0 only (multi_signal.py:55 takes [0])
                tllogics[tid] = [(int(float(p.get('duration'))), p.get('state')) for p in el.findall('phase')]
                tl_order.append(tid)
            el.clear()
    return Net(edges, lanes, conns, junctions, tllogics, tl_order)


def parse_routes(path_or_file):
    """Returns (vtypes: dict id -> attrib dict, trips: list of (id, type, depart, from, to) or
    (id, type, depart, [edges]))."""
    vtypes, trips = {}, []
    root = ET.parse(path_or_file).getroot()
    for el in root:
        if el.tag == 'vType':
            vtypes[el.get('id')] = dict(el.attrib)
        elif el.tag == 'trip':
            trips.append((el.get('id'), el.get('type', 'DEFAULT_VEHTYPE'), float(el.get('depart')),
                          el.get('from'), el.get('to'), None))
        elif el.tag == 'vehicle':
            r = el.find('route')
            edges = r.get('edges').split() if r is not None else None
            trips.append((el.get('id'), el.get('type', 'DEFAULT_VEHTYPE'), float(el.get('depart')),
                          edges[0] if edges else None, edges[-1] if edges else None, edges))
    return vtypes, trips


def parse_sumocfg(path):
    root = ET.parse(path).getroot()
    d = os.path.dirname(path)
    net = root.find('./input/net-file').get('value')
    rou = root.find('./input/route-files').get('value')
    begin = int(float(root.find('./time/begin').get('value')))
    end = int(float(root.find('./time/end').get('value')))
    return os.path.join(d, net), os.path.join(d, rou), begin, end


# --------------------------------------------------------------------------- signal plan
def green_phases(program):
    """multi_signal.py:52-59: phases without 'y' that contain a 'g'/'G'."""
    return [(d, s) for (d, s) in program if 'y' not in s and 'g' in s.lower()]


def build_yellow_program(greens, yellow_length):
    """Restatement of create_yellows (traffic_signal.py:7-24).

    Returns (phases [(duration, state)], yellow_idx dict "i_j" -> phase index).
    Green k keeps index k; yellows are appended in (i, j) scan order, duplicates kept.
    """
    phases = list(greens)
    yellow = {}
    for i, (_, si) in enumerate(greens):
        for j, (_, sj) in enumerate(greens):
            if i == j:
                continue
            need, chars = False, []
            for a, b in zip(si, sj):
                if a in 'Gg' and b in 'rs':
                    need = True
                    chars.append('y')
                else:
                    chars.append(a)
            if need:
                phases.append((yellow_length, ''.join(chars)))
                yellow['%d_%d' % (i, j)] = len(phases) - 1
    return phases, yellow


def program_index_at(program, t):
    """[SUMO-K] phase index / remaining time of a static program (offset 0) at time t."""
    cycle = sum(d for d, _ in program)
    r = t % cycle
    for k, (d, _) in enumerate(program):
        if r < d:
            return k, d - r
        r -= d
    return 0, program[0][0]


MOVEMENTS = ['S-W', 'S-S', 'S-E', 'W-N', 'W-W', 'W-S', 'N-E', 'N-N', 'N-W', 'E-S', 'E-E', 'E-N']


def controlled_links(net, sid):
    """trafficlight.getControlledLinks(sid): per link index of the programme the list of (from lane, to lane, via lane) it controls
    (traffic_signal.py:34, 117), from the net's <connection tl= linkIndex=> elements"""
    by = {}
    for c in net.conns:
        if c.tl == sid and c.link_index >= 0:
            by.setdefault(c.link_index, []).append(('%s_%d' % (c.frm, c.from_lane), '%s_%d' % (c.to, c.to_lane), c.via or ''))
    return [by.get(i, []) for i in range(max(by) + 1)] if by else []


def generate_signal_config(links, sid=''):
    """Restatement of Signal.generate_config (traffic_signal.py:106-170), the reference's fallback for a signal WITHOUT an entry in
    signal_configs[map]: made for the grid maps (three links per movement, lane ids that start with the upstream node's name).
    lanes = the from-lanes of the controlled links in link order; every third link opens one of the twelve movements, in the fixed order
    S-W .. E-N; downstream[d] = the leading `letters + digits` of the first lane of the straight movement that LEAVES towards d, unless
    that names the network's fringe.  lane_sets_outbound stays EMPTY (states that read it -- mplight, mplight_full, fma2c -- raise
    KeyError in the reference; here their outbound sums are simply empty)."""
    import re
    lanes = []
    lane_sets = {m: [] for m in MOVEMENTS}
    downstream = {'N': None, 'E': None, 'S': None, 'W': None}
    for i, link in enumerate(links):
        if not link:
            raise EnvironmentError('signal %s: link index %d controls nothing (generate_config reads link[0])' % (sid, i))
        frm = link[0][0]
        if frm not in lanes:
            lanes.append(frm)
        if i % 3 == 0:
            if i // 3 >= len(MOVEMENTS):
                raise EnvironmentError('signal %s has more than 36 controlled links: generate_config (traffic_signal.py:121-123) has '
                                       'twelve movements of three links' % sid)
            lane_sets[MOVEMENTS[i // 3]].append(frm)
    for mv, d in (('S-S', 'N'), ('N-N', 'S'), ('W-W', 'E'), ('E-E', 'W')):
        if not lane_sets[mv]:
            raise EnvironmentError('signal %s: generate_config needs a lane for movement %s (traffic_signal.py:134-160)' % (sid, mv))
        found = re.findall('[a-zA-Z]+[0-9]+', lane_sets[mv][0])
        if not found:
            raise EnvironmentError('signal %s: lane id %r does not start with a node name (generate_config is made for the grid maps)'
                                   % (sid, lane_sets[mv][0]))
        if not any(f in found[0] for f in ('top', 'right', 'left', 'bottom')):
            downstream[d] = found[0]
    return lanes, lane_sets, downstream


def derive_signal_lanes(sig_cfg_map, sid, links=None):
    """Restatement of Signal.__init__ lane bookkeeping (traffic_signal.py:49-87); for a signal without an entry in the map's
    signal_configs the reference's generate_config fallback over the controlled `links` (traffic_signal.py:88-89, 106-170).

    Returns dict(lanes, lane_sets, lane_sets_outbound, outbound_lanes, out_lane_to_signalid,
    inbounds_fr_direction, downstream).
    """
    rev = {'N': 'S', 'E': 'W', 'S': 'N', 'W': 'E'}
    if sid not in sig_cfg_map:
        lanes, lane_sets, downstream = generate_signal_config(links or [], sid)
        return dict(lanes=lanes, lane_sets=lane_sets, lane_sets_outbound={k: [] for k in lane_sets}, outbound_lanes=[],
                    out_lane_to_signalid={}, inbounds_fr_direction={}, downstream=downstream, generated=True)
    cfg = sig_cfg_map[sid]
    lane_sets = cfg['lane_sets']
    downstream = cfg['downstream']
    lanes, inb = [], {}
    for direction in lane_sets:
        for lane in lane_sets[direction]:
            fr = rev[direction.split('-')[0]]
            if fr in inb:
                if lane not in inb[fr]:
                    inb[fr].append(lane)
            else:
                inb[fr] = [lane]
            if lane not in lanes:
                lanes.append(lane)
    out_sets = {k: [] for k in lane_sets}
    outbound, out2sig = [], {}
    for direction in downstream:
        dwn = downstream[direction]
        if dwn is None:
            continue
        dsets = sig_cfg_map[dwn]['lane_sets']
        for key in dsets:
            if key.split('-')[0] != direction:
                continue
            dset = dsets[key]
            if dset is None:
                raise Exception('Invalid signal config')
            for lane in dset:
                if lane not in outbound:
                    outbound.append(lane)
                out2sig[lane] = dwn
                for selfkey in lane_sets:
                    if selfkey.split('-')[1] == key.split('-')[0]:
                        out_sets[selfkey] = out_sets[selfkey] + list(dset)
    for key in out_sets:     # de-duplicate (traffic_signal.py:86-87 uses set(); order is irrelevant to sums)
        seen, uniq = set(), []
        for lane in out_sets[key]:
            if lane not in seen:
                seen.add(lane)
                uniq.append(lane)
        out_sets[key] = uniq
    return dict(lanes=lanes, lane_sets=lane_sets, lane_sets_outbound=out_sets, outbound_lanes=outbound,
                out_lane_to_signalid=out2sig, inbounds_fr_direction=inb, downstream=downstream)


# --------------------------------------------------------------------------- scenario
_ARRAY_FIELDS = [
    # lanes
    'lane_len', 'lane_vmax', 'lane_edge', 'lane_left', 'lane_right', 'lane_link_start', 'lane_link_cnt',
    'lane_obs', 'lane_internal',
    # links
    'link_to_lane', 'link_dest_lane', 'link_to_edge', 'link_tls', 'link_tls_pos', 'link_minor', 'link_cont',
    'link_foe_start', 'link_foe_cnt', 'link_via_len', 'link_via1', 'link_via2', 'link_from_lane',
    'foe_link',
    # edges
    'edge_lane0', 'edge_nlanes',
    # routes
    'route_start', 'route_edge', 'route_tlsdist', 'route_cont',
    # demand
    'trip_depart', 'trip_route', 'trip_vtype', 'trips_cum', 'vtype_params',
    # signals (controlled, in all_ts_ids order)
    'tls_nphase', 'tls_ngreen', 'tls_nlinks', 'tls_state_off', 'tls_dur_off', 'tls_yel_off', 'tls_init_phase',
    'tls_states', 'tls_dur', 'tls_yellow',
    # fixed-time (original) programs for the FIXED plumbing baseline
    'fix_nphase', 'fix_state_off', 'fix_dur_off', 'fix_init_phase', 'fix_init_left', 'fix_states', 'fix_dur',
    # observation tables
    'obs_lane', 'sig_obs_start', 'mv_in_start', 'mv_in_idx', 'mv_out_start', 'mv_out_idx',
    'pr_out_start', 'pr_out_idx',
]


@dataclass
class Scenario:
    name: str
    begin: int
    end: int
    yellow_length: int
    capacity: int                        # vehicle ring slots per environment (power of two)
    arrays: dict = field(default_factory=dict)
    # names (host-side only)
    signal_ids: list = field(default_factory=list)
    lane_ids: list = field(default_factory=list)        # compact lane index -> SUMO lane id
    edge_ids: list = field(default_factory=list)
    obs_lane_ids: list = field(default_factory=list)    # global observed-lane index -> SUMO lane id
    trip_ids: list = field(default_factory=list)
    vtype_ids: list = field(default_factory=list)
    signal_meta: dict = field(default_factory=dict)     # sid -> derive_signal_lanes(...) + phases/yellow_dict
    phase_pairs: list = field(default_factory=list)
    valid_acts: dict | None = None
    demand_tag: str = 'trip'             # element name of the demand in the rou.xml: 'trip' (routed at load) or 'vehicle' (explicit
                                         # routes); utils/readXML.py:59-68 charges never-departed demand only for 'vehicle' files

    def __getattr__(self, k):
        arrays = self.__dict__.get('arrays')
        if arrays is not None and k in arrays:
            return arrays[k]
        raise AttributeError(k)

    # sizes
    @property
    def n_lanes(self): return len(self.arrays['lane_len'])
    @property
    def n_links(self): return len(self.arrays['link_to_lane'])
    @property
    def n_edges(self): return len(self.arrays['edge_lane0'])
    @property
    def n_routes(self): return len(self.arrays['route_start']) - 1
    @property
    def n_trips(self): return len(self.arrays['trip_depart'])
    @property
    def n_signals(self): return len(self.arrays['tls_nphase'])
    @property
    def n_obs(self): return len(self.arrays['obs_lane'])
    @property
    def kmax(self): return int(self.arrays['route_cont'].shape[1])
    @property
    def horizon(self): return self.end - self.begin

    def save(self, path):
        import json
        meta = dict(name=self.name, begin=self.begin, end=self.end, yellow_length=self.yellow_length,
                    capacity=self.capacity, signal_ids=self.signal_ids, lane_ids=self.lane_ids,
                    edge_ids=self.edge_ids, obs_lane_ids=self.obs_lane_ids, trip_ids=self.trip_ids,
                    vtype_ids=self.vtype_ids, signal_meta=self.signal_meta, phase_pairs=self.phase_pairs,
                    demand_tag=self.demand_tag,
                    valid_acts=None if self.valid_acts is None else
                    {k: [[int(a), int(b)] for a, b in v.items()] for k, v in self.valid_acts.items()})
        blob = np.frombuffer(json.dumps(meta).encode('utf-8'), dtype=np.uint8)
        np.savez_compressed(path, __meta__=blob, **self.arrays)

    @staticmethod
    def load(path) -> 'Scenario':
        import json
        z = np.load(path)
        meta = json.loads(bytes(z['__meta__']).decode('utf-8'))
        va = meta.pop('valid_acts')
        if va is not None:
            va = {k: {int(a): int(b) for a, b in v} for k, v in va.items()}
        arrays = {k: np.ascontiguousarray(z[k]) for k in z.files if k != '__meta__'}
        return Scenario(arrays=arrays, valid_acts=va, **meta)


CONT_BIG = np.float32(1.0e6)    # "continues to the end of the route"


def route_continuation(A):
    """route_cont[route step][k] (float32, [n_route_steps, kmax]): how far a vehicle that enters edge c of its route on
    lane k can drive along the route without changing lanes, measured from the start of the edge (the lane itself,
    plus the best continuation through the connections to the next route edge; CONT_BIG on the last edge, where every
    lane arrives).  This is SUMO's per-lane `length` of MSVehicle::getBestLanes [SUMO-K]; the simulator derives the
    strategic lane-change need and the choice between parallel connections from it."""
    kmax = int(A['edge_nlanes'].max())
    rs = A['route_start']
    n_steps = len(A['route_edge'])
    cont = np.zeros((n_steps, kmax), np.float32)
    l0, nl, llen = A['edge_lane0'], A['edge_nlanes'], A['lane_len']
    ls, lc = A['lane_link_start'], A['lane_link_cnt']
    for r in range(len(rs) - 1):
        a, b = int(rs[r]), int(rs[r + 1])
        for c in range(b - 1, a - 1, -1):
            e = int(A['route_edge'][c])
            for k in range(int(nl[e])):
                if c == b - 1:
                    cont[c, k] = CONT_BIG
                    continue
                lane = int(l0[e]) + k
                ne = int(A['route_edge'][c + 1])
                best = np.float32(0.0)
                for li in range(int(ls[lane]), int(ls[lane]) + int(lc[lane])):
                    if int(A['link_to_edge'][li]) != ne:
                        continue
                    v = np.float32(A['link_via_len'][li]) + cont[c + 1, int(A['link_dest_lane'][li]) - int(l0[ne])]
                    if v > best:
                        best = v
                cont[c, k] = min(CONT_BIG, np.float32(llen[lane]) + best)
    return np.ascontiguousarray(cont)


def _dijkstra(succ, cost, src, dst):
    dist = {src: cost[src]}
    prev = {}
    heap = [(cost[src], src)]
    done = set()
    while heap:
        d, u = heapq.heappop(heap)
        if u in done:
            continue
        done.add(u)
        if u == dst:
            break
        for v, w in succ.get(u, ()):
            nd = d + w + cost[v]
            if v not in dist or nd < dist[v]:
                dist[v] = nd
                prev[v] = u
                heapq.heappush(heap, (nd, v))
    if dst not in done:
        return None
    path = [dst]
    while path[-1] != src:
        path.append(prev[path[-1]])
    return path[::-1]


def compile_scenario(name, net: Net, vtypes_xml, trips_xml, begin, end, sig_cfg_map, lights=(),
                     yellow_length=3, capacity=None, minor_penalty=1.5, turnaround_penalty=0.0) -> Scenario:
    """Build the flat tables for one map.

    minor_penalty / turnaround_penalty (seconds): router cost terms next to edge length / speed + junction-lane time.  [SUMO-K]
    `--weights.minor-penalty` (default 1.5 s): every junction lane entered over a link that is neither traffic-light controlled nor
    has priority costs that much more (MSEdge::recalcCache) -- part of the shipped cost model since round 5: it re-routes 583 of
    ingolstadt21's 4283 trips and moves that map's FIXED / random-policy delays from 2.07 x / 1.27 x to 1.72 x / 1.05 x of the
    reference's (profiles/r05_route_sensitivity.txt, tools/route_sensitivity.py).  `--weights.turnaround-penalty` (later SUMO
    versions) moves 6 trips and nothing else: left at 0.

    sig_cfg_map = signal_configs[map] (the reference's per-map dict: phase_pairs, valid_acts, per-signal
    lane_sets/downstream; resco_benchmark/config/signal_config.py).
    lights      = map_configs[map]['lights'] (empty -> sorted tlLogic ids, multi_signal.py:64).
    """
    E, L = net.edges, net.lanes

    # ---- lane-level connection index
    conn_by_from = {}            # (edge, laneidx) -> [conn]
    for c in net.conns:
        conn_by_from.setdefault((c.frm, c.from_lane), []).append(c)

    def lane_of(edge_id, idx):
        for l in E[edge_id].lanes:
            if l.index == idx:
                return l
        return None

    def via_chain(c):
        """internal lanes between the from-lane and the destination lane of a normal->normal connection"""
        chain = []
        via = c.via
        guard = 0
        while via is not None and guard < 4:
            chain.append(via)
            vl = L[via]
            nxt = None
            for c2 in conn_by_from.get((vl.edge, vl.index), ()):
                nxt = c2
                break
            via = nxt.via if nxt is not None else None
            guard += 1
        return chain

    # normal->normal passenger connections
    nconns = []
    for c in net.conns:
        if c.frm not in E or c.to not in E or E[c.frm].internal or E[c.to].internal:
            continue
        fl, tl_ = lane_of(c.frm, c.from_lane), lane_of(c.to, c.to_lane)
        if fl is None or tl_ is None or not fl.passenger or not tl_.passenger:
            continue
        nconns.append(c)

    # ---- routing graph (edge level)
    def edge_speed(e):
        return max(l.speed for l in E[e].lanes if l.passenger)

    def edge_len(e):
        return E[e].lanes[0].length

    cost = {}
    for eid, e in E.items():
        if not e.internal and any(l.passenger for l in e.lanes):
            cost[eid] = edge_len(eid) / edge_speed(eid)
    succ = {}
    pair_w = {}
    def via_penalty(c):
        """[SUMO-K MSEdge::recalcCache] every junction lane whose incoming link is uncontrolled and minor costs `minor_penalty`"""
        pen = turnaround_penalty if c.dir == 't' else 0.0
        cc, guard = c, 0
        while cc is not None and cc.via is not None and guard < 4:
            if cc.tl is None and not ('A' <= cc.state[:1] <= 'Z'):
                pen += minor_penalty
            vl = L[cc.via]
            cc = next(iter(conn_by_from.get((vl.edge, vl.index), ())), None)
            guard += 1
        return pen

    for c in nconns:
        w = sum(L[v].length / max(L[v].speed, 0.1) for v in via_chain(c))
        if minor_penalty or turnaround_penalty:
            w += via_penalty(c)
        key = (c.frm, c.to)
        if key not in pair_w or w < pair_w[key]:
            pair_w[key] = w
    for (a, b), w in pair_w.items():
        succ.setdefault(a, []).append((b, w))

    # ---- demand: route every trip (distinct OD pairs share one route)
    trips_xml = sorted(trips_xml, key=lambda t: t[2])       # stable; files are depart-sorted already
    od_route, routes, route_key = {}, [], {}
    trip_route, trip_depart, trip_vt, trip_ids = [], [], [], []
    vtype_ids = []
    dropped = 0
    for (tid, vt, depart, frm, to, explicit) in trips_xml:
        if depart < begin:       # [SUMO-K] vehicles that departed before the simulation begin are not loaded
            dropped += 1
            continue
        if explicit is not None:
            path = [e for e in explicit if e in cost]
            key = tuple(path)
        else:
            if (frm, to) not in od_route:
                p = _dijkstra(succ, cost, frm, to) if (frm in cost and to in cost) else None
                od_route[(frm, to)] = tuple(p) if p else None
            key = od_route[(frm, to)]
        if not key:
            dropped += 1
            continue
        if key not in route_key:
            route_key[key] = len(routes)
            routes.append(list(key))
        if vt not in vtype_ids:
            vtype_ids.append(vt)
        trip_ids.append(tid)
        trip_route.append(route_key[key])
        trip_depart.append(max(0, int(math.ceil(depart - begin - 1e-9))))
        trip_vt.append(vtype_ids.index(vt))

    # ---- compact edges / lanes
    used_edges = set()
    used_pairs = set()
    for r in routes:
        used_edges.update(r)
        used_pairs.update(zip(r[:-1], r[1:]))
    edge_ids = [eid for eid in E if eid in used_edges]           # file order
    edge_index = {e: i for i, e in enumerate(edge_ids)}
    lane_ids, lane_index = [], {}
    edge_lane0, edge_nl = [], []
    for eid in edge_ids:
        pl = [l for l in E[eid].lanes if l.passenger]
        edge_lane0.append(len(lane_ids))
        edge_nl.append(len(pl))
        for l in pl:
            lane_index[l.id] = len(lane_ids)
            lane_ids.append(l.id)
    n_normal = len(lane_ids)
    used_conns = [c for c in nconns if (c.frm, c.to) in used_pairs]
    for c in used_conns:
        for v in via_chain(c):
            if v not in lane_index:
                lane_index[v] = len(lane_ids)
                lane_ids.append(v)
    nl = len(lane_ids)
    assert nl < 65000, 'lane ids must fit u16'

    lane_len = np.array([L[i].length for i in lane_ids], np.float32)
    lane_vmax = np.array([L[i].speed for i in lane_ids], np.float32)
    lane_edge = np.full(nl, -1, np.int32)
    lane_left = np.full(nl, -1, np.int32)
    lane_right = np.full(nl, -1, np.int32)
    lane_internal = np.zeros(nl, np.int32)
    lane_internal[n_normal:] = 1
    for ei, eid in enumerate(edge_ids):
        l0, n = edge_lane0[ei], edge_nl[ei]
        assert n <= 32
        for k in range(n):
            lane_edge[l0 + k] = ei
            if k + 1 < n:
                lane_left[l0 + k] = l0 + k + 1
            if k > 0:
                lane_right[l0 + k] = l0 + k - 1

    # ---- links
    # request index of a first-stage connection inside its junction
    def request_index(c, chain):
        j = net.junctions.get(E[c.frm].to)
        if j is None:
            return None, -1
        for v in reversed(chain):
            if v in j.int_lanes:
                return j, j.int_lanes.index(v)
        return j, -1

    links = []          # dict records
    lane_links = [[] for _ in range(nl)]
    first_stage_by_req = {}     # (junction id, request idx) -> [link ids]
    pending_foes = []           # (link id, junction, request idx)
    for c in used_conns:
        chain = via_chain(c)
        fl = lane_index[lane_of(c.frm, c.from_lane).id]
        dl = lane_index[lane_of(c.to, c.to_lane).id]
        via_ids = [lane_index[v] for v in chain]
        j, ridx = request_index(c, chain)
        cont = 0
        if j is not None and 0 <= ridx < len(j.requests):
            cont = j.requests[ridx][2]
        if len(via_ids) < 2:
            cont = 0
        minor = 1 if c.state in 'm=' else 0
        lid = len(links)
        links.append(dict(frm=fl, to=via_ids[0] if via_ids else dl, dest=dl, to_edge=edge_index[c.to],
                          tls=c.tl, pos=c.link_index, minor=minor, cont=cont,
                          via_len=float(sum(L[v].length for v in chain)),
                          via1=via_ids[0] if len(via_ids) > 0 else -1,
                          via2=via_ids[1] if len(via_ids) > 1 else -1, foes=[], stage=0))
        lane_links[fl].append(lid)
        if j is not None and ridx >= 0:
            first_stage_by_req.setdefault((j.id, ridx), []).append(lid)
            pending_foes.append((lid, j, ridx))
        # internal hops
        for k, v in enumerate(via_ids):
            nxt = via_ids[k + 1] if k + 1 < len(via_ids) else dl
            hid = len(links)
            # the hop that leaves the first internal lane of a 2-stage turn is the internal-junction
            # link (state 'm' in the net file): it inherits the prohibitors of the first stage
            links.append(dict(frm=v, to=nxt, dest=dl, to_edge=edge_index[c.to], tls=None, pos=-1,
                              minor=1 if (k == 0 and len(via_ids) > 1) else 0, cont=0,
                              via_len=float(sum(L[x].length for x in chain[k + 1:])),
                              via1=via_ids[k + 1] if k + 1 < len(via_ids) else -1, via2=-1, foes=[],
                              stage=1, parent=lid))
            if not lane_links[v]:
                lane_links[v].append(hid)
    # prohibitors from the junction's response matrix
    for lid, j, ridx in pending_foes:
        if ridx >= len(j.requests):
            continue
        resp = j.requests[ridx][0]
        n = len(resp)
        foes = []
        for b in range(n):
            if resp[n - 1 - b] == '1' and b != ridx:
                foes.extend(first_stage_by_req.get((j.id, b), ()))
        links[lid]['foes'] = foes
    for rec in links:
        if rec['stage'] == 1 and rec['minor']:
            rec['foes'] = links[rec['parent']]['foes']

    # ---- signals
    tl_ids = list(lights) if len(lights) > 0 else sorted(net.tllogics.keys())
    tl_index = {t: i for i, t in enumerate(tl_ids)}
    tls_nphase, tls_ngreen, tls_nlinks = [], [], []
    tls_state_off, tls_dur_off, tls_yel_off, tls_init = [], [], [], []
    tls_states, tls_dur, tls_yellow = [], [], []
    fix_nphase, fix_state_off, fix_dur_off, fix_init_phase, fix_init_left = [], [], [], [], []
    fix_states, fix_dur = [], []
    signal_meta = {}
    for sid in tl_ids:
        prog = net.tllogics[sid]
        greens = green_phases(prog)
        phases, ydict = build_yellow_program(greens, yellow_length)
        G, P, nlk = len(greens), len(phases), len(prog[0][1])
        init_idx, init_left = program_index_at(prog, begin)
        tls_nphase.append(P)
        tls_ngreen.append(G)
        tls_nlinks.append(nlk)
        tls_state_off.append(len(tls_states))
        tls_dur_off.append(len(tls_dur))
        tls_yel_off.append(len(tls_yellow))
        # [SUMO-K] the re-installed program starts at the original program's current index
        tls_init.append(init_idx if init_idx < P else 0)
        for d, s in phases:
            tls_dur.append(d)
            tls_states.extend(_TLS_CODE[ch] for ch in s)
        ytab = [-1] * (G * G)
        for k, v in ydict.items():
            i, jn = k.split('_')
            ytab[int(i) * G + int(jn)] = v
        tls_yellow.extend(ytab)
        fix_nphase.append(len(prog))
        fix_state_off.append(len(fix_states))
        fix_dur_off.append(len(fix_dur))
        fix_init_phase.append(init_idx)
        fix_init_left.append(init_left)
        for d, s in prog:
            fix_dur.append(d)
            fix_states.extend(_TLS_CODE[ch] for ch in s)
        signal_meta[sid] = dict(phases=[[d, s] for d, s in phases], yellow_dict=ydict, n_green=G,
                                green_durations=[d for d, _ in greens], orig_program=[[d, s] for d, s in prog])

    # ---- flatten links
    nk = len(links)
    A = {}
    A['link_to_lane'] = np.array([r['to'] for r in links], np.int32)
    A['link_dest_lane'] = np.array([r['dest'] for r in links], np.int32)
    A['link_to_edge'] = np.array([r['to_edge'] for r in links], np.int32)
    A['link_tls'] = np.array([tl_index.get(r['tls'], -1) if r['tls'] else -1 for r in links], np.int32)
    A['link_tls_pos'] = np.array([r['pos'] if (r['tls'] in tl_index) else -1 for r in links], np.int32)
    A['link_minor'] = np.array([r['minor'] for r in links], np.int32)
    A['link_cont'] = np.array([r['cont'] for r in links], np.int32)
    A['link_via_len'] = np.array([r['via_len'] for r in links], np.float32)
    A['link_via1'] = np.array([r['via1'] for r in links], np.int32)
    A['link_via2'] = np.array([r['via2'] for r in links], np.int32)
    A['link_from_lane'] = np.array([r['frm'] for r in links], np.int32)
    fs, fc, fl_ = [], [], []
    for r in links:
        fs.append(len(fl_))
        fc.append(len(r['foes']))
        fl_.extend(r['foes'])
    A['link_foe_start'] = np.array(fs, np.int32)
    A['link_foe_cnt'] = np.array(fc, np.int32)
    A['foe_link'] = np.array(fl_ if fl_ else [0], np.int32)
    # lane -> links CSR (links re-ordered so each lane's links are contiguous)
    order, lstart, lcnt = [], [], []
    for lane in range(nl):
        lstart.append(len(order))
        lcnt.append(len(lane_links[lane]))
        order.extend(lane_links[lane])
    # links not attached (duplicate internal hops) are dropped by the permutation
    perm = np.array(order, np.int64)
    inv = np.full(nk, -1, np.int64)
    inv[perm] = np.arange(len(perm))
    for k in ('link_to_lane', 'link_dest_lane', 'link_to_edge', 'link_tls', 'link_tls_pos', 'link_minor',
              'link_cont', 'link_via_len', 'link_via1', 'link_via2', 'link_from_lane', 'link_foe_start',
              'link_foe_cnt'):
        A[k] = np.ascontiguousarray(A[k][perm])
    A['foe_link'] = np.array([inv[f] for f in A['foe_link']], np.int32) if fl_ else A['foe_link']
    assert (A['foe_link'] >= 0).all()
    A['lane_link_start'] = np.array(lstart, np.int32)
    A['lane_link_cnt'] = np.array(lcnt, np.int32)

    # ---- routes
    conn_lanes = {}         # (from edge idx, to edge idx) -> [(from lane k, to lane k)]
    for c in used_conns:
        fe, te = edge_index[c.frm], edge_index[c.to]
        fk = lane_index[lane_of(c.frm, c.from_lane).id] - edge_lane0[fe]
        tk = lane_index[lane_of(c.to, c.to_lane).id] - edge_lane0[te]
        conn_lanes.setdefault((fe, te), []).append((fk, tk, c))
    route_start, route_edge, route_tls = [0], [], []
    for r in routes:
        ids = [edge_index[e] for e in r]
        n = len(ids)
        # distance from the end of edge c to the next TLS stop line along the route
        td = [float(BIG)] * n
        nxt = float(BIG)
        for c_ in range(n - 2, -1, -1):
            cl = conn_lanes.get((ids[c_], ids[c_ + 1]), ())
            if not cl:
                td[c_] = float(BIG)
                nxt = td[c_]
                continue
            cc = cl[0][2]
            if cc.tl is not None:
                td[c_] = 0.0
            else:
                via = sum(L[v].length for v in via_chain(cc))
                ahead = td[c_ + 1]
                td[c_] = float(BIG) if ahead >= float(BIG) else via + edge_len(r[c_ + 1]) + ahead
        route_edge.extend(ids)
        route_tls.extend(td)
        route_start.append(len(route_edge))
    A['route_start'] = np.array(route_start, np.int32)
    A['route_edge'] = np.array(route_edge, np.int32)
    A['route_tlsdist'] = np.array(route_tls, np.float32)

    # ---- demand tables
    horizon = end - begin
    A['trip_depart'] = np.array(trip_depart, np.int32)
    A['trip_route'] = np.array(trip_route, np.int32)
    A['trip_vtype'] = np.array(trip_vt, np.int32)
    cum = np.searchsorted(A['trip_depart'], np.arange(horizon + 2), side='right').astype(np.int32)
    A['trips_cum'] = cum            # trips_cum[t] = #trips with depart_tick <= t
    vt = np.zeros((max(1, len(vtype_ids)), VT_COLS), np.float32)
    for i, v in enumerate(vtype_ids):
        attrs = vtypes_xml.get(v, {})
        d = dict(_VCLASS_DEFAULTS.get(attrs.get('vClass', 'passenger'), _VCLASS_DEFAULTS['passenger']))
        for k in list(d):
            if k in attrs:
                d[k] = float(attrs[k])
        vt[i] = [d['length'], d['minGap'], d['accel'], d['decel'], d['tau'], d['sigma'], d['maxSpeed'],
                 d['speedFactor'], d['speedDev'], d['emergencyDecel']]
    A['vtype_params'] = vt

    # ---- observation tables
    obs_lane_ids, obs_lane, sig_obs_start = [], [], [0]
    obs_index = {}
    missing_obs_lanes = []
    for sid in tl_ids:
        signal_meta[sid]['controlled_links'] = [[list(t) for t in lk] for lk in controlled_links(net, sid)]
        meta = derive_signal_lanes(sig_cfg_map, sid, signal_meta[sid]['controlled_links'])
        signal_meta[sid].update(meta)
        for lane in meta['lanes']:
            if lane in obs_index:
                # one detector row per lane: the reference gives every signal its own view of a shared lane
                # (traffic_signal.py:189-247); no packaged map has one, a recompiled one must not get it silently wrong
                raise ValueError('lane %s is observed by more than one signal (%s): not supported' % (lane, sid))
            obs_index[lane] = len(obs_lane_ids)
            obs_lane_ids.append(lane)
            if lane not in lane_index:
                missing_obs_lanes.append(lane)
            obs_lane.append(lane_index.get(lane, -1))
        sig_obs_start.append(len(obs_lane_ids))
    if missing_obs_lanes:
        # signal_configs names lanes the net does not have (cologne8: '-24487264_0', '-22959475#4_0'): the reference's
        # getLastStepVehicleIDs would raise on them only when called; their rows stay zero here
        import warnings
        warnings.warn('%s: observed lanes missing from the net (their state rows stay zero): %s' % (name, missing_obs_lanes))
    lane_obs = np.full(nl, -1, np.int32)
    for gi, li in enumerate(obs_lane):
        if li >= 0:
            lane_obs[li] = gi
    mv_in_start, mv_in_idx, mv_out_start, mv_out_idx = [0], [], [0], []
    pr_out_start, pr_out_idx = [0], []
    for sid in tl_ids:
        meta = signal_meta[sid]
        keys = list(meta['lane_sets'].keys())
        assert len(keys) == MOVEMENT_SLOTS, (sid, keys)
        for key in keys:
            mv_in_idx.extend(obs_index[l] for l in meta['lane_sets'][key])
            mv_in_start.append(len(mv_in_idx))
            for l in meta['lane_sets_outbound'][key]:
                if meta['out_lane_to_signalid'][l] in tl_index:       # states.py:75 "if dwn_signal in signals"
                    mv_out_idx.append(obs_index[l])
            mv_out_start.append(len(mv_out_idx))
        for l in meta['outbound_lanes']:
            if meta['out_lane_to_signalid'][l] in tl_index:           # rewards.py:37
                pr_out_idx.append(obs_index[l])
        pr_out_start.append(len(pr_out_idx))

    A.update(lane_len=lane_len, lane_vmax=lane_vmax, lane_edge=lane_edge, lane_left=lane_left,
             lane_right=lane_right, lane_obs=lane_obs, lane_internal=lane_internal,
             edge_lane0=np.array(edge_lane0, np.int32), edge_nlanes=np.array(edge_nl, np.int32),
             tls_nphase=np.array(tls_nphase, np.int32), tls_ngreen=np.array(tls_ngreen, np.int32),
             tls_nlinks=np.array(tls_nlinks, np.int32), tls_state_off=np.array(tls_state_off, np.int32),
             tls_dur_off=np.array(tls_dur_off, np.int32), tls_yel_off=np.array(tls_yel_off, np.int32),
             tls_init_phase=np.array(tls_init, np.int32), tls_states=np.array(tls_states, np.int32),
             tls_dur=np.array(tls_dur, np.int32), tls_yellow=np.array(tls_yellow, np.int32),
             fix_nphase=np.array(fix_nphase, np.int32), fix_state_off=np.array(fix_state_off, np.int32),
             fix_dur_off=np.array(fix_dur_off, np.int32), fix_init_phase=np.array(fix_init_phase, np.int32),
             fix_init_left=np.array(fix_init_left, np.int32), fix_states=np.array(fix_states, np.int32),
             fix_dur=np.array(fix_dur, np.int32),
             obs_lane=np.array(obs_lane, np.int32), sig_obs_start=np.array(sig_obs_start, np.int32),
             mv_in_start=np.array(mv_in_start, np.int32), mv_in_idx=np.array(mv_in_idx or [0], np.int32),
             mv_out_start=np.array(mv_out_start, np.int32), mv_out_idx=np.array(mv_out_idx or [0], np.int32),
             pr_out_start=np.array(pr_out_start, np.int32), pr_out_idx=np.array(pr_out_idx or [0], np.int32))
    A['route_cont'] = route_continuation(A)
    for k in _ARRAY_FIELDS:
        assert k in A, k
        A[k] = np.ascontiguousarray(A[k])

    if capacity is None:
        # vehicle slots per environment: next power of two above ~4x the free-flow concurrency estimate
        # (trips wait for a free slot when an environment is full; that shows up as departDelay)
        dur = []
        for r in routes:
            dur.append(sum(cost[e] for e in r))
        mean_dur = float(np.mean([dur[i] for i in trip_route])) if trip_route else 1.0
        conc = len(trip_route) * mean_dur / max(1, horizon)
        capacity = 64
        while capacity < 4 * conc + 32:
            capacity *= 2
        capacity = min(capacity, 4096)

    sc = Scenario(name=name, begin=begin, end=end, yellow_length=yellow_length, capacity=int(capacity),
                  arrays=A, signal_ids=list(tl_ids), lane_ids=lane_ids, edge_ids=edge_ids,
                  obs_lane_ids=obs_lane_ids, trip_ids=trip_ids, vtype_ids=vtype_ids,
                  signal_meta=signal_meta, phase_pairs=[list(p) for p in sig_cfg_map.get('phase_pairs', [])],
                  valid_acts=sig_cfg_map.get('valid_acts'),
                  demand_tag='vehicle' if any(t[5] is not None for t in trips_xml) else 'trip')
    sc.dropped_trips = dropped
    sc.router = dict(cost=cost, succ=succ, od_route=od_route)      # host-side only (not saved): tools/route_sensitivity.py
    return sc


def compile_from_sumocfg(name, sumocfg_path, sig_cfg_map, lights=(), yellow_length=3, capacity=None, **router):
    net_path, rou_path, begin, end = parse_sumocfg(sumocfg_path)
    net = parse_net(net_path)
    vtypes, trips = parse_routes(rou_path)
    return compile_scenario(name, net, vtypes, trips, begin, end, sig_cfg_map, lights=lights,
                            yellow_length=yellow_length, capacity=capacity, **router)
