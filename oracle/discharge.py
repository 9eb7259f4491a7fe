"""Queue-discharge harness (TEST INFRASTRUCTURE, like everything under oracle/): a standing queue at a stop line, released by
a green of G seconds, in the CPU oracle -- next to an INDEPENDENT float64 restatement of the same situation written from the
formulas of SURVEY.md section 8(a) alone (Krauss, Euler update, dt = 1 s: `brakeGap`, `maximumSafeStopSpeedEuler`,
`maximumSafeFollowSpeed`, `finalizeSpeed`; the stop rule at red / yellow `seen >= brakeGap(v)`; 1 m kept to the stop line).

What it pins: the multi-vehicle start-up sequence of the oracle (and, through the bit-exact HIP == oracle tests, of the kernel) is
the published Krauss start-up sequence second by second -- which vehicle crosses the stop line in which second of a 7 / 17 / 27 s
green, what the last vehicle does when the yellow comes on.  What it cannot pin: that SUMO's binary does the same (PARITY
UNPINNED); the restatement is [SUMO-K] like the oracle.  tools/discharge_study.py prints the tables
(`profiles/r05_discharge_study.txt`), tests/test_discharge.py asserts them.

The scenario is cologne1's own net (resco_amd/scenarios/cologne1.npz) with the demand replaced: n passenger cars (`pkw`: length 4.3,
minGap 1.5, accel 2.6, decel 4.5, tau 1) on ONE route whose approach lane is the only lane that continues it, so nobody changes
lanes: W approach `-32038056#3_0` (351 m, 13.89 m/s), right turn (link 0, `G` in the W-E phase) over a 10.9 m junction lane
(16.66 m/s) into `32038051#0` (89 m, 19.44 m/s), where the trip ends.
"""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# phases of cologne1's installed programme (green k = index k; yellows follow in (i, j) scan order, traffic_signal.py:7-24)
PH_NS, PH_WE = 0, 2
YEL_NS_WE, YEL_WE_NS = 5, 9         # yellow_dict['0_2'], yellow_dict['2_0']

APPROACH, TARGET = '-32038056#3', '32038051#0'


# ------------------------------------------------------------------ the independent restatement (float64, pure Python)
def brake_gap(v, b):
    k = int(v / b)
    return k * v - b * k * (k + 1) / 2.0


def stop_speed(gap, b, tau):
    g = gap - 0.001
    if g < 0:
        return 0.0
    n = math.floor(0.5 - (tau - 0.5 * math.sqrt(1.0 + 4.0 * ((2.0 * g / b - tau) + tau * tau))))
    h = 0.5 * n * (n - 1) * b + n * b * tau
    return n * b + (g - h) / (n + tau)


def follow_speed(gap, vl, b, bl, tau):
    return stop_speed(gap + brake_gap(vl, max(b, bl)), b, tau)


class Road:
    """consecutive lanes of one path: [(length, speed limit)], the stop line at the end of lane 0; the trip ends at the end of the last"""

    def __init__(self, lanes):
        self.lanes = lanes
        self.starts = np.concatenate([[0.0], np.cumsum([l for l, _ in lanes])])
        self.stop_line = lanes[0][0]
        self.end = float(self.starts[-1])

    def vmax_at(self, x):
        for i, (l, v) in enumerate(self.lanes):
            if x <= self.starts[i + 1] or i == len(self.lanes) - 1:     # a front exactly at the lane end is still on the lane
                return v
        return self.lanes[-1][1]


def reference_discharge(road, x0, v0, signal, ticks, length=4.3, mingap=1.5, a=2.6, b=4.5, tau=1.0, emergency=9.0,
                        stop_offset=1.0, dawdle=None):
    """x0 / v0: front positions and speeds at tick 0, in driving order (leader first).  signal(t) -> 'G' | 'y' | 'r' for the
    tick that starts at t.  dawdle(i, t) -> sigma * U[0,1) of vehicle i in tick t (None: sigma 0).  Returns (cross, traj):
    cross[i] = the tick at whose end vehicle i's front is beyond the stop line (None: it never is), traj[t] = positions."""
    x = list(map(float, x0))
    v = list(map(float, v0))
    n = len(x)
    alive = [True] * n
    cross = [None] * n
    traj = []
    for t in range(ticks):
        st = signal(t)
        vn = [0.0] * n
        for i in range(n):
            if not alive[i]:
                continue
            vfree = min(v[i] + a, road.vmax_at(x[i]))
            vsafe = 1e30
            j = i - 1
            while j >= 0 and not alive[j]:
                j -= 1
            look = brake_gap(vfree, b) + vfree * tau + mingap + 1.0
            near = j >= 0 and x[j] - x[i] <= look + length           # a leader further away than the look-ahead plays no role
            on_approach = x[i] <= road.stop_line
            stopped_by_signal = False
            if near and (not on_approach or x[j] <= road.stop_line):   # leader on my own lane (or both of us beyond the line)
                vsafe = follow_speed(x[j] - length - x[i] - mingap, v[j], b, b, tau)
            else:
                seen = road.stop_line - x[i]
                if on_approach and seen < look and st in 'ry' and seen >= brake_gap(v[i], b):
                    vsafe = stop_speed(max(0.0, seen - stop_offset), b, tau)
                    stopped_by_signal = True
                if near and not stopped_by_signal and (not on_approach or seen < look):
                    vsafe = follow_speed(x[j] - length - x[i] - mingap, v[j], b, b, tau)
            vmin_n = max(0.0, v[i] - b)
            vmin_e = max(0.0, v[i] - emergency)
            vmin = min(vmin_n, max(vsafe, vmin_e))
            vmax = max(min(vfree, vsafe), vmin)
            vd = vmax
            if dawdle is not None:
                r = dawdle(i, t)
                vd -= r * (vd if vd < a else a)
                vd = max(0.0, vd)
            vn[i] = max(vd, vmin)
        for i in range(n):
            if not alive[i]:
                continue
            v[i] = vn[i]
            x[i] += vn[i]
            if cross[i] is None and x[i] > road.stop_line:
                cross[i] = t
            if x[i] > road.end:
                alive[i] = False
        traj.append([x[i] if alive[i] else None for i in range(n)])
    return cross, traj


# ------------------------------------------------------------------ the same situation in the oracle
def queue_scenario(n, headway=2, name='cologne1', approach=APPROACH, target=TARGET):
    """cologne1 with its demand replaced by n cars on the route approach -> target, one departing every `headway` seconds"""
    from resco_amd.scenario import Scenario
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    A = dict(sc.arrays)
    e0, e1 = sc.edge_ids.index(approach), sc.edge_ids.index(target)
    rs = A['route_start']
    route = [r for r in range(sc.n_routes) if list(A['route_edge'][rs[r]:rs[r + 1]]) == [e0, e1]]
    assert len(route) == 1, route
    A['trip_depart'] = (np.arange(n) * headway).astype(np.int32)
    A['trip_route'] = np.full(n, route[0], np.int32)
    A['trip_vtype'] = np.zeros(n, np.int32)
    A['trips_cum'] = np.searchsorted(A['trip_depart'], np.arange(sc.horizon + 2), side='right').astype(np.int32)
    sc.arrays = A
    sc.trip_ids = ['q%d' % i for i in range(n)]
    return sc


def road_of(sc, approach=APPROACH, target=TARGET):
    A = sc.arrays
    e0, e1 = sc.edge_ids.index(approach), sc.edge_ids.index(target)
    l0 = int(A['edge_lane0'][e0])
    link = [l for l in range(A['lane_link_start'][l0], A['lane_link_start'][l0] + A['lane_link_cnt'][l0])
            if A['link_to_edge'][l] == e1][0]
    lanes, lane_idx = [(float(A['lane_len'][l0]), float(A['lane_vmax'][l0]))], [l0]
    nl = int(A['link_to_lane'][link])
    while A['lane_internal'][nl]:
        lanes.append((float(A['lane_len'][nl]), float(A['lane_vmax'][nl])))
        lane_idx.append(nl)
        nl = int(A['link_to_lane'][A['lane_link_start'][nl]])
    lanes.append((float(A['lane_len'][nl]), float(A['lane_vmax'][nl])))
    road = Road(lanes)
    offset = {l: float(road.starts[i]) for i, l in enumerate(lane_idx)}
    for k in range(int(A['edge_nlanes'][e1])):         # a car may change lanes on the last edge (speed gain): same place on the path
        offset[int(A['edge_lane0'][e1]) + k] = float(road.starts[len(lane_idx)])
    return road, offset


def oracle_discharge(n, green, sigma=0.0, seed=0, fill=None, after=40, yellow=3, headway=2, env_index=0):
    """Fill the approach under red, switch as the reference's step does (3 s yellow of the cross phase, `green` s of green, 3 s yellow,
    red) and record.  Returns dict(cross = tick (0 = first green tick) at whose end vehicle i is beyond the stop line or None,
    x0 / v0 = the standing queue at green onset, traj = positions per tick along the path (None once arrived), signal = state per tick,
    road, sc, dawdle = the sigma * U[0,1) the oracle drew per (vehicle, tick))."""
    from oracle.pyoracle import OracleEnv, lib
    sc = queue_scenario(n, headway)
    road, lane_idx = road_of(sc)
    env = OracleEnv(sc, env_index=env_index, seed=seed, sigma=sigma, speed_dev=0, trip_log=0)
    fill = fill if fill is not None else n * headway + 60
    t_abs = [0]

    def run(phase, ticks, rec=None):
        for _ in range(ticks):
            env.set_phase(0, phase)         # restarts the phase: it never expires on its own while we hold it
            env.tick()
            t_abs[0] += 1
            if rec is not None:
                v = env.vehicles()
                pos = [None] * n
                for s in range(v['hw']):
                    k = int(v['trip'][s])
                    if k < 0 or v['lane'][s] >= 0xFFFE:
                        continue
                    pos[k] = lane_idx[int(v['lane'][s])] + float(v['pos'][s])
                rec.append(pos)

    run(PH_NS, fill)
    run(YEL_NS_WE, yellow)
    v = env.vehicles()
    order = sorted([s for s in range(v['hw']) if v['trip'][s] >= 0], key=lambda s: int(v['trip'][s]))
    assert [int(v['trip'][s]) for s in order] == list(range(n)), 'not every car got onto the approach: lengthen fill'
    x0 = [float(v['pos'][s]) for s in order]
    v0 = [float(v['speed'][s]) for s in order]
    t_green = t_abs[0]
    traj = []
    run(PH_WE, green, traj)
    run(YEL_WE_NS, yellow, traj)
    run(PH_NS, after, traj)
    sig = ['G'] * green + ['y'] * yellow + ['r'] * after
    cross = [None] * n
    for t, pos in enumerate(traj):
        for i in range(n):
            gone = pos[i] is None or pos[i] > road.stop_line
            if cross[i] is None and gone:
                cross[i] = t
    sg = sigma
    L = lib()

    def dawdle(i, t):
        return sg * (L.orc_hash(seed, env_index, i, t_green + t, 0) >> 8) / 16777216.0
    env.close()
    return dict(cross=cross, x0=x0, v0=v0, traj=traj, signal=sig, road=road, sc=sc, dawdle=dawdle if sigma > 0 else None)


def compare(n, green, sigma=0.0, seed=0):
    """oracle vs restatement from the oracle's standing queue at green onset: (oracle record, reference crossings, largest
    position difference in metres over every vehicle and tick while the car is on the approach or on the junction lane -- on the
    two-lane edge behind the junction the oracle's cars may change lanes for speed gain, which the one-lane restatement does not
    know)"""
    o = oracle_discharge(n, green, sigma=sigma, seed=seed)
    sig = o['signal']
    cross, traj = reference_discharge(o['road'], o['x0'], o['v0'], lambda t: sig[t], len(sig), dawdle=o['dawdle'])
    limit = float(o['road'].starts[-2])
    worst = 0.0
    for t in range(len(sig)):
        for i in range(n):
            a, b = o['traj'][t][i], traj[t][i]
            if a is not None and b is not None and a <= limit and b <= limit:
                worst = max(worst, abs(a - b))
            elif (a is None or a > limit) != (b is None or b > limit):
                worst = max(worst, abs((a if a is not None else limit) - (b if b is not None else limit)))
    return o, cross, worst
