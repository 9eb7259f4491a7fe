"""CPU: queue-discharge known answers (oracle/discharge.py).  A standing queue of passenger cars at a stop line, released by a green
of 7 / 17 / 27 s after the 3 s yellow of the cross phase: the oracle's start-up sequence must be the one an independent float64
restatement of the published Krauss formulas gives, vehicle by vehicle and second by second -- and through
the bit-exact HIP == oracle tests it is the kernel's too.  [SUMO-K]: that SUMO's binary does the same is not pinned."""
import pytest

from oracle import discharge as D


@pytest.mark.parametrize('green', [7, 17, 27])
def test_start_up_sequence_is_the_krauss_closed_form(green):
    o, ref_cross, worst = D.compare(20, green, sigma=0.0)
    assert o['cross'] == ref_cross
    assert worst < 5e-3                      # metres, over every vehicle and tick (fp32 oracle vs float64 restatement)
    # the leader stands 1 m before the stop line, the others at length + minGap = 5.8 m spacing
    assert abs(o['x0'][0] - (o['road'].stop_line - 1.0)) < 2e-3
    assert all(abs((a - b) - 5.8) < 1e-2 for a, b in zip(o['x0'], o['x0'][1:]))


def test_known_crossing_seconds_sigma_zero():
    """accel 2.6, decel 4.5, tau 1, length 4.3, minGap 1.5: vehicle k crosses in second ... of the green (0 = its first tick)."""
    o, _, _ = D.compare(20, 27)
    assert o['cross'][:19] == [0, 2, 4, 6, 7, 9, 10, 12, 13, 15, 16, 17, 19, 20, 22, 23, 24, 26, 27]
    assert o['cross'][19] is None            # the 20th car reaches the line under yellow, can stop, and does
    # a 7 s green passes four cars, a 17 s green twelve: 1.75 s / 1.42 s per car
    assert [c for c in D.compare(20, 7)[0]['cross'] if c is not None] == [0, 2, 4, 6]
    assert len([c for c in D.compare(20, 17)[0]['cross'] if c is not None]) == 12


def test_last_vehicle_at_yellow_onset():
    """a car that can still stop (distance to the line >= its brake gap) stops at yellow; one that cannot drives on."""
    for green in (7, 17, 27):
        o, _, _ = D.compare(20, green)
        sig, traj, line = o['signal'], o['traj'], o['road'].stop_line
        t_y = sig.index('y')
        for i, c in enumerate(o['cross']):
            if c is not None and c >= t_y:
                # crossed under yellow / red: it was too close to stop when the yellow came on
                x_prev, x_prev2 = traj[t_y - 1][i], traj[t_y - 2][i]
                v = x_prev - x_prev2
                assert line - x_prev < D.brake_gap(v, 4.5) + 1e-3, (green, i)
        stopped = [i for i, c in enumerate(o['cross']) if c is None]
        assert stopped and abs(traj[-1][stopped[0]] - (line - 1.0)) < 0.05      # the first one left behind waits 1 m before the line


@pytest.mark.parametrize('seed', [0, 3])
def test_start_up_with_driver_imperfection(seed):
    """sigma 0.5 with the oracle's own random draws fed to the restatement: same crossing seconds, same positions"""
    o, ref_cross, worst = D.compare(20, 17, sigma=0.5, seed=seed)
    assert o['cross'] == ref_cross
    assert worst < 5e-3
    assert 8 <= len([c for c in o['cross'] if c is not None]) <= 11      # 12 at sigma 0
