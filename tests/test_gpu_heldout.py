"""Held-out fidelity check of the microsimulation model (gpu): the reference's published IDQN learning curves
(resco_benchmark/utils/avg_timeLoss.py, rows 'IDQN <map>', reduced to tests/golden/ref_bands.json:trained_best = the best of its
100 episodes) are figures NO constant of include/resco_model.h was calibrated on -- the model constants were tuned, in rounds
2-3, on the static controllers' delays only.  Here the in-repo IDQN (resco_amd/agents: the reference's network of
agents/pfrl_dqn.py:24-40 and its hyper-parameters of config/agent_config.py:83-94, batched over lock-step environments) is trained
from scratch on this package's simulator, with three learner seeds per map; the median of their best training episodes must land
within +-35 % of the reference's best episode and no seed above 1.35 x.
"Existing agents plug in unchanged" means just that: the learner sees states.drq_norm / rewards.wait_norm of this model and
converges to the delays it reached on SUMO."""
import json
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# (environments, episodes): the reference trains 100 episodes of ONE environment (36 000 agent steps, epsilon reaching 0 after 80);
# 256 lock-step environments see that much experience in a fraction of the episodes
RUNS = {'cologne1': (256, 30), 'ingolstadt1': (256, 30), 'cologne8': (256, 40), 'ingolstadt21': (256, 60)}
SEEDS = (0, 1, 2)       # learner seeds: network initialisation, exploration draws, replay sampling, demand seeds


@pytest.mark.timeout(900, method='thread')
@pytest.mark.parametrize('name', ['cologne1', 'ingolstadt1', 'cologne8', 'ingolstadt21'])
def test_trained_idqn_reaches_the_reference_s_trained_delay(name):
    """A DISTRIBUTION, not a seed (round 4 asserted on one learner seed, which happened to be the most favourable of four on
    ingolstadt21): three learner seeds per map; the MEDIAN of their best training episodes must lie within +-35 % of the
    reference's best episode (utils/avg_timeLoss.py rows 'IDQN <map>') and EVERY seed at or below 1.35 x; the mean of a seed's last
    five episodes (no minimum involved) is reported next to it."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import idqn_train
    import numpy as np
    with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
        ref = json.load(f)[name]
    envs, episodes = RUNS[name]
    target = ref['trained_best']['delay']
    ratios = []
    for seed in SEEDS:
        # replay ring of 10 000 env-steps = the 27.8 episodes of history the reference's ReplayBuffer(10000) holds
        rows, final = idqn_train.main(name, envs, episodes, 256, 1, True, 10000, 0.0, evaluate=False, quiet=True, seed=seed)
        best = final['best_training_episode_delay_s']
        last5 = float(np.mean([r['avg_delay_s'] for r in rows[-5:]]))
        ratios.append(best / target)
        print('heldout %-13s seed %d  IDQN best episode %6.1f s / reference best episode %.1f s = %.2f   last five episodes %6.1f s (%.2f)   '
              'random policy %.1f s   curve %s' % (name, seed, best, target, best / target, last5, last5 / target, final['random_avg_delay_s'],
                                                  [round(r['avg_delay_s']) for r in rows][::3]))
        assert best < 0.6 * final['random_avg_delay_s']        # it did learn: far below the random policy of the same demand
    med = float(np.median(ratios))
    print('heldout %-13s median of %d seeds %.2f   (min %.2f, max %.2f)' % (name, len(SEEDS), med, min(ratios), max(ratios)))
    assert 0.65 <= med <= 1.35, (name, ratios)
    assert max(ratios) <= 1.35, (name, ratios)
