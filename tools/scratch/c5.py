import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
from tools.pipes_ab import group_run
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', 'ingolstadt21.npz'))
n, k, agent = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
r = group_run(sc, n, k, 40, agent)
print('%s prio=%s %d x %d: %.0f env-steps/s' % (agent, os.environ.get('RESCO_POLICY_PRIORITY'), n, k, r['env_steps_per_s']), flush=True)
