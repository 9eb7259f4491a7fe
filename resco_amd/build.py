"""Build the HIP shared library in-tree (resco_amd/csrc/libresco_sim.so) for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'resco_sim.hip')
LIB = os.path.join(HERE, 'csrc', 'libresco_sim.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: fp32 results must equal the CPU oracle bit-for-bit (no FMA fusion)
# -disable-machine-licm: the step kernel's tick loop is long and register-starved (80 VGPRs for three workgroups per CU); with
#   machine LICM every 32-bit literal of the loop body is hoisted into a VGPR of its own and five values end up in scratch,
#   written per thread and tick: 110 MB of HBM writes per launch at 4096 environments and 2.5 % of the time (profiles/r03_*)
# -fno-honor-nans: no NaN exists in the model; with the flag the `x < y ? x : y` selects become v_min_f32 / v_max_f32 instead of v_cmp +
#   v_cndmask pairs (and the s_nop the pair needs on gfx950): +0.6-0.9 % (profiles/r06_ab_linkrec_nnan.txt); results bit-identical
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-honor-nans', '-mllvm', '-disable-machine-licm', '-fPIC', '-shared',
         '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(HERE, 'csrc')]


def build_library(force=False, verbose=False):
    deps = [SRC] + [os.path.join(HERE, 'csrc', f) for f in ('resco_step.h', 'resco_tables.h', 'resco_policy.h')] + \
           [os.path.join(ROOT, 'include', f) for f in ('resco_sim.h', 'resco_model.h')]
    force = force or os.environ.get('GRAFT_FORCE_BUILD') == '1'
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = [HIPCC] + FLAGS + [SRC, '-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


CHECK_LIB = os.path.join(HERE, 'csrc', 'libresco_sim_check.so')


def build_check_library(force=False):
    """The CHECKING build: the step kernel's classification invariants (RS_ASSERT) counted on the device instead of compiled out
    (rs_stats()[11]); one capacity (256 slots: cologne8, ingolstadt7, cologne3) so that it compiles in seconds.  Test infrastructure
    of tests/test_gpu_parity.py::test_device_invariant_counter -- never loaded by the package."""
    deps = [SRC] + [os.path.join(HERE, 'csrc', f) for f in ('resco_step.h', 'resco_tables.h', 'resco_policy.h')] + \
           [os.path.join(ROOT, 'include', f) for f in ('resco_sim.h', 'resco_model.h')]
    force = force or os.environ.get('GRAFT_FORCE_BUILD') == '1'
    if not force and os.path.exists(CHECK_LIB) and all(os.path.getmtime(CHECK_LIB) >= os.path.getmtime(d) for d in deps):
        return CHECK_LIB
    subprocess.check_call([HIPCC] + FLAGS + ['-DRS_DEVICE_ASSERT', '-DRS_ONE_CAP=256', SRC, '-o', CHECK_LIB])
    return CHECK_LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
