"""CPU: the C-ABI shared library loads, exports every symbol include/resco_sim.h declares, and fails loudly
(without computing anything) when no GPU is visible."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_scenario
from resco_amd import sim as rsim


def header_symbols():
    with open(os.path.join(ROOT, 'include', 'resco_sim.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(rs_[a-z_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from resco_amd.build import build_library
    build_library()
    L = rsim.load_library()
    syms = header_symbols()
    assert sorted(rsim.ABI_SYMBOLS) == syms
    for s in syms:
        assert hasattr(L, s), s


def test_struct_layout_matches_header():
    """the ctypes mirror lists the fields of rs_scenario in header order"""
    with open(os.path.join(ROOT, 'include', 'resco_sim.h')) as f:
        text = f.read()
    body = text[text.index('typedef struct rs_scenario {') + len('typedef struct rs_scenario {'):text.index('} rs_scenario;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl or decl.startswith('typedef'):
            continue
        decl = re.sub(r'^(const\s+)?(int32_t|uint32_t|float)\s*', '', decl)
        names += [n.strip().lstrip('*') for n in decl.split(',')]
    from resco_amd._abi import ScenarioStruct
    assert [f[0] for f in ScenarioStruct._fields_] == names


def _struct_fields(text, name):
    body = text[text.index('typedef struct %s {' % name) + len('typedef struct %s {' % name):text.index('} %s;' % name)]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r'^(const\s+)?(int32_t|uint32_t|float|rs_policy_handle)\s*', '', decl)
        names += [n.strip().lstrip('*') for n in decl.split(',')]
    return names


def test_params_and_group_agent_layouts_match_header():
    """rs_params (incl. tls_hold, round 6: the inverse of round 5's tls_expiry) and rs_group_agent: the ctypes mirrors list the header's fields in header order, with
    4-byte scalars and one pointer"""
    with open(os.path.join(ROOT, 'include', 'resco_sim.h')) as f:
        text = f.read()
    from resco_amd._abi import ParamsStruct
    assert [f[0] for f in ParamsStruct._fields_] == _struct_fields(text, 'rs_params')
    assert C.sizeof(ParamsStruct) == 4 * len(ParamsStruct._fields_)
    assert [f[0] for f in rsim.GroupAgent._fields_] == _struct_fields(text, 'rs_group_agent')
    assert C.sizeof(rsim.GroupAgent) == 8 + 8 + 4 * 4      # kind, step_key | policy pointer | mode, epsilon, epsilon_step, seed
    enum = re.search(r'enum rs_agent \{(.*?)\}', text, flags=re.S).group(1)
    vals = {k.strip(): int(v) for k, v in (item.split('=') for item in enum.split(','))}
    assert {k[len('RS_AGENT_'):].lower(): v for k, v in vals.items()} == rsim.AGENT


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    sc = load_scenario('cologne1')
    with pytest.raises(RuntimeError, match='rs_create failed'):
        rsim.BatchedSim(sc, 2)


def test_missing_extension_fails_loudly(monkeypatch):
    """no HIP library -> RuntimeError, never a silent fallback"""
    monkeypatch.setattr(rsim, '_lib', None)
    monkeypatch.setattr(rsim, 'LIB_PATH', os.path.join(ROOT, 'resco_amd', 'csrc', 'does_not_exist.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        rsim.load_library()
    monkeypatch.undo()
    rsim.load_library()
