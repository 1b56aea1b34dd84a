"""Generated model views (csrc/rmpc_spec_gen.hpp): the committed header is what the generator writes from the
library's own table builder, the shipped configurations resolve to a view, test configurations to the runtime
tables; on the GPU a view and the runtime tables (RMPC_NO_SPEC=1) give the same plans."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_specs", os.path.join(ROOT, "scripts", "gen_specs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_header_is_up_to_date():
    g = _gen()
    assert open(g.HEADER).read() == g.render(), "run python scripts/gen_specs.py and rebuild"


def test_shipped_configs_resolve_to_a_view():
    from robot_mpcs_amd._lib import spec_for
    from robot_mpcs_amd.scenarios import make_scenario

    names = {cfg: spec_for(make_scenario(cfg, B=1).desc) for cfg in
             ("cfg1", "cfg2", "cfg3", "cfg4", "pointRobot", "boxer", "panda", "wc_point", "wc_boxer", "wc_panda")}
    assert names["cfg2"] == "SpecPointRobot"
    assert names["cfg1"] == names["pointRobot"] == "SpecPointRobotExample"
    # the boxer, the arm and other structures (another module list, another obstacle count) run over the runtime tables
    assert names["cfg3"] == names["boxer"] == ""
    assert names["cfg4"] == names["panda"] == names["wc_panda"] == ""
    assert names["wc_point"] == names["wc_boxer"] == ""
    # the horizon, the weights and the solver options are not part of a view
    assert spec_for(make_scenario("cfg2", B=1, time_horizon=12).desc) == "SpecPointRobot"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B", [("cfg1", 8), ("cfg2", 512), ("pointRobot", 16)])
def test_view_equals_runtime_tables(cfg, B, monkeypatch):
    from robot_mpcs_amd._lib import Solver
    from robot_mpcs_amd.scenarios import make_scenario

    sc = make_scenario(cfg, B=B, seed=77)
    monkeypatch.delenv("RMPC_NO_SPEC", raising=False)
    a = Solver(sc.desc, max_batch=B)
    assert a.spec_name() != ""
    ra = a.solve(sc.xinit, sc.x0, sc.params)
    a.close()
    monkeypatch.setenv("RMPC_NO_SPEC", "1")     # (read once, at rmpc_create)
    b = Solver(sc.desc, max_batch=B)
    assert b.spec_name() == ""
    rb = b.solve(sc.xinit, sc.x0, sc.params)
    b.close()
    # same arithmetic, differently contracted multiply-adds: flags and iteration counts agree, plans to 1e-8
    assert np.mean(ra["exitflag"] == rb["exitflag"]) >= 0.99
    same = (ra["exitflag"] == rb["exitflag"]) & (ra["iters"] == rb["iters"]) & np.isin(ra["exitflag"], (1, 2))
    assert same.mean() >= 0.97, same.mean()
    scale = np.maximum(1.0, np.abs(rb["z"]).max(axis=(1, 2)))
    err = np.abs(ra["z"] - rb["z"]).max(axis=(1, 2)) / scale
    assert err[same].max() <= 1e-8, err[same].max()
