"""CPU: the oracle (oracle/reference_path.py) against every golden fixture the
imported reference produced (tests/golden/make_golden.py).  This is what pins
the oracle; tolerance is fp32 reordering noise."""
import pytest
import torch

from tests.conftest import load_fixture
from tests.golden import synth
from tests.golden.cases import CASES
from tests.golden.runner import OracleBackend, run_case

TOL = 2e-6          # oracle vs reference: same ATen kernels, same order -> ~1e-7
BIG = {"ffno2d_cfg3_256", "fno2d_512", "fs2d_r256", "sc2d_256", "ffno2d_cfg3_128"}
# the shipped yaml's 64 modes x 4 layers amplify fp32 reordering noise in the input gradient: measured 1.3e-6 (sampled
# scalars) / 2.1e-6 (projections) at 256^2 and 2.0e-6 at 64^2 between two fp32 evaluations of the same graph
TOL_CASE = {"ffno2d_yaml_256": 5e-6, "ffno2d_yaml_64": 5e-6}


@pytest.mark.parametrize("name", [c["name"] for c in CASES])
def test_oracle_matches_reference_fixture(name):
    torch.set_num_threads(8)
    case, spec, digests = load_fixture(name)
    sd = synth.fill_state_dict(spec, case["seed"])
    res = run_case(case, OracleBackend(), sd)
    synth.check_results(res, digests, TOL_CASE.get(name, TOL), label=name)
