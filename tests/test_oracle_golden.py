"""The oracle against the hand-derived golden vectors (tests/golden/mode_r_cases.json)."""
import math

import pytest

from helpers import load_golden, run_golden_case
from microservice_matchmaking_amd.config import make_config, mode_1v1

GOLD = load_golden()


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_oracle_golden_case(oracle_cls, case):
    run_golden_case(oracle_cls, case)


def test_oracle_rating_group_edges(oracle_cls):
    eng = oracle_cls(make_config([mode_1v1()], capacity=8))
    for rating, want in GOLD["rating_group_cases"]["cases"]:
        r = math.nan if rating == "nan" else rating
        assert eng.find_rating_group(r) == want, (rating, want)
