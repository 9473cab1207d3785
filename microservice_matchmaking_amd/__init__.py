"""MI355X-native matchmaking search engine — the search/seed path of
OpenMatchmaking/microservice-matchmaking (Matchmaking.Search.Worker + the strategist
predicate) as hand-written HIP kernels behind a C ABI.  See DESIGN.md."""
from ._abi import (MM_CFG_TIMING, MM_MODE_PARTY_FILTER, MM_MODE_REGION_FILTER, NO_SLOT, Matches,
                   MMConfig, MMError, cons_make)
from .config import REFERENCE_RATING_GROUPS, make_config, mode_1v1, mode_team

__all__ = ["MMConfig", "MMError", "Matches", "cons_make", "make_config", "mode_1v1", "mode_team",
           "REFERENCE_RATING_GROUPS", "NO_SLOT", "MM_CFG_TIMING", "MM_MODE_REGION_FILTER",
           "MM_MODE_PARTY_FILTER", "Engine"]


def __getattr__(name):
    if name == "Engine":          # lazy: importing the package must not need the .so
        from .engine import Engine
        return Engine
    raise AttributeError(name)
