"""Configuration helpers: the reference's shipped rating groups and mode presets."""
from __future__ import annotations

from ._abi import (MM_ABI_VERSION, MM_CFG_TIMING, MM_MODE_PARTY_FILTER, MM_MODE_REGION_FILTER,
                   MMConfig)

# config/config.exs:27-36 of the reference
REFERENCE_RATING_GROUPS = [
    (0, 1499, "bronze"), (1500, 1999, "silver"), (2000, 2499, "gold"),
    (2500, 2999, "platinum"), (3000, 3499, "diamond"), (3500, 3999, "master"),
    (4000, 5000, "grandmaster"),
]


def mode_1v1(window=50, region_filter=False, party_filter=False):
    return dict(team_size=1, teams=2, window=window, n_roles=1, role_quota=[1],
                region_filter=region_filter, party_filter=party_filter)


def mode_team(team_size=5, teams=2, window=50, role_quota=(1, 1, 1, 1, 1),
              region_filter=False, party_filter=False):
    assert sum(role_quota) == team_size
    return dict(team_size=team_size, teams=teams, window=window, n_roles=len(role_quota),
                role_quota=list(role_quota), region_filter=region_filter,
                party_filter=party_filter)


def make_config(modes, capacity=1 << 20, groups=REFERENCE_RATING_GROUPS, default_group=None,
                device=0, timing=True) -> MMConfig:
    """Build an mm_config.  `default_group` None = the reference rule div(n,2)+1
    (lib/generic/worker.ex:27)."""
    cfg = MMConfig()
    cfg.abi_version = MM_ABI_VERSION
    cfg.n_groups = len(groups)
    for i, g in enumerate(groups):
        cfg.groups[i].from_ = int(g[0])
        cfg.groups[i].to = int(g[1])
    cfg.default_group = (len(groups) // 2 + 1) if default_group is None else default_group
    if cfg.default_group >= len(groups):
        cfg.default_group = len(groups) - 1
    cfg.n_modes = len(modes)
    for i, m in enumerate(modes):
        mc = cfg.modes[i]
        mc.team_size = m["team_size"]
        mc.teams = m["teams"]
        mc.window = m["window"]
        mc.flags = (MM_MODE_REGION_FILTER if m.get("region_filter") else 0) | (
            MM_MODE_PARTY_FILTER if m.get("party_filter") else 0)
        mc.n_roles = m["n_roles"]
        for r, q in enumerate(m["role_quota"]):
            mc.role_quota[r] = q
    cfg.capacity = capacity
    cfg.device = device
    cfg.flags = MM_CFG_TIMING if timing else 0
    return cfg


def mode_dicts(cfg: MMConfig):
    """Inverse of make_config's mode part (used by the literal reference in tests)."""
    out = []
    for i in range(cfg.n_modes):
        m = cfg.modes[i]
        out.append(dict(team_size=int(m.team_size), teams=int(m.teams), window=int(m.window),
                        n_roles=int(m.n_roles), role_quota=[int(m.role_quota[r]) for r in range(m.n_roles)],
                        region_filter=bool(m.flags & MM_MODE_REGION_FILTER),
                        party_filter=bool(m.flags & MM_MODE_PARTY_FILTER)))
    return out
