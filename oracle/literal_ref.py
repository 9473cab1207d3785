"""literal_ref.py — line-by-line Python restatement of the reference search stage.

TEST INFRASTRUCTURE ONLY (small cases; pure-Python loops).  PARITY UNPINNED by the
reference — see oracle/mode_r.h.  Its job is to pin oracle/mode_r.c: this file keeps the
reference's *shape* (JSON-like player maps, a broker queue, a lobby table with pop-on-read,
an ActiveUser set, one strategist RPC per attempt), while mode_r.c uses the compact
pass formulation.  tests/test_oracle_literal.py checks that both produce identical
emissions.

Reference lines followed (paths relative to /root/reference/matchmaking/):
  find_rating_group_by_rating   lib/generic/worker.ex:46-53, :27
  LobbyState.get_state/update_state   lib/models/lobby_state.ex:61-131
  ActiveUser.in_queue?/add_user/remove_user   lib/models/active_user.ex:33-66
  SearchWorker.consume   lib/search/worker.ex:291-324
  remove_inactive_players / get_players_count   lib/search/worker.ex:263-280
  requeue (tail re-entry)   lib/search/worker.ex:239-248, lib/requeue/worker.ex:51-54
  strategist_match_check   docs/MATCH_CHECK.md §2 (external service, call site
                           lib/search/worker.ex:296-306)
"""
from __future__ import annotations

import copy
import math
from collections import deque

RATING_GROUPS = [  # config/config.exs:27-36
    (0, 1499, "bronze"),
    (1500, 1999, "silver"),
    (2000, 2499, "gold"),
    (2500, 2999, "platinum"),
    (3000, 3499, "diamond"),
    (3500, 3999, "master"),
    (4000, 5000, "grandmaster"),
]


def default_rating_group(groups):
    # generic/worker.ex:27  Enum.at(@groups, Integer.floor_div(length(@groups), 2) + 1)
    return groups[len(groups) // 2 + 1]


def find_rating_group_by_rating(rating, groups=RATING_GROUPS):
    """generic/worker.ex:46-53.  A non-number rating compares greater than every number
    under Erlang term order, so it falls through to the default group."""
    if isinstance(rating, bool) or not isinstance(rating, (int, float)) or (
        isinstance(rating, float) and math.isnan(rating)
    ):
        return default_rating_group(groups)
    for g in groups:
        if rating >= g[0] and rating <= g[1]:
            return g
    return default_rating_group(groups)


def team_name(t):
    return "team %d" % (t + 1)


def strategist_match_check(mode_cfg, game_mode, new_player, grouped_players):
    """docs/MATCH_CHECK.md §2.  `grouped_players` is the lobby map team-name -> [player];
    `%{}` (lobby_state.ex:54-56) for a fresh lobby.  Returns the reply content the
    reference reads: added / is_filled / grouped-players."""
    teams = mode_cfg["teams"]
    lobby = {team_name(t): list(grouped_players.get(team_name(t), [])) for t in range(teams)}
    anchor = None
    for t in range(teams):
        if lobby[team_name(t)]:
            anchor = lobby[team_name(t)][0]
            break
    ok = True
    if anchor is not None:
        if abs(new_player["rating"] - anchor["rating"]) > mode_cfg["window"]:
            ok = False
        if mode_cfg.get("region_filter") and new_player["region"] != anchor["region"]:
            ok = False
        if mode_cfg.get("party_filter") and new_player["party"] != anchor["party"]:
            ok = False
    added = False
    if ok:
        role = new_player["role"]
        best = None
        for t in range(teams):
            members = lobby[team_name(t)]
            if sum(1 for m in members if m["role"] == role) >= mode_cfg["role_quota"][role]:
                continue
            s = sum(m["rating"] for m in members)
            if best is None or s < best[0]:
                best = (s, t)
        if best is not None:
            lobby[team_name(best[1])].append(new_player)
            added = True
    filled = all(len(lobby[team_name(t)]) == mode_cfg["team_size"] for t in range(teams))
    return {"added": added, "is_filled": filled, "grouped-players": lobby}


class ActiveUser:
    """lib/models/active_user.ex — a set of ids."""

    def __init__(self):
        self.table = set()

    def in_queue(self, user_id):
        return user_id in self.table

    def add_user(self, user_id):
        self.table.add(user_id)

    def remove_user(self, user_id):
        self.table.discard(user_id)


class LobbyState:
    """lib/models/lobby_state.ex — per group a table of {id, dump, game_mode}; get_state
    pops the first record of the mode (select limit 1 + delete), update_state inserts
    under a fresh id.  Records are kept in insertion order (the canonical schedule never
    has more than one per (group, mode), so the choice of record is never exercised)."""

    def __init__(self, groups):
        self.tables = {g[2]: [] for g in groups}
        self.next_id = 0

    def get_state(self, group, game_mode):
        tab = self.tables[group]
        for k, (rid, dump, gm) in enumerate(tab):
            if gm == game_mode:
                del tab[k]
                return copy.deepcopy(dump)
        return {}

    def update_state(self, group, game_mode, state):
        self.next_id += 1
        self.tables[group].append((self.next_id, copy.deepcopy(state), game_mode))


def get_players_count(teams):
    return sum(len(v) for v in teams.values())


class SearchStage:
    """The generic hop + one Search.Worker per rating group, driven synchronously."""

    def __init__(self, mode_cfgs, groups=RATING_GROUPS):
        self.groups = groups
        self.mode_cfgs = mode_cfgs            # game-mode name -> dict
        self.queues = {g[2]: deque() for g in groups}
        self.lobbies = LobbyState(groups)
        self.active = ActiveUser()
        self.emitted = []                     # publish order, search/worker.ex:313-319
        self.pairs = 0

    # middleware + generic hop --------------------------------------------------------
    def deliver(self, player):
        self.active.add_user(player["id"])    # middleware/worker.ex:65-70
        g = find_rating_group_by_rating(player.get("rating"), self.groups)
        self.queues[g[2]].append(player)      # generic/worker.ex:55-66

    def cancel(self, player_id):
        self.active.remove_user(player_id)

    # search/worker.ex:291-324 --------------------------------------------------------
    def consume(self, group_name, payload):
        player_data = dict(payload)
        game_mode = player_data["game-mode"]
        player = {k: v for k, v in player_data.items() if k != "game-mode"}
        grouped_players = self.lobbies.get_state(group_name, game_mode)
        if get_players_count(grouped_players) > 0 and self.active.in_queue(player_data["id"]):
            self.pairs += 1
        data = strategist_match_check(self.mode_cfgs[game_mode], game_mode, player, grouped_players)
        requeued = False
        if self.active.in_queue(player_data["id"]) and not data["added"]:
            self.queues[group_name].append(payload)            # requeue -> tail
            requeued = True
        # remove_inactive_players, worker.ex:267-280
        teams = data["grouped-players"]
        updated = {t: [p for p in teams[t] if self.active.in_queue(p["id"])] for t in teams}
        is_changed = get_players_count(teams) != get_players_count(updated)
        seated = data["added"] and self.active.in_queue(player_data["id"])
        if data["is_filled"] and not is_changed:
            self.emitted.append({"teams": updated, "game-mode": game_mode, "group": group_name})
            for t in updated:                                   # game-lobby/worker.ex:73-103
                for p in updated[t]:
                    self.active.remove_user(p["id"])
        else:
            self.lobbies.update_state(group_name, game_mode, updated)
        return seated, requeued

    def run_group_to_quiescence(self, group_name, pass_log=None):
        """Canonical schedule (docs/MATCH_CHECK.md §4): full rotations until one seats
        nobody or the queue drains.  All modes of the group share the queue."""
        q = self.queues[group_name]
        passes = 0
        while q:
            changed = False
            for _ in range(len(q)):
                payload = q.popleft()
                n_before = len(self.emitted)
                seated, _ = self.consume(group_name, payload)
                changed = changed or seated
                if pass_log is not None and len(self.emitted) > n_before:
                    pass_log.append(passes)
            passes += 1
            if not changed:
                break
        return passes

    def tick(self):
        """Runs every group; returns emissions of this tick, group-major."""
        start = len(self.emitted)
        out = []
        for g in self.groups:
            n0 = len(self.emitted)
            self.run_group_to_quiescence(g[2])
            out.extend(self.emitted[n0:])
        assert len(self.emitted) - start == len(out)
        return out
