defmodule Matchmaking.Search.EngineOwner do
  @moduledoc """
  The one process that owns the GPU engine (include/mm_engine.h: one owner thread per engine).

  It takes over what `Matchmaking.Search.Worker.consume/5` did per delivery
  (lib/search/worker.ex:291-324) — for all rating groups and game modes at once, once per
  `@tick_ms`:

    * the per-group `Search.Worker`s keep their AMQP consumers (worker.ex:46-52, :352-358) and
      only forward `{:delivery, payload, meta}` here instead of spawning `consume/5` (:356);
    * deliveries of a tick period are decoded in one call (`Engine.decode/7`:
      Poison.decode! + `find_rating_group_by_rating/1`, generic/worker.ex:46-57) and enqueued
      (`Engine.enqueue/4`); the payload binary and the decoded id stay in ETS under the slot;
    * a delivery is ACKED WHEN IT IS INGESTED, as the reference acks every delivery at the end of its
      attempt (worker.ex:323) whether the player was seated or requeued: from then on the player
      lives in the engine's pool (the reference: in the Mnesia lobby or in the requeue message).
      Holding deliveries unacked until they match would stop the broker at the channels'
      `prefetch_count: 10` (worker.ex:29) and cap the pool at ten players per group;
    * `Engine.tick/2` per game mode returns the lobbies in the reference's publish order; each is
      encoded (`Engine.encode_lobby/4`, worker.ex:315-318) and published to
      `@exchange_forward` / `@queue_forward` exactly as `prepare_game_lobby/4` does (:250-261);
      the rows of its players leave both tables;
    * durability of the waiting players: the reference never loses a waiting player that was not acked
      (worker.ex:323) — so a delivery is acked only AFTER its payload is in the JOURNAL (an append-only
      file, `:file.datasync/1` per ingested batch, i.e. once per tick at most).  The pool snapshot
      (SURVEY.md 8(f) row 4) only bounds the journal: every `@snapshot_ms` the engine blob is taken in the
      owner (a device-to-host copy) and handed, with a copy of the slot table, to a separate process that
      serialises and writes it (`term_to_binary` of a 1M-player table would stall the 10 ms tick loop for
      hundreds of ms); when the file is in place the journal generation before it is deleted.  `init/1`
      restores snapshot + slot table and replays the journal IN ORDER: an `{:out, ids}` record takes out what
      of those ids came before it — journalled deliveries and rows of the restored pool alike (those are
      cancelled in the engine before anything is re-ingested) — and nothing that came after it.  Loss window
      of acked players after kill -9: none.  What a crash CAN do is re-emit a lobby whose `{:out, _}` record
      had not reached the disk — the reference has the same at-least-once edge between its publish
      (worker.ex:319) and its ack (:323).  (An `:out` names ids, not deliveries: of two deliveries of one id
      the replay drops both when one of them was matched.)
    * a failed tick (`{:error, _}` from `Engine.tick/2`: the engine is mid-tick and refuses everything but
      reset / restore — include/mm_engine.h, MM_ERR_STATE) STOPS the owner without a snapshot; the supervisor
      restarts it (`restart: :transient`, application.ex:8-14) and `init/1` rebuilds the pool from the last
      snapshot + journal, exactly as after a crash;
    * `ActiveUser.remove_user/1` callers additionally cast `{:cancel, player_id}` (active_user.ex:57-66)
      with the DECODED id (`player["id"]`, the term the lobby worker holds: game-lobby/worker.ex:80, :96).
      The id table maps it to the slot; the row is checked against the slot's current holder before
      `Engine.cancel/2`, and both rows are deleted on cancel and on emission, so a late cancel for a
      matched player can never hit whoever holds the recycled slot.

  No requeue publish and no strategist RPC remain: a rejected player stays in the device queue
  (worker.ex:239-248 -> requeue/worker.ex:51-54 is the rotation the engine performs itself).
  """
  use GenServer
  require Logger
  alias Matchmaking.Search.Engine

  @tick_ms 10
  @snapshot_ms 5_000
  @exchange_forward "open-matchmaking.matchmaking.game-lobby.direct"   # worker.ex:31
  @queue_forward "matchmaking.queues.lobbies"                          # worker.ex:32
  @status_ok 0
  @status_rating_inexact 4
  @status_rating_not_number 5

  def start_link(opts), do: GenServer.start_link(__MODULE__, opts, name: __MODULE__)
  def deliver(payload, meta, channel), do: GenServer.cast(__MODULE__, {:delivery, payload, meta, channel})
  def cancel(player_id), do: GenServer.cast(__MODULE__, {:cancel, player_id})

  @impl true
  def init(opts) do
    config = Keyword.fetch!(opts, :config)            # binary mm_config
    modes = Keyword.fetch!(opts, :modes)              # [{"duel", teams, team_size}, ...] in mode-index order
    tuning = Keyword.get(opts, :tuning, [])           # [team_late: 0, ...]: this engine's mm_tuning (include/mm_engine.h); [] = defaults
    case Engine.create(config, Enum.map(tuning, fn {k, _} -> Atom.to_string(k) end), for({_, v} <- tuning, into: <<>>, do: <<v::little-32>>)) do
      {:ok, engine} ->
        slots = :ets.new(:mm_slots, [:set, :private])     # {slot, payload, decoded id}
        ids = :ets.new(:mm_ids, [:bag, :private])         # {decoded id, slot}; a bag: the same id may be delivered twice
        state = %{engine: engine, config: config, modes: modes, pending: [], opts: opts, failed: false,
                  slots: slots, ids: ids, publish: Keyword.fetch!(opts, :publish), journal: nil, generation: 0}
        generation =
          with path when is_binary(path) <- opts[:snapshot_path],
               {:ok, file} <- File.read(path),
               {:mm_pool, 2, gen, blob, rows} <- decode_snapshot(file),
               :ok <- Engine.restore(engine, blob) do
            # the pool and its slot table are ONE unit: a slot the engine can emit always has its row
            for {slot, payload, id} <- rows do
              :ets.insert(slots, {slot, payload, id})
              :ets.insert(ids, {id, slot})
            end
            Logger.info("search engine: pool of #{length(rows)} players restored from #{path}")
            gen
          else
            _ -> 0                                        # no file, unknown format (logged) or a refused blob: start empty
          end
        state = replay_journal(%{state | generation: generation})
        Process.send_after(self(), :tick, @tick_ms)
        if opts[:snapshot_path], do: Process.send_after(self(), :snapshot, @snapshot_ms)
        {:ok, state}
      {:error, {code, text}} ->
        {:stop, {:engine, code, List.to_string(text)}}                  # like {:error, :noconn}, worker.ex:225-228
    end
  end

  @impl true
  def handle_cast({:delivery, payload, meta, channel}, state),
    do: {:noreply, %{state | pending: [{payload, meta, channel} | state.pending]}}

  def handle_cast({:cancel, player_id}, state) do
    # ActiveUser.remove_user/1 removes the id, not a delivery: every slot the id still holds goes
    for {_, slot} <- :ets.lookup(state.ids, player_id),
        match?([{_, _payload, ^player_id}], :ets.lookup(state.slots, slot)) do   # still the slot's holder?
      :ok = Engine.cancel(state.engine, <<slot::little-32>>)
      # a cancelled player is never emitted (remove_inactive_players, worker.ex:267-280): its rows go now
      :ets.delete(state.slots, slot)
      :ets.match_delete(state.ids, {player_id, slot})          # this pair only: never another holder's mapping
    end
    journal(state, {:out, [player_id]}, false)
    {:noreply, state}
  end

  @impl true
  def handle_info(:tick, state) do
    Process.send_after(self(), :tick, @tick_ms)
    state = ingest(state)
    result =
      state.modes
      |> Enum.with_index()
      |> Enum.reduce_while(:ok, fn {{name, teams, team_size}, mode}, :ok ->
        case search(state, name, teams, team_size, mode) do
          :ok -> {:cont, :ok}
          error -> {:halt, error}
        end
      end)
    case result do
      :ok -> {:noreply, state}
      # include/mm_engine.h: after a failed tick the pool is mid-tick and the engine answers MM_ERR_STATE to
      # everything but reset / restore.  Stop WITHOUT snapshotting it; the restart restores snapshot + journal.
      {:error, {code, text}} -> {:stop, {:engine_tick_failed, code, List.to_string(text)}, %{state | failed: true}}
    end
  end

  def handle_info(:snapshot, state) do
    Process.send_after(self(), :snapshot, @snapshot_ms)
    {:noreply, write_snapshot(state)}
  end

  # the writer process reports back: the snapshot of `generation` is in place, older journals can go
  def handle_info({:snapshot_written, generation}, state) do
    with path when is_binary(path) <- state.opts[:snapshot_path] do
      for old <- Path.wildcard(path <> ".journal.*"),
          String.to_integer(Path.extname(old) |> String.trim_leading(".")) < generation,
          do: File.rm(old)
    end
    {:noreply, state}
  end

  # generic/worker.ex:55-69 + search/worker.ex:352-358 for the batch
  defp ingest(%{pending: []} = state), do: state
  defp ingest(state) do
    batch = Enum.reverse(state.pending)
    payloads = Enum.map(batch, &elem(&1, 0))
    {offsets, _} = Enum.map_reduce([0 | Enum.map(payloads, &byte_size/1)], 0, fn n, acc -> {acc + n, acc + n} end)
    offsets_bin = for o <- offsets, into: <<>>, do: <<o::little-64>>
    names = Enum.map(state.modes, &elem(&1, 0))
    {:ok, ratings, cons, groups, status, id_off, id_len} =
      Engine.decode(state.config, names, "region", "party", "role", IO.iodata_to_binary(payloads), offsets_bin)
    status = :binary.bin_to_list(status)
    keep = for s <- status, do: s in [@status_ok, @status_rating_inexact, @status_rating_not_number]
    pick = fn bin, width -> for {<<v::binary-size(width)>>, true} <- Enum.zip(chunk(bin, width), keep), into: <<>>, do: v end
    {:ok, slots, _accepted, _rejected} =
      Engine.enqueue(state.engine, pick.(ratings, 4), pick.(cons, 4), pick.(groups, 1))
    kept = for {item, true} <- Enum.zip(Enum.zip([batch, chunk(id_off, 4), chunk(id_len, 4)]), keep), do: item
    rows =
      for {{{payload, _meta, _channel}, <<o::little-32>>, <<l::little-32>>}, <<slot::little-32>>} <- Enum.zip(kept, chunk(slots, 4)),
          slot != 0xFFFFFFFF do
        id = decoded_id(payload, o, l)
        :ets.insert(state.slots, {slot, payload, id})
        :ets.insert(state.ids, {id, slot})
        payload
      end
    # journal first (one datasync per batch), ack after: an acked player is on disk, as an unacked one is in the broker
    journal(state, {:in, rows}, true)
    # the engine holds the player now (or refused it for good: unknown mode / role): ack, as the
    # reference acks every delivery once its attempt is over (worker.ex:323)
    for {{_payload, meta, channel}, _o, _l} <- kept, channel != :replay, do: AMQP.Basic.ack(channel, meta.delivery_tag)
    # messages the reference would have crashed on (Poison.decode!, worker.ex:292) are rejected, not requeued
    for {{_payload, meta, channel}, false} <- Enum.zip(batch, keep), channel != :replay,
      do: AMQP.Basic.reject(channel, meta.delivery_tag, requeue: false)
    %{state | pending: []}
  end

  # The "id" member as Poison.decode! would give it (player["id"], worker.ex:272, :308): mm_decode_players
  # returns the span of its value — the contents of a string with escapes unresolved, or the text of a number.
  defp decoded_id(_payload, _o, 0), do: nil
  defp decoded_id(payload, o, l) do
    span = binary_part(payload, o, l)
    if o > 0 and binary_part(payload, o - 1, 1) == "\"",
      do: Poison.decode!(<<?", span::binary, ?">>),
      else: Poison.decode!(span)
  end

  # search/worker.ex:291-324 to quiescence for one game mode, then :250-261 per emitted lobby
  defp search(state, name, teams, team_size, mode) do
    case Engine.tick(state.engine, mode) do
      {:ok, 0, _l, _slots, _scores, _groups, _stats} -> :ok
      {:ok, _n, lobby_size, slots, _scores, _groups, _stats} ->
        out =
          for lobby <- chunk(slots, 4 * lobby_size) do
            members = for <<slot::little-32 <- lobby>>, do: hd(:ets.lookup(state.slots, slot))
            {:ok, json} = Engine.encode_lobby(name, teams, team_size, Enum.map(members, &elem(&1, 1)))
            state.publish.(@exchange_forward, @queue_forward, json)
            for {slot, _payload, id} <- members do
              :ets.delete(state.slots, slot)               # the slot goes back to the ring: forget who held it,
              :ets.match_delete(state.ids, {id, slot})     # and only THIS holder's mapping (the id may hold another slot)
              id
            end
          end
        journal(state, {:out, List.flatten(out)}, false)
        :ok
      {:error, {code, text}} = error ->
        Logger.error("search engine tick failed: #{code} #{text}")
        error
    end
  end

  @impl true
  def terminate(_reason, %{failed: true} = state), do: Engine.close(state.engine)   # never snapshot a mid-tick pool
  def terminate(_reason, state) do
    write_snapshot(state, :sync)
    Engine.close(state.engine)
  end

  # ---- durability: journal (what was acked since the last snapshot) + snapshot (what bounds the journal) ----

  # a snapshot file: term_to_binary({:mm_pool, 2, generation, engine_blob, slot_rows}).  Anything else — the raw
  # engine blob a previous version of this module wrote, a truncated file — is logged and ignored, never raised on.
  defp decode_snapshot(<<131, _::binary>> = file) do
    try do
      :erlang.binary_to_term(file, [:safe])
    rescue
      ArgumentError -> Logger.warn("search engine: snapshot file is not a term; starting empty"); :unknown
    end
  end
  defp decode_snapshot(_other), do: (Logger.warn("search engine: snapshot file of an unknown format; starting empty"); :unknown)

  defp journal_path(state, generation), do: state.opts[:snapshot_path] <> ".journal." <> Integer.to_string(generation)

  defp journal(%{journal: nil}, _record, _sync), do: :ok
  defp journal(_state, {:in, []}, _sync), do: :ok
  defp journal(_state, {:out, []}, _sync), do: :ok
  defp journal(state, record, sync) do
    bin = :erlang.term_to_binary(record)
    :ok = :file.write(state.journal, [<<byte_size(bin)::32>>, bin])
    if sync, do: :ok = :file.datasync(state.journal)
    :ok
  end

  # after a restart: what the journal of the restored generation says came in and did not go out is ingested again
  # (fresh slots; the deliveries were acked before the crash, so there is nothing to ack)
  defp replay_journal(%{opts: opts} = state) do
    case opts[:snapshot_path] do
      nil -> state
      _path ->
        # every generation from the restored snapshot's on: a crash between "blob taken" and "file in place" leaves
        # the snapshot of generation g on disk with the journals g (up to the blob) and g + 1 (after it)
        generations =
          for file <- Path.wildcard(opts[:snapshot_path] <> ".journal.*"),
              gen = String.to_integer(Path.extname(file) |> String.trim_leading(".")), gen >= state.generation, do: gen
        last = Enum.max([state.generation | generations])
        records = Enum.flat_map(Enum.sort(generations), &read_journal(journal_path(state, &1)))
        # The records IN ORDER: an {:out, id} takes out everything of that id that came before it — the journalled
        # deliveries AND the id's rows in the restored pool (matched or cancelled after the snapshot was taken) — and
        # nothing that came after it (a player who cancelled and queued again within the snapshot period stays).
        {live, gone, _seq} =
          Enum.reduce(records, {%{}, MapSet.new(), 0}, fn
            {:in, rows}, {live, gone, seq} ->
              Enum.reduce(rows, {live, gone, seq}, fn payload, {l, g, n} ->
                {Map.update(l, Poison.decode!(payload)["id"], [{n, payload}], &[{n, payload} | &1]), g, n + 1}
              end)
            {:out, ids}, {live, gone, seq} -> {Map.drop(live, ids), Enum.into(ids, gone), seq}
          end)
        # at this point the id table holds the restored pool only: whoever of it has left since is cancelled in the engine
        # (a cancelled player is never emitted, remove_inactive_players worker.ex:267-280) before anybody is re-ingested
        for id <- gone, {_, slot} <- :ets.lookup(state.ids, id) do
          :ok = Engine.cancel(state.engine, <<slot::little-32>>)
          :ets.delete(state.slots, slot)
          :ets.match_delete(state.ids, {id, slot})
        end
        payloads = live |> Map.values() |> List.flatten() |> Enum.sort() |> Enum.map(&elem(&1, 1))   # arrival order
        state = %{state | generation: last}
        {:ok, fd} = :file.open(journal_path(state, last), [:append, :raw, :binary])
        state = %{state | journal: fd}
        pending = for payload <- payloads, do: {payload, %{delivery_tag: nil}, :replay}
        # ingest/1 without acks (channel :replay) and without journalling again (the records are on disk already).  Every
        # replayed payload held a slot when it was journalled: one that finds none now (a pool filled by restore + replay:
        # mm_enqueue refuses the whole batch, the match in ingest/1 fails; a row refused on its own comes back as 0xFFFFFFFF)
        # must not vanish with its journal generation — init fails instead, the journal files stay (ADVICE r04)
        held = :ets.info(state.slots, :size)
        state = ingest(%{state | pending: Enum.reverse(pending), journal: nil}) |> Map.put(:journal, fd)
        if :ets.info(state.slots, :size) - held != length(payloads),
          do: raise("replay: #{length(payloads) - (:ets.info(state.slots, :size) - held)} acked players found no slot")
        state
    end
  end

  defp read_journal(path) do
    case File.read(path) do
      {:ok, bin} -> for <<n::32, rec::binary-size(n) <- bin>>, do: :erlang.binary_to_term(rec, [:safe])   # a torn tail is dropped
      _ -> []
    end
  end

  # engine snapshot + slot table in ONE file, written to a temporary name and renamed — by a process of its own:
  # only the blob (a device-to-host copy) and the table copy happen in the owner
  defp write_snapshot(state, how \\ :async) do
    with path when is_binary(path) <- state.opts[:snapshot_path],
         {:ok, blob} <- Engine.snapshot(state.engine) do
      generation = state.generation + 1
      rows = :ets.tab2list(state.slots)
      owner = self()
      write = fn ->
        file = :erlang.term_to_binary({:mm_pool, 2, generation, blob, rows})
        tmp = path <> ".tmp." <> Integer.to_string(generation)       # a name per generation: writers never share a file
        :ok = File.write(tmp, file, [:sync])
        :ok = File.rename(tmp, path)
        send(owner, {:snapshot_written, generation})
      end
      # from here on the journal of the NEW generation takes the acks; the old one stays until the file is in place
      if state.journal, do: :file.close(state.journal)
      {:ok, fd} = :file.open(journal_path(state, generation), [:append, :raw, :binary])
      if how == :sync, do: write.(), else: spawn(write)
      %{state | generation: generation, journal: fd}
    else
      _ -> state
    end
  end

  defp chunk(bin, width), do: for(<<c::binary-size(width) <- bin>>, do: c)
end
