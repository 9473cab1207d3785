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
    * durability of the waiting players is the pool snapshot (SURVEY.md 8(f) row 4): `terminate/2`
      and every `@snapshot_ms` write ONE file holding the engine snapshot AND the slot table
      (slot -> payload, decoded id), `init/1` restores both — a restored slot always has its row,
      and nothing is redelivered twice because nothing was left unacked;
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
    case Engine.create(config) do
      {:ok, engine} ->
        slots = :ets.new(:mm_slots, [:set, :private])     # {slot, payload, decoded id}
        ids = :ets.new(:mm_ids, [:set, :private])         # {decoded id, slot}
        with path when is_binary(path) <- opts[:snapshot_path],
             {:ok, file} <- File.read(path),
             {:mm_pool, 1, blob, rows} <- :erlang.binary_to_term(file, [:safe]),
             :ok <- Engine.restore(engine, blob) do
          # the pool and its slot table are ONE unit: a slot the engine can emit always has its row
          for {slot, payload, id} <- rows do
            :ets.insert(slots, {slot, payload, id})
            :ets.insert(ids, {id, slot})
          end
          Logger.info("search engine: pool of #{length(rows)} players restored from #{path}")
        end
        Process.send_after(self(), :tick, @tick_ms)
        if opts[:snapshot_path], do: Process.send_after(self(), :snapshot, @snapshot_ms)
        {:ok, %{engine: engine, config: config, modes: modes, pending: [], opts: opts,
                slots: slots, ids: ids, publish: Keyword.fetch!(opts, :publish)}}
      {:error, {code, text}} ->
        {:stop, {:engine, code, List.to_string(text)}}                  # like {:error, :noconn}, worker.ex:225-228
    end
  end

  @impl true
  def handle_cast({:delivery, payload, meta, channel}, state),
    do: {:noreply, %{state | pending: [{payload, meta, channel} | state.pending]}}

  def handle_cast({:cancel, player_id}, state) do
    with [{_, slot}] <- :ets.lookup(state.ids, player_id),
         [{_, _payload, ^player_id}] <- :ets.lookup(state.slots, slot) do   # still the slot's holder?
      :ok = Engine.cancel(state.engine, <<slot::little-32>>)
      # a cancelled player is never emitted (remove_inactive_players, worker.ex:267-280): its rows go now
      :ets.delete(state.slots, slot)
    end
    :ets.delete(state.ids, player_id)          # matched, cancelled or unknown: the id maps to nothing from here on
    {:noreply, state}
  end

  @impl true
  def handle_info(:tick, state) do
    Process.send_after(self(), :tick, @tick_ms)
    state = ingest(state)
    state.modes
    |> Enum.with_index()
    |> Enum.each(fn {{name, teams, team_size}, mode} -> search(state, name, teams, team_size, mode) end)
    {:noreply, state}
  end

  def handle_info(:snapshot, state) do
    Process.send_after(self(), :snapshot, @snapshot_ms)
    write_snapshot(state)
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
    Enum.zip(kept, chunk(slots, 4))
    |> Enum.each(fn {{{payload, meta, channel}, <<o::little-32>>, <<l::little-32>>}, <<slot::little-32>>} ->
      if slot != 0xFFFFFFFF do
        id = decoded_id(payload, o, l)
        :ets.insert(state.slots, {slot, payload, id})
        :ets.insert(state.ids, {id, slot})
      end
      # the engine holds the player now (or refused it for good: unknown mode / role): ack, as the
      # reference acks every delivery once its attempt is over (worker.ex:323)
      AMQP.Basic.ack(channel, meta.delivery_tag)
    end)
    # messages the reference would have crashed on (Poison.decode!, worker.ex:292) are rejected, not requeued
    for {{_payload, meta, channel}, false} <- Enum.zip(batch, keep), do: AMQP.Basic.reject(channel, meta.delivery_tag, requeue: false)
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
        for lobby <- chunk(slots, 4 * lobby_size) do
          members = for <<slot::little-32 <- lobby>>, do: hd(:ets.lookup(state.slots, slot))
          {:ok, json} = Engine.encode_lobby(name, teams, team_size, Enum.map(members, &elem(&1, 1)))
          state.publish.(@exchange_forward, @queue_forward, json)
          for {slot, _payload, id} <- members do
            :ets.delete(state.slots, slot)       # the slot goes back to the ring: forget who held it,
            :ets.delete(state.ids, id)           # and a late {:cancel, id} finds nothing
          end
        end
      {:error, {code, text}} -> Logger.error("search engine tick failed: #{code} #{text}")
    end
  end

  @impl true
  def terminate(_reason, state) do
    write_snapshot(state)
    Engine.close(state.engine)
  end

  # engine snapshot + slot table in ONE file, written to a temporary name and renamed
  defp write_snapshot(state) do
    with path when is_binary(path) <- state.opts[:snapshot_path],
         {:ok, blob} <- Engine.snapshot(state.engine) do
      file = :erlang.term_to_binary({:mm_pool, 1, blob, :ets.tab2list(state.slots)})
      :ok = File.write(path <> ".tmp", file)
      File.rename(path <> ".tmp", path)
    end
  end

  defp chunk(bin, width), do: for(<<c::binary-size(width) <- bin>>, do: c)
end
