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
      (`Engine.enqueue/4`); the payload binary, channel and delivery tag stay in ETS under the slot;
    * `Engine.tick/2` per game mode returns the lobbies in the reference's publish order; each is
      encoded (`Engine.encode_lobby/4`, worker.ex:315-318) and published to
      `@exchange_forward` / `@queue_forward` exactly as `prepare_game_lobby/4` does (:250-261),
      then the deliveries of its players are acked (:323);
    * players still queued stay unacked in the broker — the reference's at-least-once behaviour:
      if this process dies, the broker redelivers and the pool is rebuilt.  `terminate/2` also
      dumps a snapshot so that a planned restart does not wait for redelivery;
    * `ActiveUser.remove_user/1` callers additionally cast `{:cancel, player_id}` (active_user.ex:57-66).

  No requeue publish and no strategist RPC remain: a rejected player stays in the device queue
  (worker.ex:239-248 -> requeue/worker.ex:51-54 is the rotation the engine performs itself).
  """
  use GenServer
  require Logger
  alias Matchmaking.Search.Engine

  @tick_ms 10
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
        with path when is_binary(path) <- opts[:snapshot_path],
             {:ok, blob} <- File.read(path),
             :ok <- Engine.restore(engine, blob) do
          Logger.info("search engine: pool restored from #{path}")
        end
        Process.send_after(self(), :tick, @tick_ms)
        {:ok, %{engine: engine, config: config, modes: modes, pending: [], opts: opts,
                slots: :ets.new(:mm_slots, [:set, :private]),
                ids: :ets.new(:mm_ids, [:set, :private]), publish: Keyword.fetch!(opts, :publish)}}
      {:error, {code, text}} ->
        {:stop, {:engine, code, List.to_string(text)}}                  # like {:error, :noconn}, worker.ex:225-228
    end
  end

  @impl true
  def handle_cast({:delivery, payload, meta, channel}, state),
    do: {:noreply, %{state | pending: [{payload, meta, channel} | state.pending]}}

  def handle_cast({:cancel, player_id}, state) do
    case :ets.lookup(state.ids, player_id) do
      [{_, slot}] -> :ok = Engine.cancel(state.engine, <<slot::little-32>>)
      [] -> :ok
    end
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
        :ets.insert(state.slots, {slot, payload, meta, channel})
        :ets.insert(state.ids, {binary_part(payload, o, l), slot})
      end
    end)
    # messages the reference would have crashed on (Poison.decode!, worker.ex:292) are rejected, not requeued
    for {{_payload, meta, channel}, false} <- Enum.zip(batch, keep), do: AMQP.Basic.reject(channel, meta.delivery_tag, requeue: false)
    %{state | pending: []}
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
          for {slot, _payload, meta, channel} <- members do
            AMQP.Basic.ack(channel, meta.delivery_tag)                          # worker.ex:323
            :ets.delete(state.slots, slot)
          end
        end
      {:error, {code, text}} -> Logger.error("search engine tick failed: #{code} #{text}")
    end
  end

  @impl true
  def terminate(_reason, state) do
    with path when is_binary(path) <- state.opts[:snapshot_path], {:ok, blob} <- Engine.snapshot(state.engine),
         do: File.write(path, blob)
    Engine.close(state.engine)
  end

  defp chunk(bin, width), do: for(<<c::binary-size(width) <- bin>>, do: c)
end
