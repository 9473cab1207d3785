defmodule Matchmaking.Search.Engine do
  @moduledoc """
  NIF front of libmm_engine.so (native/mm_nif.c).  Replaces the body of
  `Matchmaking.Search.Worker.consume/5` (lib/search/worker.ex:291-324) and the RPC to the
  strategist it makes (worker.ex:296-306) with a batched search on the GPU.

  Columns are little-endian binaries: `ratings = for r <- rs, into: <<>>, do: <<r::little-signed-32>>`.
  Every function returns `:ok | {:ok, ...} | {:error, {code, charlist}}`; bad arguments raise
  `ArgumentError`.  One process owns an engine (see `Matchmaking.Search.EngineOwner`).
  """
  @on_load :load_nif

  def load_nif do
    :erlang.load_nif(:filename.join(:code.priv_dir(:matchmaking), ~c"mm_nif"), 0)
  end

  # mm_config as a binary; build with Matchmaking.Search.EngineConfig.encode/1
  def default_config(), do: :erlang.nif_error(:nif_not_loaded)
  def find_rating_group(_config, _rating), do: :erlang.nif_error(:nif_not_loaded)
  def create(_config), do: :erlang.nif_error(:nif_not_loaded)
  # ... with this engine's own tuning: names = ["team_late", ...] (fields of mm_tuning, include/mm_engine.h), values = <<v::little-32, ...>>
  def create(_config, _tuning_names, _tuning_values), do: :erlang.nif_error(:nif_not_loaded)
  def close(_engine), do: :erlang.nif_error(:nif_not_loaded)
  def reset(_engine), do: :erlang.nif_error(:nif_not_loaded)
  def enqueue(_engine, _ratings, _cons, _groups), do: :erlang.nif_error(:nif_not_loaded)
  def cancel(_engine, _slots), do: :erlang.nif_error(:nif_not_loaded)
  def tick(_engine, _mode), do: :erlang.nif_error(:nif_not_loaded)
  def queue_depth(_engine, _mode), do: :erlang.nif_error(:nif_not_loaded)
  # the launch shapes and fall-backs of the last tick: the u32 fields of mm_path_stats (include/mm_engine.h) as a binary
  def path_stats(_engine), do: :erlang.nif_error(:nif_not_loaded)
  def queue_slots(_engine, _mode, _group), do: :erlang.nif_error(:nif_not_loaded)
  def lobby_state(_engine, _mode, _group), do: :erlang.nif_error(:nif_not_loaded)
  def snapshot(_engine), do: :erlang.nif_error(:nif_not_loaded)
  def restore(_engine, _blob), do: :erlang.nif_error(:nif_not_loaded)
  def decode(_config, _mode_names, _region_key, _party_key, _role_key, _payloads, _offsets), do: :erlang.nif_error(:nif_not_loaded)
  def encode_lobby(_game_mode, _teams, _team_size, _payloads), do: :erlang.nif_error(:nif_not_loaded)
end
