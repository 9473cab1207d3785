defmodule Matchmaking.Search.EngineConfig do
  @moduledoc """
  Builds the `mm_config` binary (include/mm_engine.h) that `Matchmaking.Search.Engine.create/1`
  takes, from the application's own configuration: the rating groups of
  `config :matchmaking, RatingGroups` (config/config.exs:27-36, `{from, to, name}` rows, same order)
  and one entry per game mode with the parameters the strategist held for it (docs/MATCH_CHECK.md).

  Layout (little endian, no padding — every field is 4-byte aligned; 604 bytes; checked against the
  C struct by tests/test_nif.py with the same recipe):

      abi_version::32, n_groups::32, groups[16]::{from::signed-32, to::signed-32},
      default_group::32, n_modes::32,
      modes[16]::{team_size::32, teams::32, window::32, flags::32, n_roles::32, role_quota[8]::8},
      capacity::32, device::signed-32, flags::32
  """
  use Bitwise
  @abi_version 1
  @max_groups 16
  @max_modes 16
  @max_roles 8
  @mode_region_filter 1
  @mode_party_filter 2

  @doc """
  `modes`: list of keyword lists `[team_size: 5, teams: 2, window: 50, role_quota: [1, 1, 1, 1, 1],
  region_filter: false, party_filter: false]` in mode-index order (the index `decode/7` maps the
  `"game-mode"` string to).
  """
  def encode(rating_groups, modes, opts \\ []) do
    n_groups = length(rating_groups)
    n_modes = length(modes)
    true = n_groups in 1..@max_groups and n_modes in 1..@max_modes
    # Generic.Worker: @default_rating_group Enum.at(groups, div(n, 2) + 1)   (lib/generic/worker.ex:27)
    default_group = Keyword.get(opts, :default_group, div(n_groups, 2) + 1)
    groups = for {from, to, _name} <- rating_groups, into: <<>>, do: <<from::little-signed-32, to::little-signed-32>>
    groups = groups <> :binary.copy(<<0::64>>, @max_groups - n_groups)
    modes_bin = for m <- modes, into: <<>>, do: encode_mode(m)
    modes_bin = modes_bin <> :binary.copy(<<0::size(28)-unit(8)>>, @max_modes - n_modes)
    <<@abi_version::little-32, n_groups::little-32>> <> groups <>
      <<default_group::little-32, n_modes::little-32>> <> modes_bin <>
      <<Keyword.get(opts, :capacity, 1 <<< 20)::little-32, Keyword.get(opts, :device, 0)::little-signed-32,
        Keyword.get(opts, :flags, 0)::little-32>>
  end

  defp encode_mode(m) do
    quota = Keyword.get(m, :role_quota, [Keyword.fetch!(m, :team_size)])
    true = length(quota) in 1..@max_roles
    flags =
      if(Keyword.get(m, :region_filter, false), do: @mode_region_filter, else: 0) |||
        if(Keyword.get(m, :party_filter, false), do: @mode_party_filter, else: 0)
    quota_bin = :binary.list_to_bin(quota) <> :binary.copy(<<0>>, @max_roles - length(quota))
    <<Keyword.fetch!(m, :team_size)::little-32, Keyword.get(m, :teams, 2)::little-32,
      Keyword.fetch!(m, :window)::little-32, flags::little-32, length(quota)::little-32>> <> quota_bin
  end
end
