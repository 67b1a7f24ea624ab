#!/bin/bash
# A/B of rank 0's share of a round (sharding.rest_root_sizes, MDQE_SHARD_ROOT_SHARE) in the N-rank root-load rehearsal, recompute and halo form.
#   bash tools/root_share_ab.sh <world> "<shares>" "<halo shares>"
cd "$(dirname "$0")/.."
w=${1:-8}
for sh in ${2:-1 0.95 0.9}; do
  echo "== N=$w recompute, root share $sh"
  MDQE_SHARD_ROOT_SHARE=$sh bash tools/root_load.sh $w 2>&1 | grep "root load"
done
for sh in ${3:-0.9 0.85}; do
  echo "== N=$w halo exchange, root share $sh"
  MDQE_SHARD_ROOT_SHARE=$sh HALO=1 bash tools/root_load.sh $w 2>&1 | grep "root load"
done
