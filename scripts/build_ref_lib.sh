#!/bin/bash
# Build libysmr_hip.so of another git revision into scripts/var_<name>.so (same-box A/B runs: scripts/ab_thr.sh, ab_bench_libs.sh)
# usage: scripts/build_ref_lib.sh <git-ref> <name> [EXTRA flags]
set -e
ref=$1; name=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C $root archive $ref ysmr_amd/csrc include | tar -x -C $tmp
make -s -j8 -C $tmp/ysmr_amd/csrc EXTRA="$extra"
cp $tmp/ysmr_amd/csrc/libysmr_hip.so $root/scripts/var_$name.so
rm -rf $tmp
echo built scripts/var_$name.so from $ref
