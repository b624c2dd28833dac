#!/bin/bash
# The build identity of csrc/Makefile (vsm_build_id()) recomputed from a tree: the working tree, or a commit.
#   tools/source_hash.sh            -> hash of the working tree's library sources
#   tools/source_hash.sh <commit>   -> hash of that commit's library sources
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ -n "$1" ]; then
  tmp=$(mktemp -d); trap 'rm -rf $tmp' EXIT
  git -C "$root" archive "$1" vsmartmom.jl_amd/csrc include/vsmartmom_hip.h | tar -x -C $tmp
  root=$tmp
fi
cd "$root/vsmartmom.jl_amd/csrc"
ls *.hip *.h | grep -v '^build_id.h$' | cat - <(echo ../../include/vsmartmom_hip.h; echo Makefile) | LC_ALL=C sort | xargs cat | sha256sum | cut -c1-16
