#!/bin/bash
# The build identity of csrc/Makefile (vsm_build_id()) recomputed from a tree: the working tree, or a commit (default make variables:
# the shipped build; the hash covers the library's sources AND the make variables that change the binary).
#   tools/source_hash.sh            -> hash of the working tree's library sources
#   tools/source_hash.sh <commit>   -> hash of that commit's library sources
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ -n "$1" ]; then
  tmp=$(mktemp -d); trap 'rm -rf $tmp' EXIT
  git -C "$root" archive "$1" vsmartmom.jl_amd/csrc include/vsmartmom_hip.h | tar -x -C $tmp
  root=$tmp
fi
make -s -C "$root/vsmartmom.jl_amd/csrc" print-build-id
