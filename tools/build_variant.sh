#!/bin/bash
# Build an experimental copy of the library with extra compiler flags: tools/build_variant.sh <name> "<flags>"  ->  build/<name>/libydorb.so
# (run with YDORB_LIB=build/<name>/libydorb.so; build/ is git-ignored but travels to the GPU box)
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $HERE/build/$1
make -s -j4 -C $HERE/ydorbslam_amd/csrc OBJDIR=$HERE/build/$1/ OUT=$HERE/build/$1/libydorb.so EXTRA="$2"
ls -la $HERE/build/$1/libydorb.so
