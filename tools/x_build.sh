#!/bin/bash
# usage: tools/x_build.sh <suffix> <extra compiler flags>  -- a variant library rust-raytracer_amd/librtamd_<suffix>.so
# for tools/x_time.sh (variants and their build directories are not part of the tree: delete them afterwards)
SUF=$1; shift
cd "$(dirname "$0")/../rust-raytracer_amd" && make OUT=build_x$SUF LIB=librtamd_x$SUF.so EXTRA="$*" librtamd_x$SUF.so 2>&1 | grep -E "error|Error"
ls -la librtamd_x$SUF.so
