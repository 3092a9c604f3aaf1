#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in "RTAMD_FUSE_DEEP=0" "RTAMD_FUSE_DEEP=1" "RTAMD_FUSE_DEEP=0" "RTAMD_FUSE_DEEP=1"; do echo "== $e"; env $e timeout -k 5 100 python tools/x_fused_count.py $1 $2 2>/dev/null || exit 1; done
