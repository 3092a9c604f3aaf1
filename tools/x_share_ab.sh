#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in "RTAMD_FUSE_DEEP=0" "RTAMD_FUSE_DEEP=1" "RTAMD_FUSE_DEEP=1 RTAMD_DEEP_EXIT_POLLS=4"; do
  echo "== $e"
  env $e timeout -k 10 150 python tools/share_serial.py 2>/dev/null || exit 1
done
