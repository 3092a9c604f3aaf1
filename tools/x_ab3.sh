#!/bin/bash
# (scratch) the three scenes that are not lean under a list of environment settings.  usage: tools/x_ab3.sh "ENV.." "ENV.." ...
for e in "$@"; do
  bash tools/x_ab.sh "--scene rings --size 4" "$e" || exit 1
  bash tools/x_ab.sh "--scene mount --size 8 --lights 8" "$e" || exit 1
  bash tools/x_ab.sh "--scene lathe --size 96 --width 1920 --height 1080" "$e" || exit 1
done
