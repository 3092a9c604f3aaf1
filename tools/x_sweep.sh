# usage (on the GPU box): bash tools/x_sweep.sh <variant suffixes...>  -- serial kernel times of four scenes per variant library
for v in "$@"; do
  [ "$v" = "base" ] && v=""
  tools/x_time.sh "$v" "balls_$v" || exit 1
  tools/x_time.sh "$v" "rings_$v" --scene rings --size 4 || exit 1
  tools/x_time.sh "$v" "mount_$v" --scene mount --size 8 || exit 1
  tools/x_time.sh "$v" "lathe_$v" --scene lathe --size 96 --width 1920 --height 1080 || exit 1
done
