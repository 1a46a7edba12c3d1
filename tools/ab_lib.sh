# A/B of two builds of the library on one box: alternates tools/time_conv_shapes.py under both.  usage: bash tools/ab_lib.sh <exp.so> [rounds]
EXP="$1"; R="${2:-3}"
for i in $(seq 1 $R); do
  python tools/time_conv_shapes.py 2>/dev/null | tail -1
  DEPGAN_LIB="$EXP" python tools/time_conv_shapes.py 2>/dev/null | tail -1
done
