# Ablations of the wave-private kernel (profiles/r03_conv_experiments.md).  The ablated instantiations are compiled only
# with -DDEPGAN_WP_ABLATIONS: build csrc/igemm_wp.hip with that flag into a second library and point DEPGAN_LIB at it
# (tools/ab_lib.sh shows the recipe), then run this script.
for abl in 0 1 2 3 4; do echo "ABL=$abl"; DEPGAN_WP_ABL=$abl timeout -k 10 120 python tools/ab_wp.py 2>&1 | grep "b32 256\|b32 128x128 64" ; done
