for abl in 0 1 2 3 4; do echo "ABL=$abl"; DEPGAN_WP_ABL=$abl timeout -k 10 120 python tools/ab_wp.py 2>&1 | grep "b32 256\|b32 128x128 64" ; done
