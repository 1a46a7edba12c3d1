# Second library with the ablated instantiations of the Winograd kernel (profiles/r03_conv_experiments.md):
#   bash tools/build_wino_abl.sh  ->  dep_gan_im_amd/libdepgan_abl.so ; DEPGAN_LIB=<that> DEPGAN_WINO_ABL=<bits> ...
set -e
cd "$(dirname "$0")/.."
python -m dep_gan_im_amd.build > /dev/null
B=dep_gan_im_amd/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -DDEPGAN_WINO_ABLATIONS \
  -c dep_gan_im_amd/csrc/igemm_wino.hip -o $B/igemm_wino_abl.o
OBJS=$(ls $B/*.o | grep -v "igemm_wino")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $B/igemm_wino_abl.o -o dep_gan_im_amd/libdepgan_abl.so
echo dep_gan_im_amd/libdepgan_abl.so
