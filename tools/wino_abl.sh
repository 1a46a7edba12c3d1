# Ablation table of the Winograd kernel on the GPU box: kernel times per shape with parts of the kernel removed.
# needs dep_gan_im_amd/libdepgan_abl.so (tools/build_wino_abl.sh).  usage: bash tools/wino_abl.sh "0 1 2 4 8"
R=$PWD
for abl in ${1:-0 1 2 3 8 10 11}; do
  rm -rf $R/gpurun_out/wabl; cd /tmp; export TMPDIR=/tmp
  WINO_ONLY=1 DEPGAN_LIB=$R/dep_gan_im_amd/libdepgan_abl.so DEPGAN_WINO_ABL=$abl timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv \
    -d $R/gpurun_out/wabl -o t -- python3 $R/tools/time_wino.py run > $R/gpurun_out/wabl_$abl.log 2>&1 || { tail -5 $R/gpurun_out/wabl_$abl.log; exit 1; }
  cd $R; echo "ABL=$abl"; WINO_ONLY=1 python tools/time_wino.py parse gpurun_out/wabl
done
