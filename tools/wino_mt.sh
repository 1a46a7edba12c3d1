# Winograd kernel, 16-row tiles (MT=2) against 8-row tiles (MT=1), next to the direct kernel: two rocprofv3 traces
R=$PWD
for mt in 2 1; do
  rm -rf $R/gpurun_out/wmt$mt; cd /tmp; export TMPDIR=/tmp
  DEPGAN_WINO_MT=$mt timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wmt$mt -o t -- \
    python3 $R/tools/time_wino.py run > $R/gpurun_out/wmt$mt.log 2>&1 || { tail -5 $R/gpurun_out/wmt$mt.log; exit 1; }
  cd $R; echo "DEPGAN_WINO_MT=$mt"; python tools/time_wino.py parse gpurun_out/wmt$mt
done
