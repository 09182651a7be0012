# A/B of the two attention kernels (32x32x16 vs 16x16x32 MFMA) on one box: stand-alone 1/4-scale call (tools/attn_probe.py), alternating
for i in 1 2; do
  for d in "-DPPMS_ATTN_SHAPE32" ""; do
    PPMS_BUILD_DEFINES="$d" python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 &&
    echo "defines '$d': $(PPMS_BUILD_DEFINES="$d" timeout -k 10 100 python tools/attn_probe.py 15 1 2>&1 | tail -1)" || exit 1
  done
done
