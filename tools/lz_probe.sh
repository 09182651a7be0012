# the GRU convs of the 1/4 scale with / without the skipped lo-plane products, for the generator's products-per-branch settings (GPU box)
# whatever happens, the committed (default) header is back in the tree when the script ends: it is part of the library digest
trap 'env -u PPMS_CONV5_SKIPGRP python tools/gen_conv5_asm.py > /dev/null' EXIT
for g in 1 2 4 8; do
  PPMS_CONV5_SKIPGRP=$g python tools/gen_conv5_asm.py > /dev/null || exit 1
  python -m ppmstereo_amd.build > /dev/null 2>&1 || exit 1
  echo "--- PPMS_CONV5_SKIPGRP=$g"
  PROBE_LOZERO=256 python tools/conv_probe.py zr1_0,q1,zr2,q2,zr3 30 2>&1 | grep -v amdgpu.ids
  python tools/conv_probe.py zr1_0,zr2 30 2>&1 | grep -v amdgpu.ids
done
