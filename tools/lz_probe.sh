# the GRU convs of the 1/4 scale with / without the skipped lo-plane products, for the generator's products-per-branch settings (GPU box)
for g in 1 2 4 8; do
  PPMS_CONV5_SKIPGRP=$g python tools/gen_conv5_asm.py > /dev/null
  python -m ppmstereo_amd.build > /dev/null 2>&1
  echo "--- PPMS_CONV5_SKIPGRP=$g"
  PROBE_LOZERO=256 python tools/conv_probe.py zr1_0,q1,zr2,q2,zr3 30 2>&1 | grep -v amdgpu.ids
  python tools/conv_probe.py zr1_0,zr2 30 2>&1 | grep -v amdgpu.ids
done
python tools/gen_conv5_asm.py > /dev/null
