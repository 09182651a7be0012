for i in 1 2; do
python tools/conv_probe.py zr1_0,q1,zr2,q2,zr3,q3 30
PROBE_LOZERO=256 python tools/conv_probe.py zr1_0,q1,zr2,q2,zr3,q3 30
done
