# timing ablations / variants of a kernel: builds the library with each define set and times a few convs (the define must be
# in the environment of the measuring process too: it is part of the build digest)
for a in "$@"; do
  export PPMS_BUILD_DEFINES="$a"
  python -m ppmstereo_amd.build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
  echo "[$a]: $(python tools/dvfs_probe.py 2>&1 | grep 'zr1_0\|fh1\|q1' | cut -c1-48 | tr '\n' ' ')"
done
