# ablation / variant builds of the line-resident pyramid build (timing only): bash tools/abl_corr.sh "-DCORR_ABL=1" "-DCORR_FORCE64" ...
trap 'unset PPMS_BUILD_DEFINES; python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1' EXIT
for a in "$@"; do
  export PPMS_BUILD_DEFINES="$a"
  python -c "from ppmstereo_amd import build; build.build()" > /dev/null 2>&1 || exit 1
  echo "$a: $(timeout -k 10 100 python tools/corr_probe.py ${GEO:-5 80 128} 2>&1 | tail -1)"
done
