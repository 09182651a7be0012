"""bench.py under the A/B switches of tools/ab_switches.py (PPMS_* environment variables -> ppmstereo_amd.engine.TUNING).
usage (GPU box): PPMS_CONV5_PAD2X=0 python tools/ab_bench.py --steps 20 --no-cpu-baseline --no-encoders"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_switches  # noqa: F401,E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv[0] = os.path.join(root, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
