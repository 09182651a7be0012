"""GPU: the driver's command -- `python bench.py` at N = 1 on BASELINE config 2 -- prints ONE JSON line that carries the contract's fields: the metric and
its configuration, `roofline` (dominant kernel: bound, achieved, peak, unit, frac, traffic), `cpu_baseline` (value, unit, cores, kind, sample), and the
library that ran.  A short run (2 steps): the numbers are not asserted beyond consistency, the shape of the line is."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_n1_line_has_the_contract_fields():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PPMS_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-encoders"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["metric"] == "disparity-px/s" and out["unit"] == "disparity-px/s" and out["higher_is_better"] is True
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["data"] == "synthetic" and out["dtype"] == "bf16"
    cfg = out["config"]
    assert "BASELINE config 2" in cfg["workload"] and (cfg["T"], cfg["H"], cfg["W"], cfg["iters"]) == (5, 320, 512, 10) and "model" not in cfg
    px = 5 * 320 * 512
    assert abs(out["value"] - px / (out["ms_per_step"] * 1e-3)) <= 2e-3 * out["value"]          # value = pixels of K steps / their wall time
    for key in ("roofline", "roofline_2", "roofline_3", "roofline_hbm"):
        rf = out[key]
        assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and rf["peak"] > 0 and "traffic" in rf
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3 and 0 < rf["frac"] < 1
    assert "conv6_kernel" in out["roofline"]["kernel"] or "memory attention" in out["roofline"]["kernel"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "disparity-px/s" and cb["value"] > 0 and 1 <= cb["cores"] <= 16 and "iterations" in cb["sample"]
    assert out["library"] == "ppmstereo_amd/libppms.so" and out["build_mode"] in ("reused", "compiled") and out["library_stamp_matches_sources"] is True
    assert out["sharded"] is None and out["sharded_check"] is None
