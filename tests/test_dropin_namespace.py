"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in import path and the
fixture generator, next to the reference.

``models/`` of this repository has NO ``__init__.py`` anywhere: like the reference's ``models/`` it is a PEP 420 namespace portion.  With
this repository in front of the reference on ``sys.path``, ``models.core.{corr,ppmstereo,ppmtereo_update}`` and
``models.core.utils.utils`` bind to the gfx950 implementation while everything else the reference keeps under ``models/`` -- the wrapper
``models/ppm_stereo_model.py`` (its line 12 is the drop-in's import site), ``models.core.extractor``, ``attention``, ``convnext`` ... --
still resolves in the reference.  ``tools/gen_golden.py`` puts the reference FIRST and refuses to run if any module it pins the
oracle with was imported from this repository."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models", "core")), reason="reference tree not present on this box")


def _run(code, path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=os.pathsep.join(path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_models_is_a_namespace_portion():
    for d, _, files in os.walk(os.path.join(ROOT, "models")):
        assert "__init__.py" not in files, f"{d}/__init__.py would hide the reference's models namespace"


@needs_ref
def test_wrapper_file_and_shim_resolve_side_by_side():
    """models/ppm_stereo_model.py:12 -- found under the reference; its import target -- found here."""
    code = """
import importlib.util as u, os
want = {"models.ppm_stereo_model": %r, "models.core.extractor": %r, "models.core.attention": %r, "models.core.convnext": %r,
        "models.core.ppmstereo": %r, "models.core.corr": %r, "models.core.ppmtereo_update": %r, "models.core.utils.utils": %r}
for name, root in want.items():
    spec = u.find_spec(name)
    assert spec is not None and spec.origin, name
    assert os.path.realpath(spec.origin).startswith(root + os.sep), (name, spec.origin)
from models.core.ppmstereo import PPMStereo            # the line of the wrapper, executed
import ppmstereo_amd.ppmstereo as P
assert PPMStereo is P.PPMStereo
print("ok")
""" % (REF, REF, REF, REF, ROOT, ROOT, ROOT, ROOT)
    assert "ok" in _run(code, [ROOT, REF])


@needs_ref
def test_reference_first_binds_the_reference():
    """The generator's order: the reference in front -> the same names are the reference's own files."""
    code = """
import importlib.util as u, os
for name in ("models.core.ppmstereo", "models.core.corr", "models.core.ppmtereo_update", "models.core.utils.utils"):
    assert os.path.realpath(u.find_spec(name).origin).startswith(%r + os.sep), name
print("ok")
""" % REF
    assert "ok" in _run(code, [REF, ROOT])


@needs_ref
def test_fixture_generator_reproduces_every_committed_fixture(tmp_path):
    """`python tools/gen_golden.py` as committed (~40 s): runs the reference's own modules and reproduces ALL committed fixtures bit for bit --
    the recipe that pins the oracle works, and tests/golden/ holds nothing but its output."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "--out", str(tmp_path)],
                       capture_output=True, text=True, cwd=ROOT, timeout=1500, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(f for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.endswith(".npz"))
    assert sorted(os.listdir(tmp_path)) == committed and len(committed) >= 29
    for f in committed:
        new, old = np.load(tmp_path / f), np.load(os.path.join(ROOT, "tests", "golden", f))
        assert set(new.files) == set(old.files), f
        for k in new.files:
            assert np.array_equal(new[k], old[k], equal_nan=True), (f, k)
