"""The optional HIP runtime configuration of the package (diffusiontexturepainting_amd/__init__.py): off by default, applied on
import with DTP_RUNTIME_ENV=1, a variable the user has set is left alone.  Each case in a fresh interpreter (the package applies it at import)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = ("import os, sys; sys.path.insert(0, %r); import diffusiontexturepainting_amd as d; "
         "print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE', 'unset'), d.RUNTIME_ENV_APPLIED, 'torch' in sys.modules)" % ROOT)


def _run(**env):
    e = {k: v for k, v in os.environ.items() if k not in ("DTP_RUNTIME_ENV", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
    e.update(env)
    out = subprocess.run([sys.executable, "-c", PROBE], env=e, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return out.stdout.split()


def test_off_by_default_and_importing_the_package_does_not_import_torch():
    assert _run() == ["unset", "False", "False"]
    assert _run(DTP_RUNTIME_ENV="0") == ["unset", "False", "False"]


def test_opt_in_sets_the_variable_before_anything_touches_hip():
    assert _run(DTP_RUNTIME_ENV="1") == ["0", "True", "False"]


def test_a_variable_the_user_has_set_wins():
    assert _run(DTP_RUNTIME_ENV="1", DEBUG_CLR_GRAPH_PACKET_CAPTURE="1") == ["1", "True", "False"]
