"""MI355X-native stamp-inpainting engine (see README.md / DESIGN.md).

Optional HIP runtime configuration, $DTP_RUNTIME_ENV=1 (off by default).  ROCm 7 replays the kernel nodes of an instantiated hipGraph from
pre-built AQL packets (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default).  Through the ordinary dispatch path the device finishes a stamp's
4 400-4 700 dependent launches sooner -- same box, arms interleaved (profiles/r06_runtime_knobs*.txt, r06_graph_vs_eager.txt): -1.0 .. -1.5 %
per 512^2 stamp, -2.4 % per 256^2 / 20-step stamp, nothing at batch 8 or at 256^2 / 8 steps -- but the HOST then pays 2.7 us of CPU per node and, above
~2 400 launches in flight, blocks on the queue: dtp_stamp returns about half-way through a stamp instead of after 1 ms (two back-to-back 512^2
stamps: after 138 of 190 ms against 2 of 196 ms; profiles/r06_host_enqueue.txt, r06_launch_host_cost.txt).  That gives up the "only enqueues" property of include/dtp.h for 1-2 %, so it is a choice for a caller who has nothing
else to do with the thread (the reference's synchronous handler), not the default.  With DTP_RUNTIME_ENV=1 importing this package sets
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 unless the user has set it -- which only works BEFORE the runtime initialises (ROCclr reads its flags once,
at the process's first HIP call): import the package above `import torch` (bench.py, __graft_entry__.py and tests/conftest.py do), or export
the variable in the shell."""
import os as _os
import sys as _sys

RUNTIME_ENV = {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}


def apply_runtime_env():
    """With $DTP_RUNTIME_ENV=1: set the variables of RUNTIME_ENV that the user has not set.  Returns True when they were applied in time,
    False when nothing was requested or the HIP runtime of this process is known to be initialised already (torch.cuda initialised:
    the setting then only reaches child processes)."""
    if _os.environ.get("DTP_RUNTIME_ENV", "") in ("", "0"):
        return False
    late = False
    t = _sys.modules.get("torch")
    if t is not None:
        try:
            late = bool(t.cuda.is_initialized())
        except Exception:
            late = False
    for k, v in RUNTIME_ENV.items():
        _os.environ.setdefault(k, v)
    return not late


RUNTIME_ENV_APPLIED = apply_runtime_env()
