"""The native operator library (mmf_amd/csrc/torch_ops.cpp -> libmmf_amd_ops.so) on the build container: it loads with
torch.ops.load_library in an interpreter that never imports the Python package, registers every operator schema the package declares
(same strings: a scripted model must find the same signatures either way), and refuses host tensors with a TORCH_CHECK message — no CPU
path, no compute without a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS_LIB = os.path.join(ROOT, "mmf_amd", "libmmf_amd_ops.so")

_PROBE = r"""
import json, sys, torch
torch.ops.load_library(sys.argv[1])
assert "mmf_amd" not in sys.modules
names = json.loads(sys.argv[2])
out = {}
for n in names:
    out[n] = str(getattr(torch.ops.mmf_amd, n).default._schema)
try:
    torch.ops.mmf_amd.additive_mask(torch.ones(2, 3, dtype=torch.long))
    out["_host"] = "accepted"
except RuntimeError as e:
    out["_host"] = str(e)
out["_state"] = [int(torch.ops.mmf_amd._set_py_mode(0)), bool(torch.ops.mmf_amd._ln_defer_set(False))]
print("JSON" + json.dumps(out))
"""


@pytest.mark.skipif(not os.path.exists(OPS_LIB), reason="libmmf_amd_ops.so not built (python -m mmf_amd.csrc.build)")
def test_ops_library_loads_alone_and_declares_the_package_schemas():
    import json

    from mmf_amd import ops
    schemas = ops.schemas()
    r = subprocess.run([sys.executable, "-c", _PROBE, OPS_LIB, json.dumps(sorted(schemas))], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("JSON")][-1][4:])
    for name, schema in schemas.items():
        want = "mmf_amd::" + schema.replace("Tensor? ", "Tensor? ")
        assert got[name].replace(" ", "") == want.replace(" ", ""), (name, got[name], want)
    assert "HBM" in got["_host"] and "CPU" in got["_host"]
    assert got["_state"] == [0, False]


def test_package_declares_the_operators_from_python_without_a_gpu():
    """Build container: no GPU, so the package must not have loaded the native library (dry runs go through kernel stubs)."""
    import torch

    from mmf_amd import _ops_native
    if torch.cuda.is_available():
        pytest.skip("GPU box")
    assert _ops_native.NATIVE is False
