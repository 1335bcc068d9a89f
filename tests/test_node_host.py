"""The Node host (raytracer_amd/host): scene layer on CPU, full path through the N-API shim on GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracer_amd", "host")
node = shutil.which("node")
needs_node = pytest.mark.skipif(node is None, reason="node is not installed on this machine")


@needs_node
def test_host_presets_flatten_to_the_reference_scene_records():
    r = subprocess.run([node, os.path.join(HOST, "test", "flatten_check.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": ok") == 17          # the thirteen scenes of rayphoton.js and the four of ray.js


@needs_node
def test_addon_loads_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    code = ("const rt=require('./index.js'); try { rt.context(); console.log('NOERR'); } "
            "catch(e) { console.log('ERR', e.message); }")
    r = subprocess.run([node, "-e", code], cwd=HOST, capture_output=True, text=True, timeout=120)
    assert "ERR" in r.stdout and "no CPU path" in r.stdout, r.stdout + r.stderr


@needs_node
@pytest.mark.gpu
def test_node_host_end_to_end_on_gpu():
    r = subprocess.run([node, os.path.join(HOST, "test", "gpu_check.js")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("PASS") >= 10 and "FAIL" not in r.stdout
