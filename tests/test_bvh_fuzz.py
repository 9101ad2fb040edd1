"""The BVH node array is the one structured input the host hands over (hpt_bvh_node[], bvh.rs:13-23): malformed
arrays must be rejected, never crash.  tests/native/bvh_fuzz.cpp mutates valid pre-order trees (child / prim indices,
truncated arrays) and pushes 20 000 of them through tree_from_reference_nodes + flatten_mesh, and the SAH builder
through degenerate and NaN inputs, all under AddressSanitizer + UBSan (host code only)."""
import os
import subprocess

from conftest import ROOT


def test_bvh_validation_under_sanitizers(tmp_path):
    src = os.path.join(ROOT, "path-tracer_amd", "csrc")
    exe = str(tmp_path / "bvh_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread", "-I", src, "-o", exe,
                           os.path.join(ROOT, "tests", "native", "bvh_fuzz.cpp"), os.path.join(src, "hpt_bvh.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    acc, rej = [int(x) for x in r.stdout.split() if x.isdigit()]
    assert acc > 1000 and rej > 1000
