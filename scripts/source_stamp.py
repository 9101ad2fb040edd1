"""sha256 over the sources that decide what the recorded counters describe (kernels, launch logic, tree layout): the
profile makers stamp it into profiles/*_pmc_{traffic,valu}.json; bench.py and tests/test_profiles.py compare it with
the tree to flag a stale recording."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMPED = ("path-tracer_amd/csrc/hpt_kernels.hip", "path-tracer_amd/csrc/hpt_api.cpp", "path-tracer_amd/csrc/hpt_bvh.cpp",
           "path-tracer_amd/csrc/hpt_device.h")


def source_sha256(root=ROOT):
    h = hashlib.sha256()
    for rel in STAMPED:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(rel.encode()); h.update(f.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_sha256())
