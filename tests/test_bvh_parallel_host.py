"""The host BVH builder's parallel form (zyg_amd/csrc/host/bvh_builder.hpp; the reference builds sub-trees as parallel tasks and merges
them in task order, builder_base.zig:83-95, 330-390): the same node and primitive arrays as the plain depth-first recursion, byte for
byte, for inputs below and above the thresholds at which the candidates of a node and the sub-trees are spread over threads
(tests/cpp/bvh_parallel_check.cpp: random references with long thin ones, so that spatial splits duplicate)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,threads", [(5000, 3), (40000, 8), (150000, 5)])
def test_parallel_build_is_the_serial_build(tmp_path, n, threads):
    exe = tmp_path / "bvh_parallel_check"
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-pthread", os.path.join(ROOT, "tests", "cpp", "bvh_parallel_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), str(n), str(threads)], capture_output=True, text=True)
    assert 0 == r.returncode and r.stdout.startswith("OK"), (r.returncode, r.stdout, r.stderr)
