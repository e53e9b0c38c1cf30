"""The task list of the pipelined panel factorisation (``stheno_amd/csrc/gpk_potrf_pipe.hpp``) against a host model of the chain
workgroup, the progress words and the counters: ``stheno_amd/csrc/pipe_check.cpp`` plays the list in order for a few thousand panel
shapes and fails on any task whose dependencies are not satisfied by earlier tasks (= a possible deadlock on the device), on a piece
that ends up with the wrong number of updates, or on a chain that cannot finish.  Plain C++, no GPU."""
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stheno_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host C++ compiler")
def test_task_list_is_deadlock_free_and_complete(tmp_path):
    exe = tmp_path / "pipe_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-parentheses", "-o", str(exe), os.path.join(CSRC, "pipe_check.cpp")], check=True, cwd=CSRC)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "shapes OK" in out.stdout
