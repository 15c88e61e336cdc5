"""-m "not gpu": the C-ABI library loads, exports every declared symbol and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from alego_amd import binding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "alego_mi355x.h")).read()
    declared = set(re.findall(r"\b(alego_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"alego_default_params"}  # static inline in alego_params.h
    L = binding.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libalego_mi355x.so lacks {missing}"
    assert set(binding.EXPORTS) == declared, (set(binding.EXPORTS) ^ declared)


def test_params_struct_matches_header():
    assert binding.lib().alego_params_sizeof() == C.sizeof(binding.AlegoParams)
    assert synth.lib().alego_synth_params_sizeof() == C.sizeof(binding.AlegoParams)


def test_no_cpu_fallback(params_a):
    """Without a gfx950 device alego_create must refuse (no oracle / CPU path behind the product API)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert binding.lib().alego_device_count() == 0
    with pytest.raises(binding.AlegoError):
        binding.Handle(params_a)


def test_product_does_not_link_the_oracle():
    """The shipped library and package never reference oracle/ (only tests, smoke and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "a-lego-loam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("// oracle/", ""), f
    so = open(binding.lib_path(), "rb").read()
    assert b"liboracle" not in so
    # neither do the host-side sources around it: the development tools, the C++ example, the ROS adapters, the headers
    for sub in ("tools", "examples", "ros_adapter", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle_py" not in txt and "liboracle" not in txt and "from oracle" not in txt, os.path.join(sub, f)


def test_cpp_example_fails_loudly_without_a_gpu():
    """examples/replay.cpp drives the C ABI from plain C++.  In a container without an MI355X it must stop at alego_create with
    ALEGO_ERR_NO_DEVICE — there is no CPU fallback for the product path (on the GPU box tests/test_gpu_parity.py runs it for real)."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "replay")
    assert os.path.exists(exe), "__graft_entry__.build() compiles it"
    r = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=120)
    if r.returncode == 0:
        pytest.skip("a GPU is present")
    assert r.returncode == 1 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)
