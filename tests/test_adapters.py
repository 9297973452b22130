"""The C++ adapters (include/ydorb/*.hpp, reference signatures over the C ABI) type-check against declarations of the
reference types they touch.  OpenCV is absent here, so a declaration-only mock stands in for <opencv2/core.hpp> (test-only)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "tests", "cpu_harness")


def _syntax(src, *extra):
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I" + os.path.join(H, "mock"), *extra, os.path.join(H, src)])


def test_extractor_adapter_typechecks():
    _syntax("adapter_syntax_check_extractor.cpp")


def test_matcher_adapter_typechecks():
    _syntax("adapter_syntax_check.cpp")


def test_optimizer_adapter_typechecks():
    eigen = "/root/reference/thirdParty/eigen"
    if not os.path.isdir(eigen):
        pytest.skip("Eigen headers (the reference's vendored copy) are not on this machine")
    _syntax("adapter_syntax_check.cpp", "-DYDORB_CHECK_OPTIMIZER", "-I" + eigen)
