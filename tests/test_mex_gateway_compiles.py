"""The MEX gateway a MATLAB maintainer would compile (codes_of_ipd_ssn_amg_method_amd/mex/ipd_mex.cpp)
cannot be built here (no MATLAB, no mex.h).  What CAN be checked without MATLAB: that it is valid
C++ against the documented signatures of the MATLAB C API it uses (tests/mex_stub/mex.h declares
them; `g++ -fsyntax-only`, nothing is linked or run) and against include/ipd_amg.h, and that
every command the same-named .m shims send is one the gateway dispatches.  CPU only."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEX = os.path.join(ROOT, "codes_of_ipd_ssn_amg_method_amd", "mex")


def test_gateway_is_valid_cpp_against_the_mex_api():
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type",
                          "-I" + os.path.join(ROOT, "tests", "mex_stub"), "-I" + os.path.join(ROOT, "include"),
                          os.path.join(MEX, "ipd_mex.cpp")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_every_shim_command_is_dispatched():
    src = open(os.path.join(MEX, "ipd_mex.cpp")).read()
    sent = set()
    for path in glob.glob(os.path.join(MEX, "*.m")):
        sent |= set(re.findall(r"ipd_mex\('([A-Za-z0-9_]+)'", open(path).read()))
    assert sent, "no shims found"
    missing = sorted(c for c in sent if '"%s"' % c not in src)
    assert not missing, "commands the shims send but the gateway does not know: %s" % missing
