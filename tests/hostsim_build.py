"""Builds tests/_build/libosqp_hostsim.so: the product's host driver (engine*.cpp + api.cpp) linked against the
plain-loop device-op simulator tests/hostsim/backend_host.cpp.  TEST INFRASTRUCTURE ONLY -- lets the CPU test tier
exercise the driver / front-end without a GPU.  The package never loads this library."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'osqp-python_amd', 'csrc')
OUT = os.path.join(ROOT, 'tests', '_build', 'libosqp_hostsim.so')
SRCS = [os.path.join(CSRC, 'engine.cpp'), os.path.join(CSRC, 'engine_setup.cpp'), os.path.join(CSRC, 'engine_api.cpp'), os.path.join(CSRC, 'api.cpp'), os.path.join(ROOT, 'tests', 'hostsim', 'backend_host.cpp')]
DEPS = SRCS + [os.path.join(CSRC, 'engine.hpp'), os.path.join(CSRC, 'engine_internal.hpp'), os.path.join(CSRC, 'backend.h'), os.path.join(CSRC, 'policy.h'), os.path.join(ROOT, 'include', 'osqp_hip.h')]


def build():
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(f) for f in DEPS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-Wall', '-Wextra', '-shared', '-fPIC', '-o', OUT] + SRCS)
    return OUT


if __name__ == '__main__':
    print(build())
