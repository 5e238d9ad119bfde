"""The C-ABI shared library without a GPU: it loads, exports every symbol include/mrs_tg.h declares,
agrees with the header's constants, and fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

from mrs_uav_trajectory_generation_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()          # hipcc cross-compiles for gfx950 without a GPU
    return api.load_library()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "mrs_tg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrs_tg_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_abi_version_and_default_options(lib):
    assert lib.mrs_tg_abi_version() == 5
    opt = api.default_options()
    # defaults follow the reference's parameters (src/mrs_trajectory_generation.cpp:884-885,
    # config/private/trajectory_generation.yaml:10)
    assert (opt.f_rel, opt.x_rel, opt.max_iterations) == (0.05, 0.1, 10)
    assert opt.f_abs == -1.0 and opt.x_abs == -1.0 and opt.time_alloc_method == -1
    assert C.sizeof(api.Options) == 120 and opt.max_time_s == 0.0 and opt.flags == 0
    # (ABI 5) the length check of the single-path seam: config/public/trajectory_generation.yaml:35-36
    assert (opt.max_trajectory_len_factor, opt.min_trajectory_len_factor) == (3.0, 0.33)
    assert (opt.time_penalty, opt.soft_constraint_weight, opt.use_soft_constraints, opt.initial_stepsize_rel) == (100.0, 1.5, 1, 0.1)


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; this test covers the GPU-less container")
    with pytest.raises(api.MrsTgError, match="no CPU fallback"):
        api.Context(0)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mrs_uav_trajectory_generation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "mrs_tg_oracle" not in src and "libmrs_tg_oracle" not in src, f


def test_cpp_wrapper_compiles_and_links(lib, tmp_path):
    """include/mrs_tg.hpp (the findTrajectory-shaped C++ wrapper of INTEGRATION.md) builds against the library."""
    import subprocess
    import torch
    src = tmp_path / "host.cpp"
    src.write_text(r"""
#include <cstdio>
#include "mrs_tg.hpp"
int main() {
  try {
    mrs_tg::TrajectoryGenerator tg(0);
    std::vector<mrs_tg::Waypoint> wps = {{{-5, -5, 5, 1}, false}, {{-5, 5, 5, 2}, false}, {{5, -5, 5, 3}, false}, {{5, 5, 5, 4}, false}};
    mrs_tg::DynamicsConstraints dc{2, 2, 20, 2, 2, 2, 2, 20, 20, 1, 2, 20};
    tg.options().derivative_to_optimize = 4;
    auto pts = tg.findTrajectory(wps, std::nullopt, dc, 0.2, false);
    std::printf("samples %zu status %d\n", pts ? pts->size() : 0, tg.status());
    return pts ? 0 : 2;
  } catch (const std::exception& e) {
    std::printf("error: %s\n", e.what());
    return 3;
  }
}
""")
    exe = tmp_path / "host"
    libdir = os.path.dirname(api.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-l:libmrs_tg.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-L/opt/rocm/lib", "-lamdhip64"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "samples" in r.stdout, r.stdout + r.stderr
    else:
        # no device: the constructor throws the library's loud error, nothing falls back to the CPU
        assert r.returncode == 3 and "no CPU fallback" in r.stdout, r.stdout + r.stderr


def test_python_constants_are_the_headers():
    """every MRS_TG_FLAG_* / MRS_TG_CAP_* / MRS_TG_ERR_* / MRS_TG_TIME_ALLOC_* of include/mrs_tg.h has the same value in the
    ctypes mirror (api.py) -- a flag added on one side only would silently be another option"""
    import re
    from mrs_uav_trajectory_generation_amd import api
    text = open(os.path.join(ROOT, "include", "mrs_tg.h")).read()
    seen = 0
    for name, value in re.findall(r"\bMRS_TG_((?:FLAG|CAP|TIME_ALLOC)_[A-Z0-9_]+)\s*=\s*(-?\d+)", text):
        assert hasattr(api, name), name
        assert getattr(api, name) == int(value), (name, getattr(api, name), value)
        seen += 1
    assert seen >= 10, seen
