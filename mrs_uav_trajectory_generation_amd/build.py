"""Build recipe for the HIP extension (libmrs_tg.so, in-tree next to this file).

    python -m mrs_uav_trajectory_generation_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present.  The .so is git-ignored but travels to the
GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmrs_tg.so")
SOURCES = ["mrs_tg_kernels.hip", "mrs_tg_tile.hip", "mrs_tg_rows.hip", "mrs_tg_quad.hip", "mrs_tg_general.hip", "mrs_tg_nonlinear.hip", "mrs_tg_wave.hip", "mrs_tg_dfo.hip", "mrs_tg_abi.hip", "mrs_tg_multi.hip", "mrs_tg_policy.hip", "mrs_tg_policy_dev.hip",
           "mrs_tg_pool.hip"]
# every header under csrc/ (a header that is split or added is picked up without editing this file) + the public ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))) + [os.path.join("..", "..", "include", "mrs_tg.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# the careful re-run of MRS_TG_FLAG_CAREFUL_COST (optimize_careful_kernel) is built in unless MRS_TG_WITH_CAREFUL=0 is set in the
# environment of the build (mrs_tg_capabilities() reports which library is loaded)
FLAGS.append("-DMRS_TG_WITH_CAREFUL=%d" % (0 if os.environ.get("MRS_TG_WITH_CAREFUL", "1") == "0" else 1))


def _hipcc():
    env = os.environ.get("HIPCC")
    if env:
        return env
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra_flags=(), lib=None, objdir_name="build"):
    """extra_flags / lib / objdir_name: experiment builds (scripts/variants.sh): another -D set into another .so"""
    lib = lib or LIB
    if lib == LIB and not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, objdir_name)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m ...build --variant NAME -DX=1 ... -> libmrs_tg_NAME.so (api: MRS_TG_LIB_PATH)
        i = sys.argv.index("--variant")
        name = sys.argv[i + 1]
        print(build(force=True, verbose=False, extra_flags=sys.argv[i + 2:], lib=os.path.join(HERE, "libmrs_tg_%s.so" % name),
                    objdir_name="build_" + name))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
