"""In-tree build of the MI355X attention library and its torch binding.

    python llm-inference_amd/build.py [--force]

Produces (all git-ignored, all shipped to the GPU box by gpurun):
  llm-inference_amd/lib/libcmhsa_hip.so                       hand-written gfx950 kernels + C ABI (attention, GELU)
  llm-inference_amd/causal_multihead_self_attention.<abi>.so  torch binding / importable module
  tests/helpers/libcmhsa_hip_planenv.so                       TEST build of the same kernels with -DCMHSA_PLAN_ENV: reads
                                                              CMHSA_WAVES/CHAIN/TABLE/VERBOSE to force a plan (the shipped
                                                              library reads no environment variable)
  tests/helpers/libcmhsa_testhelpers.so                       test helper kernels (LDS poisoning)

The reference builds its extension with torch.utils.cpp_extension.CUDAExtension
(reference torch_extensions/causal_multihead_self_attention/setup.py:4-12); here the kernel
library is plain `hipcc --offload-arch=gfx950` (no hipify pass, no torch headers), and the
binding is one host-only C++ file compiled against torch's headers.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INCLUDE = os.path.join(ROOT, "include")
ROCM = os.environ.get("ROCM_HOME", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

KERNEL_LIB = os.path.join(LIBDIR, "libcmhsa_hip.so")
TEST_HELPERS = os.path.join(ROOT, "tests", "helpers")
PLANENV_LIB = os.path.join(TEST_HELPERS, "libcmhsa_hip_planenv.so")
TESTHELPER_LIB = os.path.join(TEST_HELPERS, "libcmhsa_testhelpers.so")
EXT_NAME = "causal_multihead_self_attention"
EXT_PATH = os.path.join(HERE, EXT_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd: list[str]) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _kernel_sources() -> tuple[list[str], list[str]]:
    hip = [os.path.join(CSRC, "cmhsa_fwd.hip"), os.path.join(CSRC, "cmhsa_fwd16.hip"), os.path.join(CSRC, "cmhsa_decode.hip"), os.path.join(CSRC, "gelu_fwd.hip"), os.path.join(CSRC, "add_layernorm_fwd.hip"), os.path.join(CSRC, "gpt2_step.hip"), os.path.join(CSRC, "linear_rows_fwd.hip")]
    headers = [os.path.join(INCLUDE, "cmhsa.h"), os.path.join(INCLUDE, "gpt2_gelu.h"), os.path.join(INCLUDE, "gpt2_layernorm.h"), os.path.join(INCLUDE, "gpt2_step.h"), os.path.join(INCLUDE, "gpt2_linear.h"), os.path.join(CSRC, "gelu_one.h"),
               os.path.join(CSRC, "cmhsa_common.h"), os.path.join(CSRC, "cmhsa_plan.h"), os.path.join(CSRC, "cmhsa_tiles16.h")]
    return hip, headers


OBJDIR = os.path.join(HERE, "build")              # git-ignored; objects only, the libraries live where they are loaded from


def _compile_objects(jobs: list[tuple[str, str, list[str]]]) -> None:
    """(source, object, extra flags) -> hipcc -c, all at once: the files are independent and the few-row linear kernels
    alone take a minute (dozens of template instantiations)."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJDIR, exist_ok=True)
    cmds = [[HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", *flags, "-I" + INCLUDE, "-o", obj, src] for src, obj, flags in jobs]
    if not cmds:
        return
    with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, min(8, os.cpu_count() or 1)))) as pool:
        list(pool.map(_run, cmds))


def _obj(src: str, tag: str = "") -> str:
    return os.path.join(OBJDIR, os.path.splitext(os.path.basename(src))[0] + tag + ".o")


def build_kernel_lib(force: bool = False, with_test_build: bool = False) -> str:
    """libcmhsa_hip.so, and with_test_build also tests/helpers/libcmhsa_hip_planenv.so: the same objects except cmhsa_fwd,
    which is compiled a second time with -DCMHSA_PLAN_ENV (the only file that reads the macro)."""
    hip, headers = _kernel_sources()
    jobs = [(src, _obj(src), []) for src in hip if force or _stale(_obj(src), [src] + headers)]
    fwd = hip[0]
    if with_test_build and (force or _stale(_obj(fwd, "_planenv"), [fwd] + headers)):
        jobs.append((fwd, _obj(fwd, "_planenv"), ["-DCMHSA_PLAN_ENV"]))
    _compile_objects(jobs)
    objs = [_obj(src) for src in hip]
    if force or _stale(KERNEL_LIB, objs):
        os.makedirs(LIBDIR, exist_ok=True)
        _run([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", KERNEL_LIB, *objs])
    if with_test_build:
        test_objs = [_obj(fwd, "_planenv")] + objs[1:]
        if force or _stale(PLANENV_LIB, test_objs):
            _run([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", PLANENV_LIB, *test_objs])
    return KERNEL_LIB


def build_test_libs(force: bool = False) -> None:
    """Test-only libraries (never loaded by the product package): the plan-forcing build and the helper kernels."""
    build_kernel_lib(force, with_test_build=True)
    helper_src = [os.path.join(TEST_HELPERS, "lds_poison.hip")]
    if force or _stale(TESTHELPER_LIB, helper_src):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", TESTHELPER_LIB, *helper_src])


def build_torch_binding(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, "torch_binding.cpp"), *(os.path.join(INCLUDE, h) for h in ("cmhsa.h", "gpt2_gelu.h", "gpt2_layernorm.h", "gpt2_step.h", "gpt2_linear.h")), KERNEL_LIB]
    if force or _stale(EXT_PATH, srcs):
        import torch
        from torch.utils import cpp_extension

        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        inc = ["-I" + p for p in cpp_extension.include_paths()]
        abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared",
              "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
              "-I" + INCLUDE, "-I" + os.path.join(ROCM, "include"), "-I" + sysconfig.get_paths()["include"],
              *inc, srcs[0], "-o", EXT_PATH,
              "-L" + LIBDIR, "-lcmhsa_hip", "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
              "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath," + tlib, "-Wl,--no-as-needed"])
    return EXT_PATH


def build_all(force: bool = False) -> None:
    build_kernel_lib(force, with_test_build=True)      # every object in one parallel batch
    build_torch_binding(force)
    build_test_libs(False)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
