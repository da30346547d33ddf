#!/usr/bin/env python3
"""oracle/build_ref.py -- TEST INFRASTRUCTURE ONLY.

Builds the reference C++ `ugs_sampler` pybind11 module from the sources where they lie under
/root/reference/src/samplers/ugs_sampler (see oracle/ref_unity.cpp for the recipe and the one
allocator-flag deviation) into oracle/_ref/.  Nothing from the reference is copied into the repo;
oracle/_ref/ is git-ignored.  The reference's own build system (setup.py / ninja) is not run: this
is one direct g++ invocation.

The built module is used only
  * by tests/ and tools under oracle/ to pin the C restatement (oracle/ugs_oracle.c) and to generate
    the golden fixtures under tests/golden/ (oracle/make_golden.py), and
  * optionally by bench.py's `cpu_baseline` leg (kind "reference").
It is never imported by the product package.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/samplers/ugs_sampler"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "ugs_sampler" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False) -> str:
    if not os.path.isdir(REF):
        raise FileNotFoundError(f"{REF} not present (the reference does not travel to the GPU box)")
    srcs = [os.path.join(HERE, "ref_unity.cpp")] + [
        os.path.join(REF, "src", f)
        for f in ("preproc.cpp", "sampler.cpp", "ugs_sampler_batch_extension.cpp", "extension.cpp")
    ] + [os.path.join(REF, "include", f) for f in ("sampler.hpp", "cache.hpp")]
    if not force and os.path.exists(OUT) and all(
        os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs
    ):
        return OUT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-w",
           "-DTORCH_EXTENSION_NAME=ugs_sampler", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{REF}", f"-I{REF}/include"]
    for p in ce.include_paths():
        cmd.append(f"-I{p}")
    cmd += [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}"]
    cmd += [os.path.join(HERE, "ref_unity.cpp"), "-o", OUT]
    for p in ce.library_paths():
        cmd += [f"-L{p}", f"-Wl,-rpath,{p}"]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    subprocess.run(cmd, check=True)
    return OUT


def load():
    """Import the built reference module under the private name `ugs_sampler` (spec-loaded, not on sys.path)."""
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    path = OUT
    if not os.path.exists(path):
        path = build()
    spec = importlib.util.spec_from_file_location("ugs_sampler", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


EPS_REF = "/root/reference/src/samplers/epsilon_uniform_sampler"
EPS_OUT = os.path.join(OUT_DIR, "epsilon_uniform_sampler" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_eps(force: bool = False) -> str:
    """The reference epsilon_uniform_sampler module (one translation unit, OpenMP), same allocator-flag deviation as above
    (oracle/ref_eps_unity.cpp).  Used only by tests/test_eps_oracle.py to pin oracle/eps_oracle.py statistically."""
    if not os.path.isdir(EPS_REF):
        raise FileNotFoundError(EPS_REF)
    unity = os.path.join(HERE, "ref_eps_unity.cpp")
    srcs = [unity, os.path.join(EPS_REF, "src", "epsilon_uniform_sampler.cpp"), os.path.join(EPS_REF, "include", "epsilon_uniform_sampler.hpp")]
    if not force and os.path.exists(EPS_OUT) and all(os.path.getmtime(EPS_OUT) >= os.path.getmtime(x) for x in srcs):
        return EPS_OUT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-w", "-fopenmp", "-DTORCH_EXTENSION_NAME=epsilon_uniform_sampler",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{EPS_REF}", f"-I{EPS_REF}/include"]
    for p in ce.include_paths():
        cmd.append(f"-I{p}")
    cmd += [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}", unity, "-o", EPS_OUT]
    for p in ce.library_paths():
        cmd += [f"-L{p}", f"-Wl,-rpath,{p}"]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    subprocess.run(cmd, check=True)
    return EPS_OUT


def load_eps():
    import importlib.util

    import torch  # noqa: F401

    path = EPS_OUT if os.path.exists(EPS_OUT) else build_eps()
    spec = importlib.util.spec_from_file_location("epsilon_uniform_sampler", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


APX_REF = "/root/reference/src/samplers/apx_ugs_sampler"
APX_OUT = os.path.join(OUT_DIR, "apx_ugs_sampler" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_apx(force: bool = False) -> str:
    """The reference apx_ugs_sampler module, compiled unmodified from its one source file (no deviation needed)."""
    if not os.path.isdir(APX_REF):
        raise FileNotFoundError(APX_REF)
    src = os.path.join(APX_REF, "src", "apx_ugs_sampler.cpp")
    hdr = os.path.join(APX_REF, "include", "apx_ugs_sampler.hpp")
    if not force and os.path.exists(APX_OUT) and all(os.path.getmtime(APX_OUT) >= os.path.getmtime(x) for x in (src, hdr)):
        return APX_OUT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-w", "-fopenmp", "-DTORCH_EXTENSION_NAME=apx_ugs_sampler",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{APX_REF}/include"]
    for p in ce.include_paths():
        cmd.append(f"-I{p}")
    cmd += [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}", src, "-o", APX_OUT]
    for p in ce.library_paths():
        cmd += [f"-L{p}", f"-Wl,-rpath,{p}"]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python"]
    subprocess.run(cmd, check=True)
    return APX_OUT


def load_apx():
    import importlib.util

    import torch  # noqa: F401

    path = APX_OUT if os.path.exists(APX_OUT) else build_apx()
    spec = importlib.util.spec_from_file_location("apx_ugs_sampler", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_eps(force="--force" in sys.argv))
    print(build_apx(force="--force" in sys.argv))
