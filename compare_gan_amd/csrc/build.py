"""Builds libcgamd.so (the C-ABI HIP library) in-tree; the oracle is pure Python, nothing to compile.

hipcc cross-compiles for gfx950 without a GPU.  Usage: python -m compare_gan_amd.csrc.build
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcgamd.so")
SOURCES = ["cg_error.hip", "cg_gconv.hip", "cg_conv_fast.hip", "cg_conv_halo.hip", "cg_conv_small.hip", "cg_multi.hip", "cg_elem.hip", "cg_bn.hip", "cg_sn.hip",
           "cg_optim.hip", "cg_attn.hip", "cg_fid.hip", "cg_tridiag.hip", "cg_calib.hip", "cg_ln.hip", "cg_head.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + HERE]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def _compile(src):
    obj = os.path.join(LIB_DIR, "obj", os.path.splitext(src)[0] + ".o")
    srcp = os.path.join(HERE, src)
    deps = [srcp, os.path.join(HERE, "cg_common.h"), os.path.join(HERE, "cg_conv_fast.h"),
            os.path.join(ROOT, "include", "cgamd.h")]
    if any(_newer(d, obj) for d in deps):
        cmd = [HIPCC] + FLAGS + ["-c", srcp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force=False):
    os.makedirs(os.path.join(LIB_DIR, "obj"), exist_ok=True)
    if force:
        for f in os.listdir(os.path.join(LIB_DIR, "obj")):
            os.remove(os.path.join(LIB_DIR, "obj", f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or any(_newer(o, LIB_PATH) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB_PATH


def build_timing():
    """libcgamd_timing.so: the same library with -DCG_CONV_TIMING (s_memtime phase stamps in
    fast_conv_kernel, read by scripts/conv_timing.py via CGAMD_LIB_PATH).  Debug aid only."""
    odir = os.path.join(LIB_DIR, "obj_timing")
    os.makedirs(odir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + FLAGS + ["-DCG_CONV_TIMING", "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        objs.append(obj)
    out = os.path.join(LIB_DIR, "libcgamd_timing.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return out


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build_timing())
    else:
        print(build(force="--force" in sys.argv))
