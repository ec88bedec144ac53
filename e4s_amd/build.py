"""Ahead-of-time build of libe4s_hip.so (gfx950) with hipcc.  No JIT at import, no torch headers:
the library is a plain C-ABI shared object (include/e4s_hip.h) loaded with ctypes."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libe4s_hip.so")
STAMP = os.path.join(HERE, ".libe4s_hip.stamp")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


# per-file flags.  conv_region1w.hip / conv_wino1w.hip: the SLP vectoriser re-packs the scalar f32 multiplies / FMAs of the operand split into v_pk_mul_f32 /
# v_pk_fma_f32, which cost ~22 cycles each beside MFMAs on gfx950 (guide: "packed f32 VALU ... an anti-lever beside MFMAs")
PER_FILE_FLAGS = {"conv_region1w.hip": ["-fno-slp-vectorize"], "conv_wino1w.hip": ["-fno-slp-vectorize"]}


def _extra_flags():
    """E4S_BUILD_ABLATIONS=1: profiling build whose conv kernels honour E4S_BF16X3_ABL / E4S_UPCONV_ABL (ablated
    variants compute wrong results by construction; never enabled in a product build)."""
    flags = ["-DE4S_ABLATIONS"] if os.environ.get("E4S_BUILD_ABLATIONS") == "1" else []
    return flags + os.environ.get("E4S_EXTRA_CFLAGS", "").split()          # A/B builds (with E4S_BUILD_OUT): e.g. -DE4S_WINO_READS_FIRST=0


def _digest():
    h = hashlib.sha256()
    h.update(" ".join(_extra_flags()).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    files = sources() + [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "e4s_hip.h")]
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libe4s_hip.so in-tree.  Returns the path.
    E4S_BUILD_OUT=<path.so>: a SECOND library beside the product one (profiling builds with E4S_BUILD_ABLATIONS=1, A/B builds; loaded
    with E4S_LIB_PATH): own object directory, no stamp, the product library is not touched."""
    alt = os.environ.get("E4S_BUILD_OUT")
    if alt:
        return _build_to(os.path.abspath(alt), os.path.join(os.path.dirname(os.path.abspath(alt)), "obj_" + os.path.basename(alt)), verbose)
    dig = _digest()
    if not force and os.path.isfile(LIB) and os.path.isfile(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC] + _extra_flags() + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        sys.stderr.write(res.stdout.decode(errors="replace"))
        raise RuntimeError("link of libe4s_hip.so failed")
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


def _build_to(lib, objdir, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + \
            _extra_flags() + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {src}")
    res = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        sys.stderr.write(res.stdout.decode(errors="replace"))
        raise RuntimeError("link failed")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
