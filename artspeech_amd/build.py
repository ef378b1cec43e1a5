"""Build recipe for libartspeech_hip.so (hipcc, gfx950 only).  Idempotent: sources newer than their
object files are recompiled, then everything is linked in-tree next to this file.

Two flavours from the same sources:
  product (default)   libartspeech_hip.so        no ablation / tuning switch exists in it (AS_DIAG_* fold to constants)
  diagnostic (--diag) libartspeech_hip_diag.so   -DAS_DIAG: the AS_* environment switches and legacy kernels the tools/
                                                 use; loaded only when ARTSPEECH_DIAG_LIB=1 (artspeech_amd/_lib.py)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libartspeech_hip.so")
OBJ_DIAG = os.path.join(HERE, "csrc", "build", "diag")
LIB_DIAG = os.path.join(HERE, "libartspeech_hip_diag.so")

ARCH = "gfx950"
COMMON = ["-O3", "-fPIC", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
# per-file extra flags: metrics.hip keeps IEEE op-by-op arithmetic (arg-min pairs and the fp64 area
# function must be bit-reproducible), so no fused multiply-add contraction there.
SOURCES = {
    "error.cpp": [],
    "prof.hip": [],
    "gemm_f32.hip": [],
    "wgrad_f32.hip": [],
    "lin_f32.hip": [],
    "gemm_s6.hip": [],
    "rowops.hip": [],
    "gru.hip": [],
    "lstm.hip": [],
    "conv.hip": [],
    "attention.hip": [],
    "metrics.hip": ["-ffp-contract=off"],
    "artspeech.hip": [],
}


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def build(force=False, verbose=True, diag=False, defines=(), suffix=""):
    """defines / suffix (diagnostic flavour only): extra -D macros (compile-time ablations such as AS_S6_ABL=2) and a name
    suffix for the library and its object directory, e.g. libartspeech_hip_diag_abl2.so (ARTSPEECH_DIAG_LIB=<path> loads it)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OBJ, LIB = (OBJ_DIAG, LIB_DIAG) if diag else (globals()["OBJ"], globals()["LIB"])
    COMMON = globals()["COMMON"] + (["-DAS_DIAG"] if diag else [])
    if defines or suffix:
        assert diag, "extra defines are for the diagnostic flavour"
        COMMON += [f"-D{d}" for d in defines]
        OBJ, LIB = OBJ + suffix, LIB.replace(".so", suffix + ".so")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "artspeech_hip.h"))
    objs, relink = [], force
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [hipcc, f"--offload-arch={ARCH}", *COMMON, *extra, "-c", src, "-o", obj]
            if name.endswith(".cpp"):
                cmd = [hipcc, *COMMON, "-c", src, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink or not os.path.exists(LIB):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    sfx = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv, defines=defs, suffix=sfx))
