"""Build the C-ABI shared library (gfx950 code objects) in-tree with hipcc.

    python -m brdf_nerf_amd.build [--verbose]

Produces brdf_nerf_amd/libbrdfnerf_hip.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbrdfnerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-pass-failed", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "brdfnerf_hip.h"),
                                                               os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False, force=False, defines=(), out=None):
    """defines/out: build a VARIANT library (A/B experiments, profiles/): objects go to build/<tag>/, the result to `out`;
    load it with BRDFNERF_HIP_LIB=<out>."""
    variant = bool(defines) or out is not None
    if not variant and not force and not needs_build():
        return LIB
    tag = "_".join(d.replace("=", "-") for d in defines) or "default"
    objdir = os.path.join(HERE, "build", tag) if variant else os.path.join(HERE, "build")
    out = out or (os.path.join(objdir, "libbrdfnerf_hip.so") if variant else LIB)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if os.path.exists(obj) and not force:
            hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
            hdr_t = max(hdr_t, os.path.getmtime(os.path.join(ROOT, "include", "brdfnerf_hip.h")))
            if os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
                continue
        cmd = [HIPCC] + FLAGS + ["-D" + d for d in defines] + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{log}\n")
        elif verbose:
            keep = [l for l in log.splitlines() if any(k in l for k in ("Function Name", "VGPRs:", "Spill", "ScratchSize",
                                                                       "warning", "LDS Size", "Occupancy"))]
            print("\n".join(keep))
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(verbose="--verbose" in sys.argv, force="--force" in sys.argv, defines=defs, out=outs[0] if outs else None))
